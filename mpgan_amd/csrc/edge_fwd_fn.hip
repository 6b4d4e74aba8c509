// mpg_edge_fwd_fn: the fused edge network's forward WITH the node network fn as the epilogue of every workgroup
// (mpgan/model.py:256-279 in one launch: fe + mask + sum/mean + cat((agg, x)) + fn).  The kernel is edge_fwd2_impl.h's
// (FN variants, chain2_impl.h's schedule for the three node layers); its six instantiation pairs compile side by side in
// edge_fwd_fn_d{0,1,2}s{0,1}.hip.  This unit holds the entry point: argument checks and the variant table.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpgan_amd.h"

#define MPG_FN_DECL(D, S) int mpg_edge_fwd_fn_d##D##s##S(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, bool sl, hipStream_t st)
MPG_FN_DECL(0, 0); MPG_FN_DECL(0, 1); MPG_FN_DECL(1, 0); MPG_FN_DECL(1, 1); MPG_FN_DECL(2, 0); MPG_FN_DECL(2, 1);
#undef MPG_FN_DECL

extern "C" int mpg_edge_fwd_fn(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, void* stream) {
    if (p->B <= 0 || p->N <= 0) return -1;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f) || c->alpha != p->alpha) return -4;
    if (!p->f16 || !c->f16 || (p->two_term != 0 && p->two_term != 1)) return -8;
    if (p->stageE2 != nullptr && (long long)p->B * ((p->N + 31) / 32) * p->N * 10240LL > 0x7fffffffLL) return -7;
    // what the epilogue form covers -- anything else: MPG_FN_NA, and the caller runs mpg_edge_fwd + mpg_chain
    // (sender chunks: only with arrival counters -- the last workgroup of a (jet, receiver block) adds the chunks up -- and
    // only in the eight-wave form: the per-mode units answer MPG_FN_NA otherwise)
    if (p->SC < 1 || (p->SC != 1 && p->tickets == nullptr) || p->N > 160 * p->SC || p->es != nullptr) return MPG_FN_NA;
    if ((p->N + p->SC - 1) / p->SC > 160) return MPG_FN_NA;
    if (c->nlayers != 3 || c->M != p->B * p->N || c->A2 == nullptr || c->in_thr != 0 || c->in_out != nullptr || c->seed != p->seed) return MPG_FN_NA;
    const int K = c->L[0].K;
    if (c->K1 != 192 || K < 192 || K > 224 || c->lda2 < K - 192 || c->L[1].K != c->L[0].N || c->L[2].K != c->L[1].N) return MPG_FN_NA;
    if (c->L[0].N < 225 || c->L[0].N > 256 || c->L[1].N < 225 || c->L[1].N > 256 || c->L[2].N < 1 || c->L[2].N > 256) return MPG_FN_NA;   // k-steps (14, 16, 16)
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
    bool sl = false;
    for (int l = 0; l < 3; ++l) {
        const MpgChainLayer& L = c->L[l];
        if (L.gateH != nullptr || L.resid != nullptr) return MPG_FN_NA;
        if (L.drop_thr != 0 && L.drop_thr != p->thr) return MPG_FN_NA;           // one dropout mode per launch
        if (L.out != nullptr && (size_t)c->M * L.ldo * 4 >= 0x7fffffffull) return MPG_FN_NA;
        const bool vec = L.N % 4 == 0 && (L.out == nullptr || (L.ldo % 4 == 0 && ((uintptr_t)L.out & 15) == 0));
        if (!vec) {
            if (l != 2) return MPG_FN_NA;
            sl = true;
        }
    }
    // (the sender chunks' slabs are addressed with 32-bit offsets through ONE buffer resource over all SC of them)
    if ((size_t)p->SC * p->B * p->N * 192 * 4 >= 0x7fffffffull) return MPG_FN_NA;
    if (c2 != nullptr) {   // the next layer's a | c projection on fn's output rows: its own mpg_chain call (one layer, K <= 32)
        const MpgChainLayer& L = c2->L[0];
        if (c2->nlayers != 1 || !c2->f16 || c2->M != c->M || c2->a_slabs != 1 || c2->in_thr != 0 || c2->in_out != nullptr) return -2;
        if (c2->A != c->L[2].out || c2->lda != c->L[2].ldo || c2->K1 != L.K || L.K != c->L[2].N || L.K > 32) return -2;
        if (L.N > 256 || L.N % 4 || L.out == nullptr || L.ldo % 4 || ((uintptr_t)L.out & 15) || (size_t)c2->M * L.ldo * 4 >= 0x7fffffffull) return -2;
        if (L.gateH != nullptr || L.resid != nullptr || L.drop_thr != 0 || L.act || c2->alpha != c->alpha) return -2;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool sg = p->sign3 != nullptr;
    switch (dm * 2 + (sg ? 1 : 0)) {
    case 0: return mpg_edge_fwd_fn_d0s0(p, c, c2, sl, st);
    case 1: return mpg_edge_fwd_fn_d0s1(p, c, c2, sl, st);
    case 2: return mpg_edge_fwd_fn_d1s0(p, c, c2, sl, st);
    case 3: return mpg_edge_fwd_fn_d1s1(p, c, c2, sl, st);
    case 4: return mpg_edge_fwd_fn_d2s0(p, c, c2, sl, st);
    default: return mpg_edge_fwd_fn_d2s1(p, c, c2, sl, st);
    }
}
