// mpg_edge_bwd_fn: the fused edge network's data-gradient kernel WITH the node network's input-gradient chain as the
// prologue of every workgroup (the backward of mpgan/model.py:256-279 from dy down to da, dc in one launch).  The kernel is
// edge_bwd2_impl.h's (FNB variants, chain2_impl.h's schedule for the three transposed node layers); its six instantiation
// pairs compile side by side in edge_bwd_fn_d{0,1,2}w{0,1}.hip.  This unit holds the entry point: argument checks and the
// variant table.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpgan_amd.h"

#define MPG_FN_DECL(D, W) int mpg_edge_bwd_fn_d##D##w##W(const MpgEdgeBwd* p, const MpgChain* c, bool sl, hipStream_t st)
MPG_FN_DECL(0, 0); MPG_FN_DECL(0, 1); MPG_FN_DECL(1, 0); MPG_FN_DECL(1, 1); MPG_FN_DECL(2, 0); MPG_FN_DECL(2, 1);
#undef MPG_FN_DECL

extern "C" int mpg_edge_bwd_fn(const MpgEdgeBwd* p, const MpgChain* c, void* stream) {
    if (p->B <= 0 || p->N <= 0) return -1;
    if (p->sign3 == nullptr || p->stageE2 == nullptr) return -3;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f) || c->alpha != p->alpha) return -4;
    if (!p->f16) return -8;
    const int RB = (p->N + 31) / 32;
    if ((long long)p->B * RB * p->N * 10240LL > 0x7fffffffLL) return -7;
    if (p->stageZ2 != nullptr && p->gexp == nullptr) return -9;
    // what the prologue form covers -- anything else: MPG_FN_NA, and the caller runs mpg_chain + mpg_edge_bwd
    if (p->SC != 1 || p->N > 180 || p->es != nullptr) return MPG_FN_NA;
    if (c->f16 || c->nlayers != 3 || c->M != p->B * p->N || c->a_slabs != 1 || c->seed != p->seed) return MPG_FN_NA;
    // the chain must end in the very rows this kernel reads as dagg
    if (c->L[2].out != p->dagg || c->L[2].ldo != p->ld_dagg || c->L[2].N < 192) return MPG_FN_NA;
    if (c->L[0].K < 1 || c->L[0].K > 32 || c->K1 != c->L[0].K) return MPG_FN_NA;                              // k-steps (2, 16, 16)
    if (c->L[0].N < 225 || c->L[0].N > 256 || c->L[1].K != c->L[0].N || c->L[1].N < 225 || c->L[1].N > 256 || c->L[2].K != c->L[1].N ||
        c->L[2].N > 256) return MPG_FN_NA;
    if (c->L[0].gateH == nullptr || c->L[1].gateH == nullptr || c->L[2].gateH != nullptr) return MPG_FN_NA;   // gates on layers 0 and 1
    const uint32_t thr = p->thr;
    if (c->in_thr != 0 && c->in_thr != thr) return MPG_FN_NA;
    if (c->in_out != nullptr && (size_t)c->M * c->ld_in_out * 4 >= 0x7fffffffull) return MPG_FN_NA;
    bool sl = false;
    for (int l = 0; l < 3; ++l) {
        const MpgChainLayer& L = c->L[l];
        if (L.resid != nullptr || L.bias != nullptr || L.act || L.drop_thr != 0) return MPG_FN_NA;
        if (L.gateH != nullptr && (L.ldh % 4 || ((uintptr_t)L.gateH & 15) || (size_t)c->M * L.ldh * 4 >= 0x7fffffffull)) return MPG_FN_NA;
        if (L.gateH != nullptr && L.gate_thr != 0 && L.gate_thr != thr) return MPG_FN_NA;
        if (L.out != nullptr && (size_t)c->M * L.ldo * 4 >= 0x7fffffffull) return MPG_FN_NA;
        const bool vec = L.N % 4 == 0 && (L.out == nullptr || (L.ldo % 4 == 0 && ((uintptr_t)L.out & 15) == 0));
        if (!vec) {
            if (l != 2) return MPG_FN_NA;
            sl = true;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int dm = thr == 0 ? 0 : (thr == 128 ? 2 : 1);
    const bool needw = p->stageZ2 != nullptr;
    switch (dm * 2 + (needw ? 1 : 0)) {
    case 0: return mpg_edge_bwd_fn_d0w0(p, c, sl, st);
    case 1: return mpg_edge_bwd_fn_d0w1(p, c, sl, st);
    case 2: return mpg_edge_bwd_fn_d1w0(p, c, sl, st);
    case 3: return mpg_edge_bwd_fn_d1w1(p, c, sl, st);
    case 4: return mpg_edge_bwd_fn_d2w0(p, c, sl, st);
    default: return mpg_edge_bwd_fn_d2w1(p, c, sl, st);
    }
}
