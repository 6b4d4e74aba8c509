// mpg_edge_bwd_fn: the fused edge network's data-gradient kernel WITH epilogue chains on every workgroup's own jet: the
// layer's input gradient dx (from da | dc and the node path) and, optionally, the next-lower MPLayer's node-network
// input-gradient chain on those rows -- the backward of mpgan/model.py:256-279 from dagg down to dx, and on through the fn of
// the layer below, in one launch.  The kernel is edge_bwd2_impl.h's (EPI variants, chain2_impl.h's schedule for the chains);
// its instantiations compile side by side in edge_bwd_fn_d{0,1,2}w{0,1}.hip.  This unit holds the entry point: argument
// checks and the variant table.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpgan_amd.h"

#define MPG_FN_DECL(D, W) int mpg_edge_bwd_fn_d##D##w##W(const MpgEdgeBwd* p, const MpgChain* cdx, const MpgChain* cnx, int epi, hipStream_t st)
MPG_FN_DECL(0, 0); MPG_FN_DECL(0, 1); MPG_FN_DECL(1, 0); MPG_FN_DECL(1, 1); MPG_FN_DECL(2, 0); MPG_FN_DECL(2, 1);
#undef MPG_FN_DECL

namespace {
// rows of whole 16-byte groups (the chains' vector stores)?
bool vec_rows(const MpgChainLayer& L) { return L.N % 4 == 0 && (L.out == nullptr || (L.ldo % 4 == 0 && ((uintptr_t)L.out & 15) == 0)); }
}  // namespace

extern "C" int mpg_edge_bwd_fn(const MpgEdgeBwd* p, const MpgChain* cdx, const MpgChain* cnx, void* stream) {
    if (p->B <= 0 || p->N <= 0 || cdx == nullptr) return -1;
    if (p->sign3 == nullptr || p->stageE2 == nullptr) return -3;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f)) return -4;
    if (!p->f16) return -8;
    if ((long long)p->B * p->N * 10240LL > 0x7fffffffLL && p->N <= 32) return -7;
    if (p->stageZ2 != nullptr && p->gexp == nullptr) return -9;
    // what the epilogue form covers -- anything else: MPG_FN_NA, and the caller runs mpg_edge_bwd + mpg_chain (+ mpg_chain)
    if (p->SC != 1 || p->N > 32 || p->es != nullptr) return MPG_FN_NA;       // a whole jet per workgroup: its dc rows are complete
    const int M = p->B * p->N;
    {   // dx = [da | dc] W + resid: one layer, K = 192 from the rows this kernel writes
        const MpgChainLayer& L = cdx->L[0];
        if (cdx->f16 || cdx->nlayers != 1 || cdx->M != M || cdx->a_slabs != 1 || cdx->in_thr != 0 || cdx->in_out != nullptr) return MPG_FN_NA;
        if (cdx->A != p->da || cdx->lda != 96 || cdx->K1 != 96 || cdx->A2 != p->dc || cdx->lda2 != 96 || L.K != 192) return MPG_FN_NA;
        if (L.N < 1 || L.N > 32 || L.out == nullptr || L.resid == nullptr || L.gateH != nullptr || L.bias != nullptr || L.act || L.drop_thr != 0) return MPG_FN_NA;
        if ((size_t)M * L.ldo * 4 >= 0x7fffffffull || (size_t)M * L.ldr * 4 >= 0x7fffffffull) return MPG_FN_NA;
    }
    const bool dx_vec = vec_rows(cdx->L[0]) && cdx->L[0].ldr % 4 == 0 && ((uintptr_t)cdx->L[0].resid & 15) == 0;
    int epi = dx_vec ? 1 : 3;
    if (cnx != nullptr) {   // the layer below: its node network's input-gradient chain on the dx rows
        if (!dx_vec) return MPG_FN_NA;
        const MpgChain* c = cnx;
        if (c->f16 || c->nlayers != 3 || c->M != M || c->a_slabs != 1 || c->seed != p->seed || c->alpha != p->alpha) return MPG_FN_NA;
        if (c->A != cdx->L[0].out || c->lda != cdx->L[0].ldo || c->K1 != c->L[0].K || c->L[0].K != cdx->L[0].N) return MPG_FN_NA;
        if (c->L[0].N < 225 || c->L[0].N > 256 || c->L[1].K != c->L[0].N || c->L[1].N < 225 || c->L[1].N > 256 || c->L[2].K != c->L[1].N ||
            c->L[2].N > 256) return MPG_FN_NA;                                                                  // k-steps (2, 16, 16)
        if (c->L[0].gateH == nullptr || c->L[1].gateH == nullptr || c->L[2].gateH != nullptr) return MPG_FN_NA;   // gates on layers 0 and 1
        if (c->in_thr != 0 && c->in_thr != p->thr) return MPG_FN_NA;
        if (c->in_out != nullptr && ((size_t)M * c->ld_in_out * 4 >= 0x7fffffffull || c->ld_in_out % 4 || ((uintptr_t)c->in_out & 15))) return MPG_FN_NA;
        for (int l = 0; l < 3; ++l) {
            const MpgChainLayer& L = c->L[l];
            if (L.resid != nullptr || L.bias != nullptr || L.act || L.drop_thr != 0) return MPG_FN_NA;
            if (L.gateH != nullptr && (L.ldh % 4 || ((uintptr_t)L.gateH & 15) || (size_t)M * L.ldh * 4 >= 0x7fffffffull)) return MPG_FN_NA;
            if (L.gateH != nullptr && L.gate_thr != 0 && L.gate_thr != p->thr) return MPG_FN_NA;
            if (L.out != nullptr && (size_t)M * L.ldo * 4 >= 0x7fffffffull) return MPG_FN_NA;
            if (!vec_rows(L)) {
                if (l != 2) return MPG_FN_NA;
                epi = 2;
            }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
    const bool needw = p->stageZ2 != nullptr;
    switch (dm * 2 + (needw ? 1 : 0)) {
    case 0: return mpg_edge_bwd_fn_d0w0(p, cdx, cnx, epi, st);
    case 1: return mpg_edge_bwd_fn_d0w1(p, cdx, cnx, epi, st);
    case 2: return mpg_edge_bwd_fn_d1w0(p, cdx, cnx, epi, st);
    case 3: return mpg_edge_bwd_fn_d1w1(p, cdx, cnx, epi, st);
    case 4: return mpg_edge_bwd_fn_d2w0(p, cdx, cnx, epi, st);
    default: return mpg_edge_bwd_fn_d2w1(p, cdx, cnx, epi, st);
    }
}
