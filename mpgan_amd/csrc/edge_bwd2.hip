// mpg_edge_bwd: the entry point and the fp16-recompute variants of the data-gradient kernel (edge_bwd2_impl.h holds the
// kernel; the bf16-recompute variants -- OPTIONS["fwd_f16"] = False -- are edge_bwd2_bf16.hip, so that the two halves of
// this slow-to-compile template build side by side).
#include "edge_bwd2_impl.h"

int mpg_edge_bwd_bf16(const MpgEdgeBwd* p, hipStream_t st);   // edge_bwd2_bf16.hip

extern "C" int mpg_edge_bwd(const MpgEdgeBwd* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->SC <= 0) return -1;
    if (p->sign3 == nullptr) return -3;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f)) return -4;
    if ((p->N + p->SC - 1) / p->SC > B2_LIST_MAX) return -6;  // senders per chunk (the list of unmasked ones lives in LDS)
    const int RB = (p->N + 31) / 32;
    if ((long long)p->B * RB * p->N * (2 * NFR2 * 1024) > 0x7fffffffLL) return -7;  // staging offsets are 32-bit
    hipStream_t st = (hipStream_t)stream;
#ifdef MPG_SINGLE_VARIANT
    return b2_launch<true>(p, st);
#else
    return p->f16 ? b2_launch<true>(p, st) : mpg_edge_bwd_bf16(p, st);
#endif
}
