// mpg_edge_bwd: the entry point and the no-dropout variants of the data-gradient kernel (edge_bwd2_impl.h holds the
// kernel; the variants with dropout are edge_bwd2_d1.hip / edge_bwd2_d2.hip, so that the three parts of this
// slow-to-compile template build side by side).
#include "edge_bwd1_impl.h"

int mpg_edge_bwd_d1(const MpgEdgeBwd* p, hipStream_t st);   // edge_bwd2_d1.hip: byte-threshold dropout
int mpg_edge_bwd_d2(const MpgEdgeBwd* p, hipStream_t st);   // edge_bwd2_d2.hip: one-bit dropout (p = 1/2)
int mpg_edge_bwd_q0(const MpgEdgeBwd* p, hipStream_t st);   // edge_bwd2_q{0,1,2}.hip: with edge scalars, by dropout mode
int mpg_edge_bwd_q1(const MpgEdgeBwd* p, hipStream_t st);
int mpg_edge_bwd_q2(const MpgEdgeBwd* p, hipStream_t st);

extern "C" int mpg_edge_bwd(const MpgEdgeBwd* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->SC <= 0) return -1;
    if (p->sign3 == nullptr || p->stageE2 == nullptr) return -3;   // the forward's by-products: sign words of Z3, parked E2
    if (!(p->alpha >= 0.f && p->alpha <= 1.f)) return -4;
    if (!p->f16) return -8;   // the recomputed layer and both gradient products take fp16 images
    if ((p->N + p->SC - 1) / p->SC > B2_LIST_MAX) return -6;  // senders per chunk (the list of unmasked ones lives in LDS)
    const int RB = (p->N + 31) / 32;
    if ((long long)p->B * RB * p->N * (NFR2 * 1024) > 0x7fffffffLL) return -7;  // staging offsets are 32-bit
    if (p->stageZ2 != nullptr && p->gexp == nullptr) return -9;
    hipStream_t st = (hipStream_t)stream;
#ifdef MPG_SINGLE_VARIANT
#ifdef MPG_BWD1   // (-DMPG_BWD1: the eight-wave form, edge_bwd1_impl.h)
    return b1_launch<MPG_SINGLE_VARIANT>(p, st);
#else
    return b2_launch<MPG_SINGLE_VARIANT>(p, st);
#endif
#else
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
    if (p->es != nullptr) {
        if (p->wq == nullptr || p->des == nullptr || p->daq == nullptr) return -3;
        if ((p->N + p->SC - 1) / p->SC > B2_LIST_MAX_Q) return -6;
        return dm == 0 ? mpg_edge_bwd_q0(p, st) : (dm == 1 ? mpg_edge_bwd_q1(p, st) : mpg_edge_bwd_q2(p, st));
    }
    return dm == 0 ? b1_launch<0>(p, st) : (dm == 1 ? mpg_edge_bwd_d1(p, st) : mpg_edge_bwd_d2(p, st));
#endif
}
