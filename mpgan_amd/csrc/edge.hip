// mpg_edge_fwd (the fused edge network's forward: edge_fwd2_impl.h holds the kernel; this unit has the entry point and the
// no-dropout variants, the variants with dropout are edge_fwd_d1.hip / edge_fwd_d2.hip so that the three parts of the
// slow-to-compile template build side by side) and mpg_pack_weights (the weight images every fused kernel takes).
#include "edge_fwd1_impl.h"

int mpg_edge_fwd_d1(const MpgEdgeFwd* p, hipStream_t st);   // edge_fwd_d1.hip: byte-threshold dropout
int mpg_edge_fwd_q0(const MpgEdgeFwd* p, hipStream_t st);   // edge_fwd_q{0,1,2}.hip: with edge scalars, by dropout mode
int mpg_edge_fwd_q1(const MpgEdgeFwd* p, hipStream_t st);
int mpg_edge_fwd_q2(const MpgEdgeFwd* p, hipStream_t st);
int mpg_edge_fwd_d2(const MpgEdgeFwd* p, hipStream_t st);   // edge_fwd_d2.hip: one-bit dropout (p = 1/2)

namespace {

// ------------------------------------------------------------------------------------------
// weight image: [part hi|lo][tile m][k-tile q][k-step s][lane][8 x bf16]
template <bool F16>
__global__ void pack_kernel(const float* __restrict__ W, int ldw, int rows, int cols, int transpose, float scale,
                            void* __restrict__ img, int MT, int QT) {
    typedef typename FragT<F16>::type V;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nfrag = MT * QT * 2;
    if (idx >= nfrag * 64) return;
    const int frag = idx >> 6, lane = idx & 63;
    const int m = frag / (QT * 2), q = (frag >> 1) % QT, s = frag & 1;
    const int r = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = 32 * m + r, col = 32 * q + chain_rho(s, h, j);
        float x = 0.f;
        if (row < rows && col < cols) x = (transpose ? W[(size_t)col * ldw + row] : W[(size_t)row * ldw + col]) * scale;
        v[j] = x;
    }
    V hi, lo;
    split8(v, hi, lo);
    reinterpret_cast<V*>(img)[idx] = hi;
    reinterpret_cast<V*>(img)[(size_t)nfrag * 64 + idx] = lo;
}

}  // namespace


extern "C" int mpg_pack_weights(const float* W, int ldw, int rows, int cols, int transpose, float scale, int f16,
                                void* img, void* stream) {
    const int MT = (rows + 31) / 32, QT = (cols + 31) / 32;
    const int n = MT * QT * 2 * 64;
    if (f16)
        hipLaunchKernelGGL((pack_kernel<true>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, ldw, rows,
                           cols, transpose, scale, img, MT, QT);
    else
        hipLaunchKernelGGL((pack_kernel<false>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, ldw, rows,
                           cols, transpose, scale, img, MT, QT);
    return (int)hipGetLastError();
}

extern "C" int mpg_edge_fwd(const MpgEdgeFwd* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->SC <= 0) return -1;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f)) return -4;  // lrelu() is max(v, alpha v)
    if (!p->f16 || !f1_terms_ok(p)) return -8;             // fp16 hi/lo images and activations; a product form that is built
    if ((p->N + p->SC - 1) / p->SC > F2_LIST_MAX) return -6;  // senders per chunk (their list lives in LDS)
    // the parked E2 fragments (10,240 bytes per block) and the sign words are addressed with 32-bit offsets behind a buffer
    // descriptor whose record count is an int: the same limit as mpg_edge_bwd's, refused here, before anything is written
    if (p->stageE2 != nullptr && (long long)p->B * ((p->N + 31) / 32) * p->N * 10240LL > 0x7fffffffLL) return -7;
    hipStream_t st = (hipStream_t)stream;
#ifdef MPG_SINGLE_VARIANT  // tools/ubench/fwd_bench.hip: one dropout mode, seconds to compile
#ifdef MPG_FWD1   // (-DMPG_FWD1: the eight-wave form, edge_fwd1_impl.h)
    return f1_launch<MPG_SINGLE_VARIANT>(p, st);
#else
    return f2_launch<MPG_SINGLE_VARIANT>(p, st);
#endif
#else
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
    if (p->es != nullptr) {
        if (p->wq == nullptr) return -3;
        return dm == 0 ? mpg_edge_fwd_q0(p, st) : (dm == 1 ? mpg_edge_fwd_q1(p, st) : mpg_edge_fwd_q2(p, st));
    }
    return dm == 0 ? f1_launch<0>(p, st) : (dm == 1 ? mpg_edge_fwd_d1(p, st) : mpg_edge_fwd_d2(p, st));
#endif
}
