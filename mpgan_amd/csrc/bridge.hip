// The rows between a GAPT generator's last attention block and the discriminator's first one (gapt/model.py:262-266 and
// :336-341): final_fc (Linear E -> F, F = the 3 particle features), tanh, then D's input_embedding (Linear F -> E, LeakyReLU,
// dropout).  Three row-local layers of almost no arithmetic that ran as three launches of the general GEMM and the tail kernel
// (8 + 5 + 8 us, each waiting for the one before; four more on the way back): here one launch each way.
//
// Sixteen lanes share a row: lane j holds columns 4j..4j+3 of the 64-wide sides (one float4 load / store per lane, a wave
// moves four whole rows per instruction), the F-wide middle is reduced over the sixteen lanes by a butterfly, so that every
// lane ends up with all of it.  fp32 FMA throughout, fixed order.
#include "common.h"
#include "../../include/mpgan_amd.h"

namespace {

MPG_DEV float act_fwd(int act, float t) { return act == 1 ? tanhf(t) : (act == 2 ? 1.f / (1.f + expf(-t)) : t); }
MPG_DEV float act_bwd(int act, float o) { return act == 1 ? 1.f - o * o : (act == 2 ? o * (1.f - o) : 1.f); }   // from the OUTPUT

// sum over the sixteen lanes of a row (lanes 16q .. 16q+15): every lane gets the total, same order in every lane group
MPG_DEV float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

constexpr int BR_FMAX = 8;

template <int F>
__global__ __launch_bounds__(256) void bridge_fwd_kernel(const MpgBridge p) {
    const int j = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= p.M) return;                              // (whole 16-lane groups leave together: the butterflies stay inside a group)
    uint32_t seed_lo = 0, seed_hi = 0;
    if (p.thr) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }
    float f[F];
    if (row >= p.row0) {
        // stage 1: feat = act1(W1 x + b1) for a generated row
        const float4 xv = *reinterpret_cast<const float4*>(p.x + (row - p.row0) * p.ldx + 4 * j);
#pragma unroll
        for (int k = 0; k < F; ++k) {
            const float4 w = *reinterpret_cast<const float4*>(p.W1 + k * p.K + 4 * j);
            float s = xv.x * w.x;
            s = fmaf(xv.y, w.y, s); s = fmaf(xv.z, w.z, s); s = fmaf(xv.w, w.w, s);
            s = row16_sum(s) + (p.b1 != nullptr ? p.b1[k] : 0.f);
            f[k] = act_fwd(p.act1, s);
        }
        if (j < F) {
            float v = f[0];
#pragma unroll
            for (int k = 1; k < F; ++k) v = j == k ? f[k] : v;
            p.feat[row * p.ldf + j] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < F; ++k) f[k] = p.feat[row * p.ldf + k];
    }
    if (p.e == nullptr) return;
    // stage 2: e = dropout(lrelu(W2 feat + b2)), outputs 4j .. 4j+3
    float o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = 4 * j + t;
        float s = p.b2 != nullptr ? p.b2[n] : 0.f;
#pragma unroll
        for (int k = 0; k < F; ++k) s = fmaf(p.W2[n * F + k], f[k], s);
        if (p.act2) s = s > 0.f ? s : p.alpha * s;
        if (p.thr) s = drop_keep_f(seed_lo, seed_hi, p.tag, (uint32_t)row, n, p.thr) ? s * p.dscale : 0.f;
        o[t] = s;
    }
    *reinterpret_cast<float4*>(p.e + row * p.lde + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
}

template <int F>
__global__ __launch_bounds__(256) void bridge_bwd_kernel(const MpgBridgeBwd p) {
    const int j = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= p.M) return;
    uint32_t seed_lo = 0, seed_hi = 0;
    if (p.thr) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }
    // through the dropout and the LeakyReLU of stage 2 (the sign read off the saved output, as mpg_gate does)
    const float4 gv = *reinterpret_cast<const float4*>(p.ge + row * p.ldge + 4 * j);
    const float4 ev = *reinterpret_cast<const float4*>(p.e + row * p.lde + 4 * j);
    const float gin[4] = {gv.x, gv.y, gv.z, gv.w}, eo[4] = {ev.x, ev.y, ev.z, ev.w};
    float g2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float gt = p.act2 ? lrelu_grad(eo[t], p.alpha) : 1.f;
        if (p.thr) gt = drop_keep_f(seed_lo, seed_hi, p.tag, (uint32_t)row, 4 * j + t, p.thr) ? gt * p.dscale : 0.f;
        g2[t] = gin[t] * gt;
    }
    if (p.g2 != nullptr) *reinterpret_cast<float4*>(p.g2 + row * p.ldg2 + 4 * j) = make_float4(g2[0], g2[1], g2[2], g2[3]);
    if (row < p.row0 || (p.g1 == nullptr && p.dx == nullptr)) return;
    // d feat = g2 W2 (+ what else flows into feat), through act1, then dx = g1 W1
    const long r1 = row - p.row0;
    float g1[F];
#pragma unroll
    for (int k = 0; k < F; ++k) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) s = fmaf(g2[t], p.W2[(4 * j + t) * F + k], s);
        s = row16_sum(s);
        if (p.gfeat != nullptr) s += p.gfeat[r1 * p.ldgf + k];
        g1[k] = s * act_bwd(p.act1, p.feat[row * p.ldf + k]);
    }
    if (p.g1 != nullptr && j < F) {
        float v = g1[0];
#pragma unroll
        for (int k = 1; k < F; ++k) v = j == k ? g1[k] : v;
        p.g1[r1 * p.ldg1 + j] = v;
    }
    if (p.dx != nullptr) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < F; ++k) {
            const float4 w = *reinterpret_cast<const float4*>(p.W1 + k * p.K + 4 * j);
            d.x = fmaf(g1[k], w.x, d.x); d.y = fmaf(g1[k], w.y, d.y); d.z = fmaf(g1[k], w.z, d.z); d.w = fmaf(g1[k], w.w, d.w);
        }
        *reinterpret_cast<float4*>(p.dx + r1 * p.lddx + 4 * j) = d;
    }
}

template <typename P>
int bridge_check(const P* p) {
    if (p->M <= 0 || p->row0 < 0 || p->row0 > p->M) return -1;
    if (p->K != 64 || p->E != 64 || p->F < 1 || (p->F > 4 && p->F != BR_FMAX)) return -2;   // (the sixteen-lane layout: 64-wide sides; 1..4 or 8 features)
    return 0;
}

}  // namespace

#define BR_DISPATCH(KERNEL, p, st)                                                                                        \
    switch ((p)->F) {                                                                                                     \
    case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(((p)->M + 15) / 16), dim3(256), 0, st, *(p)); break;                       \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(((p)->M + 15) / 16), dim3(256), 0, st, *(p)); break;                       \
    case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(((p)->M + 15) / 16), dim3(256), 0, st, *(p)); break;                       \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(((p)->M + 15) / 16), dim3(256), 0, st, *(p)); break;                       \
    default: hipLaunchKernelGGL(KERNEL<BR_FMAX>, dim3(((p)->M + 15) / 16), dim3(256), 0, st, *(p)); break;                \
    }

// float4 accesses: the leading dimensions are checked in floats, the base addresses here (a view into a flat parameter buffer
// at an offset that is not a multiple of four floats, or a storage-offset view of the rows, would be a misaligned vector access)
static inline bool br_al16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

extern "C" int mpg_bridge_fwd(const MpgBridge* p, void* stream) {
    if (const int rc = bridge_check(p)) return rc;
    if (p->feat == nullptr || p->ldf < p->F) return -3;
    if (p->row0 < p->M && (p->x == nullptr || p->W1 == nullptr || p->ldx % 4)) return -3;
    if (p->e != nullptr && (p->W2 == nullptr || p->lde % 4)) return -3;
    if (!br_al16(p->x, p->W1, p->e)) return -5;
    if (p->thr && p->seed == nullptr) return -4;
    BR_DISPATCH(bridge_fwd_kernel, p, (hipStream_t)stream);
    return (int)hipGetLastError();
}

extern "C" int mpg_bridge_bwd(const MpgBridgeBwd* p, void* stream) {
    if (const int rc = bridge_check(p)) return rc;
    if (p->ge == nullptr || p->e == nullptr || p->ldge % 4 || p->lde % 4 || (p->g2 != nullptr && p->ldg2 % 4)) return -3;
    if ((p->g1 != nullptr || p->dx != nullptr) && (p->feat == nullptr || p->W2 == nullptr)) return -3;
    if (p->dx != nullptr && (p->W1 == nullptr || p->lddx % 4)) return -3;
    if (p->thr && p->seed == nullptr) return -4;
    if (!br_al16(p->ge, p->e, p->g2, p->dx) || (p->dx != nullptr && !br_al16(p->W1))) return -5;
    BR_DISPATCH(bridge_bwd_kernel, p, (hipStream_t)stream);
    return (int)hipGetLastError();
}
