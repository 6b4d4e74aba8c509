// Multi-head attention core of GAPT's MAB: per (jet, head)  P = softmax(q k^T / sqrt(d) + mask),  o = P v.
//
// Replaces the scaled-dot-product part of nn.MultiheadAttention as MAB uses it (gapt/model.py:107,
// :129: batch_first, key-ignore mask expanded over heads :127, need_weights=False) and its
// backward.  The projections around it (in_proj, out_proj, ff) run on mpg_gemm.  Sets here are
// tiny (L, S <= 150 particles or 10 inducing points, d = 16): the work is latency/launch bound,
// so one workgroup owns one (jet, head), keeps K and V in LDS and does the arithmetic in fp32 VALU.
#include "common.h"
#include "../../include/mpgan_amd.h"

namespace {
constexpr int DMAX = 32;

__global__ __launch_bounds__(64) void attn_fwd_kernel(const MpgAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int d = p.d;
    float* ks = sm;              // [S][d]
    float* vs = sm + p.S * d;    // [S][d]
    for (int t = threadIdx.x; t < p.S * d; t += blockDim.x) {
        const int s = t / d, c = t % d;
        ks[t] = p.k[(size_t)(b * p.S + s) * p.ldk + hd * d + c];
        vs[t] = p.v[(size_t)(b * p.S + s) * p.ldv + hd * d + c];
    }
    __syncthreads();
    const float scale = rsqrtf((float)d);
    for (int l = threadIdx.x; l < p.L; l += blockDim.x) {
        float q[DMAX];
        const float* qp = p.q + (size_t)(b * p.L + l) * p.ldq + hd * d;
        for (int c = 0; c < d; ++c) q[c] = qp[c] * scale;
        float* prow = p.P + ((size_t)(b * p.H + hd) * p.L + l) * p.S;
        float mx = -INFINITY;
        for (int s = 0; s < p.S; ++s) {
            float sc = 0.f;
            for (int c = 0; c < d; ++c) sc += q[c] * ks[s * d + c];
            if (p.ignore != nullptr && p.ignore[b * p.S + s] != 0.f) sc = -INFINITY;
            prow[s] = sc;
            mx = fmaxf(mx, sc);
        }
        float den = 0.f;
        for (int s = 0; s < p.S; ++s) {
            const float e = __expf(prow[s] - mx);
            prow[s] = e;
            den += e;
        }
        const float inv = 1.f / den;
        float o[DMAX];
        for (int c = 0; c < d; ++c) o[c] = 0.f;
        for (int s = 0; s < p.S; ++s) {
            const float pr = prow[s] * inv;
            prow[s] = pr;
            for (int c = 0; c < d; ++c) o[c] += pr * vs[s * d + c];
        }
        float* op = p.o + (size_t)(b * p.L + l) * p.ldo + hd * d;
        for (int c = 0; c < d; ++c) op[c] = o[c];
    }
}

// backward: dV = P^T dO ; dP = dO V^T ; dS = P * (dP - rowsum(dP * P)) ; dQ = dS K / sqrt(d) ; dK = dS^T Q / sqrt(d)
__global__ __launch_bounds__(64) void attn_bwd_kernel(const MpgAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int d = p.d;
    float* ks = sm;                    // [S][d]
    float* vs = ks + p.S * d;          // [S][d]
    float* qs = vs + p.S * d;          // [L][d]
    float* gs = qs + p.L * d;          // [L][d]  dO
    float* ds = gs + p.L * d;          // [L][S]  dS
    for (int t = threadIdx.x; t < p.S * d; t += blockDim.x) {
        const int s = t / d, c = t % d;
        ks[t] = p.k[(size_t)(b * p.S + s) * p.ldk + hd * d + c];
        vs[t] = p.v[(size_t)(b * p.S + s) * p.ldv + hd * d + c];
    }
    for (int t = threadIdx.x; t < p.L * d; t += blockDim.x) {
        const int l = t / d, c = t % d;
        qs[t] = p.q[(size_t)(b * p.L + l) * p.ldq + hd * d + c];
        gs[t] = p.d_o[(size_t)(b * p.L + l) * p.ldo + hd * d + c];
    }
    __syncthreads();
    const float scale = rsqrtf((float)d);
    const float* Pb = p.P + (size_t)(b * p.H + hd) * p.L * p.S;
    for (int l = threadIdx.x; l < p.L; l += blockDim.x) {
        float g[DMAX];
        for (int c = 0; c < d; ++c) g[c] = gs[l * d + c];
        float dot = 0.f;
        for (int s = 0; s < p.S; ++s) {
            float dp = 0.f;
            for (int c = 0; c < d; ++c) dp += g[c] * vs[s * d + c];
            const float pr = Pb[l * p.S + s];
            ds[l * p.S + s] = dp;
            dot += dp * pr;
        }
        float dq[DMAX];
        for (int c = 0; c < d; ++c) dq[c] = 0.f;
        for (int s = 0; s < p.S; ++s) {
            const float x = Pb[l * p.S + s] * (ds[l * p.S + s] - dot);
            ds[l * p.S + s] = x;
            for (int c = 0; c < d; ++c) dq[c] += x * ks[s * d + c];
        }
        float* qo = p.dq + (size_t)(b * p.L + l) * p.lddq + hd * d;
        for (int c = 0; c < d; ++c) qo[c] = dq[c] * scale;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < p.S; s += blockDim.x) {
        float dk[DMAX], dv[DMAX];
        for (int c = 0; c < d; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int l = 0; l < p.L; ++l) {
            const float x = ds[l * p.S + s], pr = Pb[l * p.S + s];
            for (int c = 0; c < d; ++c) { dk[c] += x * qs[l * d + c]; dv[c] += pr * gs[l * d + c]; }
        }
        float* ko = p.dk + (size_t)(b * p.S + s) * p.lddk + hd * d;
        float* vo = p.dv + (size_t)(b * p.S + s) * p.lddv + hd * d;
        for (int c = 0; c < d; ++c) { ko[c] = dk[c] * scale; vo[c] = dv[c]; }
    }
}
}  // namespace

extern "C" int mpg_attn_fwd(const MpgAttn* p, void* stream) {
    if (p->d > DMAX || p->B <= 0) return -1;
    const size_t lds = (size_t)2 * p->S * p->d * 4;
    if (lds > 160 * 1024) return -2;
    if (lds > 64 * 1024) HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(p->B * p->H), dim3(64), lds, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

extern "C" int mpg_attn_bwd(const MpgAttn* p, void* stream) {
    if (p->d > DMAX || p->B <= 0) return -1;
    const size_t lds = ((size_t)2 * p->S * p->d + 2 * p->L * p->d + (size_t)p->L * p->S) * 4;
    if (lds > 160 * 1024) return -2;
    static size_t attr = 0;
    if (lds > 64 * 1024 && lds > attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(p->B * p->H), dim3(64), lds, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}
