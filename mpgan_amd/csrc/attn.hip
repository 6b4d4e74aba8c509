// Multi-head attention core of GAPT's MAB: per (jet, head)  P = softmax(q k^T / sqrt(d) + mask),  o = P v.
//
// Replaces the scaled-dot-product part of nn.MultiheadAttention as MAB uses it (gapt/model.py:107,
// :129: batch_first, key-ignore mask expanded over heads :127, need_weights=False) and its
// backward.  The projections around it (in_proj, out_proj, ff) run on mpg_gemm.  Sets here are
// tiny (L, S <= 150 particles or 10 inducing points, d = 16): the work is latency/launch bound, fp32 VALU.
//
// Fast path (S <= 32 keys, L <= 64 queries, d in {8, 16, 32}: every GAPT block at N = 30): ONE WAVE per (jet, head),
// four per workgroup; lane = query.  K and V sit in LDS and are read as broadcasts, the 32 scores of a query
// live in registers (fully unrolled), P is written once, transposed ([S][L]: lanes contiguous).  The backward
// keeps dS and P in LDS between its query-major half (dQ) and its key-major half (dK, dV).
// Anything larger takes the generic kernels below (one workgroup per (jet, head), scores through memory).
#include "common.h"
#include "../../include/mpgan_amd.h"

namespace {
constexpr int DMAX = 32;
constexpr int AT_S = 32, AT_L = 64;  // fast-path limits

template <int D>
MPG_DEV void lds_row(const float* base, int row, float (&out)[D]) {  // broadcast read of one [D] row
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(base + row * D + c);
        out[c] = t.x; out[c + 1] = t.y; out[c + 2] = t.z; out[c + 3] = t.w;
    }
}

template <int D, bool TWO>
__global__ __launch_bounds__(256) void attn_fwd_fast(const MpgAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // TWO (L <= 32): each 32-lane half of a wave owns its own (jet, head) pair (8 pairs per workgroup), else the wave one
    const int w = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
    constexpr bool two = TWO;
    const int half = two ? lane64 >> 5 : 0, lane = two ? lane64 & 31 : lane64, nl = two ? 32 : 64;
    int pair = two ? (blockIdx.x * 4 + w) * 2 + half : blockIdx.x * 4 + w;   // (jet, head)
    const bool live = pair < p.B * p.H;
    if (!live) pair = p.B * p.H - 1;               // idle half: recompute the last pair, store nothing
    const int b = pair / p.H, hd = pair % p.H;
    float* ks = sm + (two ? w * 2 + half : w) * (2 * AT_S * D);   // [S][D]
    float* vs = ks + AT_S * D;
    for (int t = lane; t < p.S * (D / 4); t += nl) {
        const int s = t / (D / 4), c = 4 * (t % (D / 4));
        *reinterpret_cast<float4*>(ks + s * D + c) = *reinterpret_cast<const float4*>(p.k + (size_t)(b * p.S + s) * p.ldk + hd * D + c);
        *reinterpret_cast<float4*>(vs + s * D + c) = *reinterpret_cast<const float4*>(p.v + (size_t)(b * p.S + s) * p.ldv + hd * D + c);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int l = lane;
    if (l >= p.L || !live) return;
    const float scale = rsqrtf((float)D);
    float q[D];
    {
        const float* qp = p.q + (size_t)(b * p.L + l) * p.ldq + hd * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + c);
            q[c] = t.x * scale; q[c + 1] = t.y * scale; q[c + 2] = t.z * scale; q[c + 3] = t.w * scale;
        }
    }
    float sc[AT_S];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < AT_S; ++s) {
        float x = -INFINITY;
        if (s < p.S) {  // wave-uniform
            float kr[D];
            lds_row<D>(ks, s, kr);
            x = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) x += q[c] * kr[c];
            if (p.ignore != nullptr && p.ignore[b * p.S + s] != 0.f) x = -INFINITY;
        }
        sc[s] = x;
        mx = fmaxf(mx, x);
    }
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < AT_S; ++s) { sc[s] = __expf(sc[s] - mx); den += sc[s]; }
    const float inv = 1.f / den;
    float o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] = 0.f;
    float* Pt = p.P + (size_t)pair * p.S * p.L;    // [S][L]
#pragma unroll
    for (int s = 0; s < AT_S; ++s) {
        if (s < p.S) {
            const float pr = sc[s] * inv;
            Pt[s * p.L + l] = pr;
            float vr[D];
            lds_row<D>(vs, s, vr);
#pragma unroll
            for (int c = 0; c < D; ++c) o[c] += pr * vr[c];
        }
    }
    float* op = p.o + (size_t)(b * p.L + l) * p.ldo + hd * D;
#pragma unroll
    for (int c = 0; c < D; c += 4) *reinterpret_cast<float4*>(op + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
}

// backward: dV = P^T dO ; dP = dO V^T ; dS = P * (dP - rowsum(dP * P)) ; dQ = dS K / sqrt(d) ; dK = dS^T Q / sqrt(d)
template <int D, bool TWO>
__global__ __launch_bounds__(256) void attn_bwd_fast(const MpgAttn p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int w = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
    constexpr bool two = TWO;            // (L <= 32, S <= 32): a 32-lane half per (jet, head) pair
    const int half = two ? lane64 >> 5 : 0, lane = two ? lane64 & 31 : lane64, nl = two ? 32 : 64;
    int pair = two ? (blockIdx.x * 4 + w) * 2 + half : blockIdx.x * 4 + w;
    const bool live = pair < p.B * p.H;
    if (!live) pair = p.B * p.H - 1;     // idle half: recompute the last pair, store nothing
    const int b = pair / p.H, hd = pair % p.H;
    const int capl = two ? 32 : AT_L;    // query capacity of this pair's LDS slice
    // row stride of dS / P: odd, so that the key-major reads (one row per lane) spread over the banks -- except where
    // four wave-sized slices with d = 32 would no longer fit the LDS
    const int ldl = (!two && D == 32) ? capl : capl + 1;
    const int PER = 2 * AT_S * D + 2 * capl * D + 2 * AT_S * ldl;
    float* ks = sm + (two ? w * 2 + half : w) * PER;   // [S][D]
    float* vs = ks + AT_S * D;           // [S][D]
    float* qs = vs + AT_S * D;           // [L][D]
    float* gs = qs + capl * D;           // [L][D]  dO
    float* dsl = gs + capl * D;          // [S][ldl]  dS (written query-major: lanes contiguous)
    float* pl = dsl + AT_S * ldl;        // [S][ldl]  P
    for (int t = lane; t < p.S * (D / 4); t += nl) {
        const int s = t / (D / 4), c = 4 * (t % (D / 4));
        *reinterpret_cast<float4*>(ks + s * D + c) = *reinterpret_cast<const float4*>(p.k + (size_t)(b * p.S + s) * p.ldk + hd * D + c);
        *reinterpret_cast<float4*>(vs + s * D + c) = *reinterpret_cast<const float4*>(p.v + (size_t)(b * p.S + s) * p.ldv + hd * D + c);
    }
    for (int t = lane; t < p.L * (D / 4); t += nl) {
        const int l = t / (D / 4), c = 4 * (t % (D / 4));
        *reinterpret_cast<float4*>(qs + l * D + c) = *reinterpret_cast<const float4*>(p.q + (size_t)(b * p.L + l) * p.ldq + hd * D + c);
        *reinterpret_cast<float4*>(gs + l * D + c) = *reinterpret_cast<const float4*>(p.d_o + (size_t)(b * p.L + l) * p.ldo + hd * D + c);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float scale = rsqrtf((float)D);
    const float* Pt = p.P + (size_t)pair * p.S * p.L;  // [S][L]
    if (lane < p.L && live) {  // ---- query-major half
        const int l = lane;
        float g[D];
        lds_row<D>(gs, l, g);
        float dp[AT_S], pr[AT_S];
        float dot = 0.f;
#pragma unroll
        for (int s = 0; s < AT_S; ++s) {
            dp[s] = 0.f; pr[s] = 0.f;
            if (s < p.S) {
                pr[s] = Pt[s * p.L + l];
                float vr[D];
                lds_row<D>(vs, s, vr);
#pragma unroll
                for (int c = 0; c < D; ++c) dp[s] += g[c] * vr[c];
                dot += dp[s] * pr[s];
            }
        }
        float dq[D];
#pragma unroll
        for (int c = 0; c < D; ++c) dq[c] = 0.f;
#pragma unroll
        for (int s = 0; s < AT_S; ++s) {
            if (s < p.S) {
                const float x = pr[s] * (dp[s] - dot);
                dsl[s * ldl + l] = x;
                pl[s * ldl + l] = pr[s];
                float kr[D];
                lds_row<D>(ks, s, kr);
#pragma unroll
                for (int c = 0; c < D; ++c) dq[c] += x * kr[c];
            }
        }
        float* qo = p.dq + (size_t)(b * p.L + l) * p.lddq + hd * D;
#pragma unroll
        for (int c = 0; c < D; c += 4)
            *reinterpret_cast<float4*>(qo + c) = make_float4(dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < p.S && live) {  // ---- key-major half
        const int s = lane;
        float dk[D], dv[D];
#pragma unroll
        for (int c = 0; c < D; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int l = 0; l < p.L; ++l) {
            const float x = dsl[s * ldl + l], pr = pl[s * ldl + l];
            float qr[D], gr[D];
            lds_row<D>(qs, l, qr);
            lds_row<D>(gs, l, gr);
#pragma unroll
            for (int c = 0; c < D; ++c) { dk[c] += x * qr[c]; dv[c] += pr * gr[c]; }
        }
        float* ko = p.dk + (size_t)(b * p.S + s) * p.lddk + hd * D;
        float* vo = p.dv + (size_t)(b * p.S + s) * p.lddv + hd * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            *reinterpret_cast<float4*>(ko + c) = make_float4(dk[c] * scale, dk[c + 1] * scale, dk[c + 2] * scale, dk[c + 3] * scale);
            *reinterpret_cast<float4*>(vo + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
        }
    }
}

// ---- generic kernels: one workgroup per (jet, head), scores through memory (P is [L][S] here)
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
// The fast path is chosen by shape alone (forward and backward must agree on the layout of P); its float4 row
// accesses then REQUIRE 16-byte aligned bases and strides -- anything else is an error, not a silent fallback.
static bool attn_fast_shape(const MpgAttn* p) {
    return p->S <= AT_S && p->L <= AT_L && (p->d == 8 || p->d == 16 || p->d == 32);
}
static bool attn_aligned(const MpgAttn* p, bool bwd) {
    auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0; };
    bool ok = al(p->q, p->ldq) && al(p->k, p->ldk) && al(p->v, p->ldv);
    if (!bwd) ok = ok && al(p->o, p->ldo);
    else ok = ok && al(p->d_o, p->ldo) && al(p->dq, p->lddq) && al(p->dk, p->lddk) && al(p->dv, p->lddv);
    return ok;
}
}  // namespace

extern "C" int mpg_attn_fwd(const MpgAttn* p, void* stream) {
    if (p->d > DMAX || p->B <= 0) return -1;
    if (attn_fast_shape(p)) {
        if (!attn_aligned(p, false)) return -3;
        const bool two = p->L <= 32;
        const int ppw = two ? 8 : 4;  // (jet, head) pairs per workgroup
        const dim3 grid((p->B * p->H + ppw - 1) / ppw), block(256);
        const size_t lds = (size_t)ppw * 2 * AT_S * p->d * 4;
        hipStream_t st = (hipStream_t)stream;
#define MPG_ATTN_FWD(DV)                                                                          \
    do {                                                                                          \
        if (two) hipLaunchKernelGGL((attn_fwd_fast<DV, true>), grid, block, lds, st, *p);         \
        else hipLaunchKernelGGL((attn_fwd_fast<DV, false>), grid, block, lds, st, *p);            \
    } while (0)
        if (p->d == 8) MPG_ATTN_FWD(8);
        else if (p->d == 16) MPG_ATTN_FWD(16);
        else MPG_ATTN_FWD(32);
#undef MPG_ATTN_FWD
        return (int)hipGetLastError();
    }
    const size_t lds = (size_t)2 * p->S * p->d * 4;
    if (lds > 160 * 1024) return -2;
    MPG_ENSURE_LDS(attn_fwd_kernel, lds);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(p->B * p->H), dim3(64), lds, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

extern "C" int mpg_attn_bwd(const MpgAttn* p, void* stream) {
    if (p->d > DMAX || p->B <= 0) return -1;
    if (attn_fast_shape(p)) {
        if (!attn_aligned(p, true)) return -3;
        auto need = [&](bool two) {
            const int capl = two ? 32 : AT_L, ldl = (!two && p->d == 32) ? capl : capl + 1;  // as in the kernel
            return (size_t)(two ? 8 : 4) * (2 * AT_S * p->d + 2 * capl * p->d + 2 * AT_S * ldl) * 4;
        };
        const bool two = p->L <= 32 && need(true) <= 160 * 1024;  // d = 32 does not fit eight pairs
        const int ppw = two ? 8 : 4;
        const dim3 grid((p->B * p->H + ppw - 1) / ppw), block(256);
        const size_t lds = need(two);
        hipStream_t st = (hipStream_t)stream;
#define MPG_ATTN_BWD(DV, TW)                                                                                              \
    do {                                                                                                                  \
        MPG_ENSURE_LDS((attn_bwd_fast<DV, TW>), lds);                                                                     \
        hipLaunchKernelGGL((attn_bwd_fast<DV, TW>), grid, block, lds, st, *p);                                            \
    } while (0)
#define MPG_ATTN_BWD2(DV) do { if (two) MPG_ATTN_BWD(DV, true); else MPG_ATTN_BWD(DV, false); } while (0)
        if (p->d == 8) MPG_ATTN_BWD2(8);
        else if (p->d == 16) MPG_ATTN_BWD2(16);
        else MPG_ATTN_BWD2(32);
#undef MPG_ATTN_BWD2
#undef MPG_ATTN_BWD
        return (int)hipGetLastError();
    }
    const size_t lds = ((size_t)2 * p->S * p->d + 2 * p->L * p->d + (size_t)p->L * p->S) * 4;
    if (lds > 160 * 1024) return -2;
    MPG_ENSURE_LDS(attn_bwd_kernel, lds);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(p->B * p->H), dim3(64), lds, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}
