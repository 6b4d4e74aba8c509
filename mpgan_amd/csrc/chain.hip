// Chained per-node MLP: up to three Linear(+LeakyReLU)(+Dropout) layers -- or their input-gradient
// chain -- applied to a block of 32 rows without the intermediate activations leaving the CU as operands.
//
// Replaces, per MPLayer call, the node network  fn = LinearNet([256, 256] -> out)  (mpgan/model.py:268-279 ->
// LinearNet.forward :70-85) and, in the backward, the three input-gradient products of the same layers; with
// one layer it is the  a | c = x W1'^T  projection in front of the edge network.
//
// Layout ("chain" layout of common.h / edge.hip): the 32 rows (nodes) of a workgroup sit on the MFMA column,
// features in accumulator registers.  The weights are the A operand (pre-packed fragment images, streamed
// from L2), the activations the B operand: a layer's output tile, converted to 16-bit hi/lo, IS a pair of
// B fragments of the next layer, so between layers the activations only pass through LDS as ready-made
// fragments (64 KiB for two layers of up to 256 features).  The eight waves of the workgroup split a layer's
// 32-feature output tiles; two waves per SIMD overlap one wave's epilogue with the other's MFMAs.
#include "common.h"
#include "../../include/mpgan_amd.h"
#include "chain_int.h"

#ifdef MPG_CHSTAMP  // diagnostic build (tools/chain_stamps.py): s_memtime at the phase boundaries, every wave of workgroup 0
__device__ unsigned long long g_ch_stamps[8 * 16];
#define CH_STAMP(i) do { ch_st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CH_STAMP(i) do {} while (0)
#endif

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int CH_MAXKS = 16;                       // k-steps of 16 features: K <= 256
constexpr int CH_FB_BYTES = CH_MAXKS * 2 * 1024;   // one fragment buffer: [k-step][hi|lo][lane] 16 B
constexpr int CH_BIAS_FLOATS = 3 * 256;            // biases of the first 256 outputs of each layer
constexpr int CH_LDS_BYTES = 2 * CH_FB_BYTES + CH_BIAS_FLOATS * 4;   // 68,608

MPG_DEV float4 chld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// KS0 / KS1 / KS2: the number of 16-feature k-steps of each layer when known at compile time (0 = run-time loop).  With
// them the k loop is straight-line code, which is what lets the prefetched weight fragments stay in flight: a branch
// inside the loop makes the compiler wait for EVERY outstanding load at each k-step (~600 clk of L2 latency each,
// measured with the MPG_CHSTAMP build: 9.5k clk for 48 MFMAs), the straight-line form waits for the oldest only.
template <bool F16, int KS0, int KS1, int KS2>
__global__ __launch_bounds__(512, 1) void chain_kernel(const MpgChain p) {
    typedef typename FragT<F16>::type V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32;
#ifdef MPG_CHSTAMP
    unsigned long long ch_st[16] = {};
#endif
    CH_STAMP(0);
    uint32_t seed_lo = 0, seed_hi = 0;
    if (p.seed != nullptr) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    // power-of-two operand scales (fp16 hi/lo carries its 22 bits only for |x| >= 2^-3, see edge_common.h): the
    // activations of every layer are multiplied by ascale before they are split, layer l's image was packed from
    // wscale_l * W, so its accumulators hold wscale_l * ascale * z
    const float ascale = p.ascale > 0.f ? p.ascale : 1.f;
    const int lane16 = lane * 16;
    // biases: one copy in LDS for the whole workgroup (zero where a layer has none / beyond nbias); visible after the
    // barrier that follows the input staging
    float* const sbias = reinterpret_cast<float*>(smem + 2 * CH_FB_BYTES);
    for (int i = tid; i < CH_BIAS_FLOATS; i += 512) {
        const int l = i >> 8, n = i & 255;
        float b = 0.f;
        if (l < p.nlayers && p.L[l].bias != nullptr && n < (p.L[l].nbias ? p.L[l].nbias : p.L[l].N)) b = p.L[l].bias[n];
        sbias[i] = b;
    }
    // Weight fragments come from L2 (~700 ns away): a wave keeps the first 8 k-steps of its NEXT tile in flight
    // while it works on the current one -- the preload of layer l+1's tile is issued before layer l's MFMAs (the
    // weights do not depend on the data), layer 0's before the input staging -- and refills each slot with
    // k-step + 8 of the same tile right after using it.
    V wbuf[2][8][2];
    auto layer_geom = [&](int l, int& KS, int& MT, int& nfrag) {
        const int QT = (p.L[l].K + 31) / 32;
        KS = 2 * QT; MT = (p.L[l].N + 31) / 32; nfrag = MT * QT * 2;
    };
    auto wfrag = [&](int l, int tile, int ks, int part) {
        int KS, MT, nfrag;
        layer_geom(l, KS, MT, nfrag);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.L[l].Wimg), 0, 2 * nfrag * 1024, 0x00020000);
        return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (part * nfrag + tile * KS + min(ks, KS - 1)) * 1024, 0));
    };
    auto preload = [&](auto lc, int tile) {
        MPG_CI(l, lc);
#pragma unroll
        for (int u = 0; u < 8; ++u) { wbuf[l & 1][u][0] = wfrag(l, tile, u, 0); wbuf[l & 1][u][1] = wfrag(l, tile, u, 1); }
    };
    {
        int KS, MT, nfrag;
        layer_geom(0, KS, MT, nfrag);
        preload(std::integral_constant<int, 0>{}, min(w, MT - 1));   // (idle waves load a valid tile: no branch)
    }

    // ---- stage the input rows as B fragments: unit = (k-step, lane) = 8 features of one row
    {
        const int K = p.L[0].K, KS = 2 * ((K + 31) / 32);
        V* fb = reinterpret_cast<V*>(smem);
        const bool vec1 = (p.lda % 4 == 0) && (p.K1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0) &&
                          (p.a_slab_stride % 4 == 0);
        const bool vec2 = p.A2 != nullptr && (p.lda2 % 4 == 0) && (p.K1 % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A2) & 15) == 0);
        for (int u = tid; u < KS * 64; u += 512) {
            const int ks = u >> 6, ln = u & 63, rr = ln & 31, hh = ln >> 5;
            const int m = m0 + rr;
            float v[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int f = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * half + 4 * hh;  // features f .. f+3
                float x4[4] = {0.f, 0.f, 0.f, 0.f};
                if (m < p.M) {
                    if (vec1 && f + 4 <= p.K1) {
                        for (int sl = 0; sl < p.a_slabs; ++sl) {
                            const float4 t = chld4(p.A + sl * p.a_slab_stride + (size_t)m * p.lda + f);
                            x4[0] += t.x; x4[1] += t.y; x4[2] += t.z; x4[3] += t.w;
                        }
                    } else if (vec2 && f >= p.K1 && f + 4 <= K) {
                        const float4 t = chld4(p.A2 + (size_t)m * p.lda2 + (f - p.K1));
                        x4[0] = t.x; x4[1] = t.y; x4[2] = t.z; x4[3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = f + e;
                            if (k < p.K1) {
                                for (int sl = 0; sl < p.a_slabs; ++sl) x4[e] += p.A[sl * p.a_slab_stride + (size_t)m * p.lda + k];
                            } else if (k < K) {
                                x4[e] = p.A2[(size_t)m * p.lda2 + (k - p.K1)];
                            }
                        }
                    }
                    if (p.in_thr) {  // backward of a trailing dropout: gate the incoming gradient
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            x4[e] = (f + e < K && drop_keep_f(seed_lo, seed_hi, p.in_tag, (uint32_t)m, f + e, p.in_thr)) ? x4[e] * p.in_scale : 0.f;
                    }
                    if (p.in_out != nullptr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (f + e < K) p.in_out[(size_t)m * p.ld_in_out + f + e] = x4[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * half + e] = x4[e] * ascale;
            }
            V hi, lo;
            split8(v, hi, lo);
            fb[(ks * 2 + 0) * 64 + ln] = hi;
            fb[(ks * 2 + 1) * 64 + ln] = lo;
        }
    }

    CH_STAMP(1);
    __syncthreads();  // input fragments staged
    CH_STAMP(2);

    static_for<0, 3>([&](auto lc) {
        MPG_CI(l, lc);
        if (l < p.nlayers) {
            const MpgChainLayer& L = p.L[l];
            const V* fin = reinterpret_cast<const V*>(smem + (l & 1) * CH_FB_BYTES);
            V* fout = reinterpret_cast<V*>(smem + ((l + 1) & 1) * CH_FB_BYTES);
            int KS, MT, nfrag;
            layer_geom(l, KS, MT, nfrag);
            const bool last = l + 1 == p.nlayers;
            // The next layer's first fragments are requested AFTER this tile's bias / gate operands: loads complete in
            // issue order, so whatever the k loop or the epilogue waits for must not sit behind 16 KiB of prefetch.
            auto preload_next = [&]() {
                if constexpr (l + 1 < 3) {
                    if (!last) {
                        int KSn, MTn, nfn;
                        layer_geom(l + 1, KSn, MTn, nfn);
                        preload(std::integral_constant<int, l + 1>{}, min(w, MTn - 1));
                    }
                }
            };
            if (w >= MT) preload_next();          // a wave without a tile in this layer may have one in the next
            for (int tile = w; tile < MT; tile += 8) {
                if (tile != w) preload(lc, tile);  // more than 8 tiles in a layer: later tiles load late
                f32x16 acc;
                const float zscale = (L.wscale > 0.f ? L.wscale : 1.f) * ascale, inv_zscale = 1.f / zscale;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.f;
                // the gate operand (an activation saved by the forward, in HBM) is requested before the MFMAs
                // (and so is the residual the epilogue adds).  Rows of 16-byte aligned groups of four take one float4 load per
                // group with clamped addresses and no branch in between -- all of a tile's loads are then in flight together.
                auto tile_rows = [&](const float* base, int ld, float (&out)[16]) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) out[k] = 0.f;
                    if (base == nullptr) return;
                    const int mr = m0 + r;
                    if ((ld % 4 == 0) && (L.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0)) {
                        const float live = mr < p.M ? 1.f : 0.f;
                        const float* row = base + (size_t)min(mr, p.M - 1) * ld;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = 32 * tile + 8 * g + 4 * h;
                            const float4 t4 = chld4(row + min(n, L.N - 4));
                            const float lv = n < L.N ? live : 0.f;
                            out[4 * g + 0] = t4.x * lv; out[4 * g + 1] = t4.y * lv; out[4 * g + 2] = t4.z * lv; out[4 * g + 3] = t4.w * lv;
                        }
                    } else if (mr < p.M) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int n = 32 * tile + 8 * g + 4 * h + t;
                                if (n < L.N) out[4 * g + t] = base[(size_t)mr * ld + n];
                            }
                    }
                };
                float hv[16], rv[16];
                tile_rows(L.gateH, L.ldh, hv);
                tile_rows(L.resid, L.ldr, rv);
                if (tile == w) preload_next();
                // B fragments (activations, LDS) are read one k-step ahead of their MFMAs
                V bh = fin[0 * 64 + lane], bl = fin[1 * 64 + lane];
                constexpr int KSC = l == 0 ? KS0 : (l == 1 ? KS1 : KS2);
                if constexpr (KSC > 0) {
                    static_for<0, KSC>([&](auto kc) {
                        MPG_CI(ks, kc);
                        constexpr int u = ks & 7, kn = ks + 1 < KSC ? ks + 1 : KSC - 1;
                        const V nh = fin[(kn * 2 + 0) * 64 + lane], nl = fin[(kn * 2 + 1) * 64 + lane];
                        acc = mfma3(wbuf[l & 1][u][0], wbuf[l & 1][u][1], bh, bl, acc);
                        bh = nh; bl = nl;
                        if constexpr (ks + 8 < KSC) { wbuf[l & 1][u][0] = wfrag(l, tile, ks + 8, 0); wbuf[l & 1][u][1] = wfrag(l, tile, ks + 8, 1); }
                        // (left alone the scheduler sinks each refill down to its use, eight k-steps later, to shorten
                        // its live range -- and the k-step then waits out the whole L2 latency)
                        __builtin_amdgcn_sched_barrier(0);
                    });
                } else
                for (int k0 = 0; k0 < KS; k0 += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int ks = k0 + u;
                        if (ks < KS) {
                            const int kn = min(ks + 1, KS - 1);
                            const V nh = fin[(kn * 2 + 0) * 64 + lane], nl = fin[(kn * 2 + 1) * 64 + lane];
                            acc = mfma3(wbuf[l & 1][u][0], wbuf[l & 1][u][1], bh, bl, acc);
                            bh = nh; bl = nl;
                            if (ks + 8 < KS) { wbuf[l & 1][u][0] = wfrag(l, tile, ks + 8, 0); wbuf[l & 1][u][1] = wfrag(l, tile, ks + 8, 1); }
                        }
                    }
                }
                CH_STAMP(3 + 3 * l);
                // ---- epilogue: register 4g+t  <->  feature 32 tile + 8g + 4h + t of row m0 + r
                const int m = m0 + r;
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * tile + 8 * g + 4 * h;
                    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (tile < 8) {   // (wave-uniform) the staged copy
                        const float4 b4 = *reinterpret_cast<const float4*>(sbias + 256 * l + n);
                        bias4[0] = b4.x; bias4[1] = b4.y; bias4[2] = b4.z; bias4[3] = b4.w;
                    } else if (L.bias != nullptr) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (n + t < (L.nbias ? L.nbias : L.N)) bias4[t] = L.bias[n + t];
                    }
                    float x4[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float x = acc[4 * g + t] * inv_zscale + bias4[t];
                        if (L.act) x = lrelu(x, p.alpha);
                        x4[t] = x;
                    }
                    // dropout masks: ONE hash word per (row, tile) in bit mode (p = 1/2), one per group of four features
                    // otherwise -- the same words drop_keep_f(row, feature) reads (common.h)
                    auto keep4 = [&](uint32_t tag, uint32_t thr, bool (&keep)[4]) {
                        if (thr == 128u) {
                            const uint32_t wd = drop_word(seed_lo, seed_hi, tag, (uint32_t)m, DROP_BIT_GRP + (uint32_t)tile) >> (8 * g + 4 * h);
#pragma unroll
                            for (int t = 0; t < 4; ++t) keep[t] = (wd >> t) & 1u;
                        } else {
                            const uint32_t wd = drop_word(seed_lo, seed_hi, tag, (uint32_t)m, (uint32_t)(8 * tile + 2 * g + h));
#pragma unroll
                            for (int t = 0; t < 4; ++t) keep[t] = drop_keep(wd, t, thr);
                        }
                    };
                    if (L.drop_thr) {
                        bool keep[4];
                        keep4(L.drop_tag, L.drop_thr, keep);
#pragma unroll
                        for (int t = 0; t < 4; ++t) x4[t] = keep[t] ? x4[t] * L.drop_scale : 0.f;
                    }
                    if (L.gateH != nullptr) {  // backward through (dropout o LeakyReLU) of the layer that produced H
                        bool keep[4] = {true, true, true, true};
                        if (L.gate_thr) keep4(L.gate_tag, L.gate_thr, keep);
                        const float gs = L.gate_thr ? L.gate_scale : 1.f;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float gt = L.gate_act ? lrelu_grad(hv[4 * g + t], p.alpha) : 1.f;
                            x4[t] *= keep[t] ? gt * gs : 0.f;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) x4[t] += rv[4 * g + t];
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (n + t >= L.N || m >= p.M) x4[t] = 0.f;  // padding stays exactly zero
                    if (L.out != nullptr && m < p.M) {
                        float* dst = L.out + (size_t)m * L.ldo + n;
                        if (n + 4 <= L.N && (L.ldo % 4 == 0)) *reinterpret_cast<float4*>(dst) = make_float4(x4[0], x4[1], x4[2], x4[3]);
                        else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) if (n + t < L.N) dst[t] = x4[t];
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[4 * g + t] = x4[t];
                }
                if (!last) {  // registers 8s .. 8s+7 are the B fragment of k-step (2 tile + s) of the next layer
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] *= ascale;
                    V hi, lo;
                    split8(v, hi, lo);
                    fout[((2 * tile + 0) * 2 + 0) * 64 + lane] = hi;
                    fout[((2 * tile + 0) * 2 + 1) * 64 + lane] = lo;
                    split8(v + 8, hi, lo);
                    fout[((2 * tile + 1) * 2 + 0) * 64 + lane] = hi;
                    fout[((2 * tile + 1) * 2 + 1) * 64 + lane] = lo;
                }
            }
            CH_STAMP(4 + 3 * l);
            __syncthreads();
            CH_STAMP(5 + 3 * l);
        }
    });
#ifdef MPG_CHSTAMP
    if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) g_ch_stamps[w * 16 + i] = ch_st[i];
    }
#endif
}

struct PackJobs { MpgPackJob j[MPG_PACK_MAX_JOBS]; int n; int frag0[MPG_PACK_MAX_JOBS + 1]; };

// all weight images of one network layer set in one launch (the per-image kernel is ~5 us of launch each)
__global__ void pack_many_kernel(const PackJobs jobs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int frag = idx >> 6, lane = idx & 63;
    if (frag >= jobs.frag0[jobs.n]) return;
    int q = 0;
    while (frag >= jobs.frag0[q + 1]) ++q;
    const MpgPackJob& J = jobs.j[q];
    const int rows = J.rows, cols = J.cols;  // of the packed matrix (= W^T when J.transpose)
    const int MT = (rows + 31) / 32, QT = (cols + 31) / 32, nfrag = MT * QT * 2;
    const int fr = frag - jobs.frag0[q];
    const int m = fr / (QT * 2), qq = (fr >> 1) % QT, s = fr & 1;
    const int r = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = 32 * m + r, col = 32 * qq + chain_rho(s, h, j);
        float x = 0.f;
        if (row < rows && col < cols) {
            int wr = J.transpose ? col : row, wc = J.transpose ? row : col;  // element of W
            if (J.row_split > 0) {  // stacked view: logical row n -> W[n % split, (n / split) * split_cols + col]
                wc += (wr / J.row_split) * J.split_cols;
                wr %= J.row_split;
            }
            x = J.W[(size_t)wr * J.ldw + wc] * J.scale;
        }
        v[j] = x;
    }
    if (J.status != nullptr) {   // range guard: fp16 images hold inf beyond 65504, and nothing downstream would notice before the loss
        int bad = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = fabsf(v[j]);
            bad |= !(a <= 3.0e38f) ? 2 : ((J.f16 && a > 65504.f) ? 1 : 0);
        }
        if (bad) atomicOr(J.status, bad);
    }
    if (J.f16) {
        f16x8 hi, lo;
        split8(v, hi, lo);
        reinterpret_cast<f16x8*>(J.img)[fr * 64 + lane] = hi;
        reinterpret_cast<f16x8*>(J.img)[(size_t)nfrag * 64 + fr * 64 + lane] = lo;
    } else {
        bf16x8 hi, lo;
        split8(v, hi, lo);
        reinterpret_cast<bf16x8*>(J.img)[fr * 64 + lane] = hi;
        reinterpret_cast<bf16x8*>(J.img)[(size_t)nfrag * 64 + fr * 64 + lane] = lo;
    }
}

}  // namespace

#ifdef MPG_CHSTAMP
extern "C" int mpg_debug_chain_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ch_stamps), sizeof(unsigned long long) * 8 * 16);
}
#endif

extern "C" int mpg_chain(const MpgChain* p, void* stream) {
    if (p->M <= 0 || p->nlayers < 1 || p->nlayers > 3) return -1;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f)) return -4;
    for (int l = 0; l < p->nlayers; ++l) {
        if (p->L[l].K <= 0 || p->L[l].K > 16 * CH_MAXKS || p->L[l].N <= 0) return -2;
        if (l + 1 < p->nlayers && (p->L[l].N > 16 * CH_MAXKS || p->L[l + 1].K != p->L[l].N)) return -2;
    }
    if (p->a_slabs < 1 || p->K1 > p->L[0].K || (p->K1 < p->L[0].K && p->A2 == nullptr)) return -2;
    hipStream_t st = (hipStream_t)stream;
    {   // the MPLayer shapes have their own schedule (chain2.hip)
        const int rc = mpg_chain2_try(p, st);
        if (rc != MPG_CHAIN2_NA) return rc;
    }
    dim3 grid((p->M + 31) / 32), block(512);
    // k-steps per layer; the shapes MPLayer uses have straight-line instantiations, anything else the run-time loops
    int ks[3] = {0, 0, 0};
    for (int l = 0; l < p->nlayers; ++l) ks[l] = 2 * ((p->L[l].K + 31) / 32);
#define MPG_CHAIN_LAUNCH(F16V, A, B, C)                                                               \
    do {                                                                                              \
        MPG_ENSURE_LDS((chain_kernel<F16V, A, B, C>), CH_LDS_BYTES);                                  \
        hipLaunchKernelGGL((chain_kernel<F16V, A, B, C>), grid, block, CH_LDS_BYTES, st, *p);         \
        return (int)hipGetLastError();                                                                \
    } while (0)
    if (p->f16) {
        if (ks[0] == 14 && ks[1] == 16 && ks[2] == 16) MPG_CHAIN_LAUNCH(true, 14, 16, 16);   // fn forward
        if (ks[0] == 2 && p->nlayers == 1) MPG_CHAIN_LAUNCH(true, 2, 0, 0);                   // a | c projection
        MPG_CHAIN_LAUNCH(true, 0, 0, 0);
    }
    if (ks[0] == 2 && ks[1] == 16 && ks[2] == 16) MPG_CHAIN_LAUNCH(false, 2, 16, 16);         // fn input-gradient chain
    if (ks[0] == 12 && p->nlayers == 1) MPG_CHAIN_LAUNCH(false, 12, 0, 0);                     // dx from da | dc
    MPG_CHAIN_LAUNCH(false, 0, 0, 0);
#undef MPG_CHAIN_LAUNCH
}

extern "C" int mpg_pack_many(const MpgPackJob* jobs, int njobs, void* stream) {
    if (njobs < 1 || njobs > MPG_PACK_MAX_JOBS) return -1;
    PackJobs pj;
    pj.n = njobs;
    pj.frag0[0] = 0;
    for (int q = 0; q < njobs; ++q) {
        pj.j[q] = jobs[q];
        pj.frag0[q + 1] = pj.frag0[q] + ((jobs[q].rows + 31) / 32) * ((jobs[q].cols + 31) / 32) * 2;
    }
    const int n = pj.frag0[njobs] * 64;
    hipLaunchKernelGGL(pack_many_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pj);
    return (int)hipGetLastError();
}
