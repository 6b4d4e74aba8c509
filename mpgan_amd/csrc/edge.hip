// Fused fully-connected message-passing edge network (MPLayer's fe + mask + aggregation).
//
// Replaces, for the default MPLayer configuration, the reference's
//   _getA_fully_connected  mpgan/model.py:284-317  (x.repeat / cat -> [B*N*N, 2F] edge tensor)
//   self.fe(A)             mpgan/model.py:256 -> LinearNet.forward :70-85 (3x Linear+LeakyReLU+Dropout)
//   A * mask ; sum/mean    mpgan/model.py:257-267
// and their autograd backward.  The [B,N,N,*] edge activations never exist in memory.
//
// Layout ("chain" layout, see common.h): a wave owns one (jet, 32-receiver block) and walks the
// senders j.  Every activation tile is [features x receivers]: the receiver i sits on the MFMA
// column (= lane & 31), features sit in accumulator registers.  A layer's accumulator tile,
// converted to bf16 hi/lo, IS the B operand of the next layer's MFMA (no LDS round trip, no
// lane movement), weights are the A operand, pre-packed by pack_weights() into per-lane
// fragment images.  The sum over senders is then a per-lane register accumulation over the
// j loop, and layer 1 is the exact factorisation  W1 [x_i ; x_j] = a_i + c_j  (SURVEY.md A.3).
#include "edge_common.h"

#ifndef MPG_EXP
#define MPG_EXP 0  // experiment bits (tools/ubench): 1 no epilogue work, 4 no bias loads, 8 no LDS fragment loads
#endif

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

// DROP: 0 off, 1 byte-threshold dropout, 2 one-bit (p = 1/2) dropout.  SIGN: also write the per-lane sign words
// of Z3 the backward needs (a run-time test here would put a branch after every accumulator register).
template <int DROP, bool F16, bool SIGN>
__global__ __launch_bounds__(256, 1) void edge_fwd_kernel(const MpgEdgeFwd p) {
    constexpr bool WLDS = true;  // W3 hi+lo and W2 hi live in LDS (the only variant kept: streaming all of them spilled)
    typedef typename FragT<F16>::type V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int RB = (p.N + 31) / 32;
    int bid = blockIdx.x;
    const int sc = bid % p.SC; bid /= p.SC;
    const int rb = bid % RB;
    const int b = bid / RB;
    const int i = rb * 32 + r;
    const bool vi = i < p.N;
    const int JC = (p.N + p.SC - 1) / p.SC;
    const int jbeg = sc * JC, jend = min(p.N, jbeg + JC);

    const V* g2hi = reinterpret_cast<const V*>(p.W2img);
    const V* g2lo = g2hi + NF2 * 64;
    const V* g3hi = reinterpret_cast<const V*>(p.W3img);
    const V* g3lo = g3hi + NF3 * 64;
    V* l3hi = reinterpret_cast<V*>(smem);
    V* l3lo = l3hi + NF3 * 64;
    V* l2hi = l3lo + NF3 * 64;
    // biases and this chunk's sender terms c_j live in LDS too (a global load feeding the very next
    // instruction costs a full L2 round trip per tile)
    float* lbias = reinterpret_cast<float*>(smem + (WLDS ? FWD_W_BYTES : RED_BYTES));
    float* lc = lbias + (H2 + H3);
    if constexpr (WLDS) {
        if (!(p.skip_masked & 2)) {
            copy_to_lds(l3hi, g3hi, 2 * NF3 * 64, tid);  // hi and lo are adjacent in the image
            copy_to_lds(l2hi, g2hi, NF2 * 64, tid);
        }
    }
    const __amdgpu_buffer_rsrc_t r2 = img_rsrc(p.W2img, 2 * NF2);
    const int lane16 = lane * 16;
    const uint32_t lb3hi = lds_base(smem, lane16), lb3lo = lds_base(smem, NF3 * 1024 + lane16),
                   lb2hi = lds_base(smem, 2 * NF3 * 1024 + lane16);
    for (int t = tid; t < H2 + H3; t += 256) lbias[t] = t < H2 ? p.b2[t] * SC_E2 : p.b3[t - H2] * SC_E3;  // (biases in the accumulators' scale)
    const float* lb2 = lbias;
    const float* lb3 = lbias + H2;

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    // receiver term a_i in B-operand order: areg[q][s][4u+t] = a[i][32q+16s+8u+4h+t]
    float areg[T1][2][8];
    {
        const float* ai = p.a + (size_t)(b * p.N + (vi ? i : 0)) * (p.ld_ac ? p.ld_ac : H1);
#pragma unroll
        for (int q = 0; q < T1; ++q)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 t4 = ld4(ai + 32 * q + 16 * s + 8 * u + 4 * h);
                    areg[q][s][4 * u + 0] = vi ? t4.x * SC_A : 0.f;
                    areg[q][s][4 * u + 1] = vi ? t4.y * SC_A : 0.f;
                    areg[q][s][4 * u + 2] = vi ? t4.z * SC_A : 0.f;
                    areg[q][s][4 * u + 3] = vi ? t4.w * SC_A : 0.f;
                }
    }
    f32x16 agg[T3];
#pragma unroll
    for (int m = 0; m < T3; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) agg[m][k] = 0.f;

    // senders are walked in chunks of FWD_C_SLOTS whose terms c_j are staged in LDS
    for (int j0 = jbeg; j0 < jend; j0 += FWD_C_SLOTS) {
    const int j1 = min(jend, j0 + FWD_C_SLOTS);
    __syncthreads();  // previous chunk fully consumed (first pass: weight/bias fill issued)
    for (int t = tid; t < (j1 - j0) * (H1 / 4); t += 256) {
        const float4 c4 = ld4(p.c + (size_t)(b * p.N + j0 + t / (H1 / 4)) * (p.ld_ac ? p.ld_ac : H1) + 4 * (t % (H1 / 4)));
        reinterpret_cast<float4*>(lc)[t] = make_float4(c4.x * SC_A, c4.y * SC_A, c4.z * SC_A, c4.w * SC_A);
    }
    __syncthreads();
    for (int j = j0 + w; j < j1; j += 4) {
        const float mj = p.mask ? p.mask[b * p.N + j] : 1.f;
        if ((p.skip_masked & 1) && mj == 0.f) continue;  // wave-uniform
        float mjs = mj * p.dscale * (1.f / SC_E3);  // (the layer-3 output carries SC_E3)
        if (p.nbr != nullptr) {  // k-nearest-neighbour graph: sender j counts for this lane's receiver only if its bit is set
            const unsigned int wb = p.nbr[(size_t)(b * p.N + (vi ? i : 0)) * ((p.N + 31) >> 5) + (j >> 5)];
            mjs = ((wb >> (j & 31)) & 1u) ? mjs : 0.f;
        }
        const uint32_t erow = (uint32_t)((b * p.N + i) * p.N + j);
        const float* cj = lc + (j - j0) * H1;

        // ---- layer 1: e1 = drop(lrelu(a_i + c_j)), produced directly as B fragments
        V e1hi[T1][2], e1lo[T1][2];
#pragma unroll
        for (int q = 0; q < T1; ++q)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 c4 = ld4(cj + 32 * q + 16 * s + 8 * u + 4 * h);
                    const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
                    const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, q, 4 * s + 2 * u + h, h);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float x = lrelu(areg[q][s][4 * u + t] + cc[t], p.alpha);
                        v[4 * u + t] = drop_apply<DROP>(x, wd, 16 * s + 8 * u + t, t, p.thr);
                    }
                }
                split8(v, e1hi[q][s], e1lo[q][s]);
            }

        // ---- layers 2 and 3 as one sequence of 11 output tiles (5 of layer 2, 6 of layer 3).  A single wave has
        //      to keep the matrix pipe and the VALU busy by itself, and the hardware issues in order: an MFMA
        //      (32 clk) hides ~6 independent VALU/LDS instructions issued right behind it, no more.  So every
        //      MFMA is followed by one "slot" of at most a few instructions of OTHER work -- the epilogue of the
        //      PREVIOUS tile cut into units (one element: read / LeakyReLU / dropout [/ sum / sign bit]; half a
        //      pair split) and, in the last k-step, the bias load of the NEXT tile -- and sched_barrier pins it.
        V e2hi[T2][2], e2lo[T2][2];
        if (MPG_EXP & 1) {
#pragma unroll
            for (int mm = 0; mm < T2; ++mm) { e2hi[mm][0] = e1hi[0][0]; e2hi[mm][1] = e1hi[0][1]; e2lo[mm][0] = e1lo[0][0]; e2lo[mm][1] = e1lo[0][1]; }
        }
        {
            constexpr int NT = T2 + T3;
            V w2l[2][T1 * 2];
#pragma unroll
            for (int k = 0; k < T1 * 2; ++k) w2l[0][k] = img_frag<V>(r2, lane16, NF2 + k);
            f32x16 accs[2];
            float v2[16];
            PairSplit<V> ps[8];
            uint32_t sgn[T3 / 2] = {0u, 0u, 0u};
            auto bias_init = [&](auto Tc) {  // accumulator of tile T starts as its bias column
                MPG_CI(T, Tc);
                const float* bb = T < T2 ? lb2 + 32 * T : lb3 + 32 * (T - T2);
                f32x16& acc = accs[T & 1];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = ld4(bb + 8 * g + 4 * h);
                    acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
                }
            };
            // epilogue units of a layer-2 tile mm: q = 4 pair + {element 0, element 1, split half 1, split half 2}
            auto epi2_unit = [&](auto mc, auto qc) {
                MPG_CI(mm, mc); MPG_CI(q, qc);
                constexpr int pr = q >> 2, u = q & 3;
                if constexpr (u < 2) {
                    constexpr int e = 2 * pr + u, g = e >> 2, t = e & 3;
                    const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow, mm, 2 * g + h, h);
                    v2[e] = drop_apply<DROP>(lrelu(accs[mm & 1][e], p.alpha), wd, 8 * g + t, t, p.thr);
                } else if constexpr (u == 2) {
                    ps[pr].first(v2[2 * pr], v2[2 * pr + 1]);
                } else {
                    ps[pr].second(v2[2 * pr + 1], e2hi[mm][pr >> 2], e2lo[mm][pr >> 2], 2 * (pr & 3));
                }
            };
            // epilogue units of a layer-3 tile mm: q = element
            auto epi3_unit = [&](auto mc, auto ec) {
                MPG_CI(mm, mc); MPG_CI(e, ec);
                constexpr int g = e >> 2, t = e & 3;
                const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, mm, 2 * g + h, h);
                const float z = accs[(T2 + mm) & 1][e];
                // sign words: this lane shifts in the sign bit of each of its 96 Z3 registers in (tile, register)
                // order -> word tile>>1, bit 31 - (16 (tile & 1) + register); one v_alignbit each
                if constexpr (SIGN) sgn[mm >> 1] = __builtin_amdgcn_alignbit(sgn[mm >> 1], __builtin_bit_cast(uint32_t, z), 31);
                const float x = drop_apply<DROP>(lrelu(z, p.alpha), wd, 8 * g + t, t, p.thr);
                agg[mm][e] += mjs * x;
            };
            // slot SL of NS slots takes the units u of the previous tile with floor(u * NS / NU) == SL
            auto side = [&](auto Tc, auto slc, auto nsc) {
                MPG_CI(T, Tc); MPG_CI(SL, slc); MPG_CI(NS, nsc);
                if constexpr (T > 0 && !(MPG_EXP & 1)) {
                    constexpr int P = T - 1;
                    constexpr int NU = P < T2 ? 32 : 16;
                    constexpr int q0 = (SL * NU + NS - 1) / NS, q1 = SL + 1 >= NS ? NU : ((SL + 1) * NU + NS - 1) / NS;
                    static_for<q0, q1>([&](auto qc) {
                        if constexpr (P < T2) epi2_unit(std::integral_constant<int, P>{}, qc);
                        else epi3_unit(std::integral_constant<int, P - T2>{}, qc);
                    });
                }
            };
            bias_init(std::integral_constant<int, 0>{});
            static_for<0, NT>([&](auto Tc) {
                MPG_CI(T, Tc);
                constexpr bool l2 = T < T2;
                constexpr int m = l2 ? T : T - T2;
                constexpr int KS = l2 ? T1 * 2 : T2 * 2;
                constexpr int NS = 3 * KS - 3;  // the last k-step's slots are left to the next tile's bias load
                if constexpr (l2 && m + 1 < T2) {
#pragma unroll
                    for (int k = 0; k < T1 * 2; ++k) w2l[(m + 1) & 1][k] = img_frag<V>(r2, lane16, NF2 + (m + 1) * T1 * 2 + k);
                }
                auto load_hi = [&](int k) { return l2 ? lds_frag<V>(lb2hi, (m * T1 * 2 + k) * 1024) : lds_frag<V>(lb3hi, (m * T2 * 2 + k) * 1024); };
                auto load_lo = [&](int k) { return l2 ? w2l[m & 1][k < T1 * 2 ? k : 0] : lds_frag<V>(lb3lo, (m * T2 * 2 + k) * 1024); };
                V ah[2], al[2];
                ah[0] = load_hi(0);
                al[0] = load_lo(0);
                if (MPG_EXP & 8) { ah[1] = ah[0]; al[1] = al[0]; }
                f32x16& acc = accs[T & 1];
                static_for<0, KS>([&](auto kc) {
                    MPG_CI(k, kc);
                    if constexpr (k + 1 < KS && !(MPG_EXP & 8)) {
                        ah[(k + 1) & 1] = load_hi(k + 1);
                        al[(k + 1) & 1] = load_lo(k + 1);
                    }
                    V bh, bl;
                    if constexpr (l2) { bh = e1hi[k >> 1][k & 1]; bl = e1lo[k >> 1][k & 1]; }
                    else { bh = e2hi[k >> 1][k & 1]; bl = e2lo[k >> 1][k & 1]; }
                    using NSc = std::integral_constant<int, NS>;
                    if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[k & 1], bh, acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[k & 1], bh, acc, 0, 0, 0);
                    side(Tc, std::integral_constant<int, 3 * k + 0>{}, NSc{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k & 1], bl, acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[k & 1], bl, acc, 0, 0, 0);
                    side(Tc, std::integral_constant<int, 3 * k + 1>{}, NSc{});
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k & 1], bh, acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[k & 1], bh, acc, 0, 0, 0);
                    side(Tc, std::integral_constant<int, 3 * k + 2>{}, NSc{});
                    if constexpr (k == KS - 1 && T + 1 < NT && !(MPG_EXP & 4)) bias_init(std::integral_constant<int, T + 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            // drain: epilogue of the last tile
            if (!(MPG_EXP & 1)) static_for<0, 16>([&](auto qc) { epi3_unit(std::integral_constant<int, T3 - 1>{}, qc); });
            else {  // experiment: keep the accumulators alive without the epilogues
#pragma unroll
                for (int q = 0; q < 16; ++q) agg[0][q] += accs[0][q] + accs[1][q];
            }
            if constexpr (SIGN) {
                uint32_t* sg = p.sign3 + ((size_t)((b * RB + rb) * p.N + j)) * (T3 * 32) + lane;
#pragma unroll
                for (int q = 0; q < T3 / 2; ++q) sg[q * 64] = sgn[q];
            }
        }
    }
    }

    // ---- reduce the four waves' partial sums through LDS and write agg[b, i, :]
    __syncthreads();  // everyone is done with the weight copy
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int m = 0; m < T3; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) red[((w * T3 + m) * 16 + k) * 64 + lane] = agg[m][k];
    __syncthreads();
    float* out = p.agg + ((size_t)sc * p.B + b) * p.N * H3;
    for (int e = tid; e < T3 * 16 * 64; e += 256) {
        const int ln = e & 63, k = (e >> 6) & 15, m = e >> 10;
        const float sum = red[e] + red[e + T3 * 1024] + red[e + 2 * T3 * 1024] + red[e + 3 * T3 * 1024];
        const int ii = rb * 32 + (ln & 31);
        const int f = 32 * m + 8 * (k >> 2) + 4 * (ln >> 5) + (k & 3);
        if (ii < p.N) out[(size_t)ii * H3 + f] = sum * p.agg_scale;
    }
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
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB * p->SC), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
#define MPG_FWD_ONE(D, H, S)                                                                                      \
    do {                                                                                                          \
        MPG_ENSURE_LDS((edge_fwd_kernel<D, H, S>), FWD_LDS_BYTES);                                                \
        hipLaunchKernelGGL((edge_fwd_kernel<D, H, S>), grid, block, FWD_LDS_BYTES, st, *p);                       \
    } while (0)
#define MPG_FWD_S(D, H)                                                                                           \
    do { if (p->sign3 != nullptr) MPG_FWD_ONE(D, H, true); else MPG_FWD_ONE(D, H, false); } while (0)
#define MPG_FWD_H(D)                                                                                              \
    do { if (p->f16) MPG_FWD_S(D, true); else MPG_FWD_S(D, false); } while (0)
#ifdef MPG_SINGLE_VARIANT  // tools/ubench/fwd_bench.hip: one instantiation, seconds to compile
    MPG_FWD_ONE(MPG_SINGLE_VARIANT, true, true);
#else
    if (dm == 0) MPG_FWD_H(0);
    else if (dm == 1) MPG_FWD_H(1);
    else MPG_FWD_H(2);
#endif
#undef MPG_FWD_H
#undef MPG_FWD_S
#undef MPG_FWD_ONE
    return (int)hipGetLastError();
}
