// Backward of the fused edge network (see edge.hip for the forward and the chain layout).
//
//   dZ3 = m_j * dagg_i * keep3 * phi'(Z3)      phi'(Z3) from the forward's saved sign words
//   dE2 = W3'^T dZ3 ;  dZ2 = dE2 * keep2 * phi'(Z2)     (Z2/E2 recomputed: one 90-MFMA layer)
//   dE1 = W2'^T dZ2 ;  dZ1 = dE1 * keep1 * phi'(Z1) ;  da_i = sum_j dZ1 ;  dc_j = sum_i dZ1
//   dW3 = s * sum_e dZ3 E2^T ;  dW2 = s * sum_e dZ2 E1^T ;  db3 = sum_e dZ3 ;  db2 = sum_e dZ2
//
// Two kernels.  edge_bwd_kernel walks the data-gradient chain exactly like the forward kernel (a wave
// per (jet, 32 receivers), senders in a loop, W3^T hi+lo and W2 hi resident in LDS, the rest
// streamed from L2) and, when weight gradients are wanted, parks E2 and dZ2 in memory as 16-bit
// hi/lo planes laid out [block of 32 receivers][feature][receiver], i.e. already transposed for the
// weight-gradient contraction over edges.  edge_dw_kernel then streams those planes, rebuilds the
// cheap operands (E1 from a_i + c_j, dZ3 from dagg and the sign words) and accumulates dW3/dW2/db3/db2
// in registers over its share of the edges; the per-workgroup partials are summed by a last small
// kernel that also undoes the fragment-order permutation of the feature indices.
#include "edge_common.h"

namespace {

constexpr int NF3T = T2 * T3 * 2;  // W3^T image: 5 row tiles x 6 k-tiles x 2 = 60 fragments
constexpr int NF2T = T1 * T2 * 2;  // W2^T image: 3 x 5 x 2 = 30
// LDS plan: W3^T hi|lo (bf16, 122,880) | dagg tile of the 32 receivers (24,576) | a tile (12,288) |
// b2 (640) | c_j of the current sender chunk (8 x 384 = 3,072)   = 163,456 B.  W2 and W2^T stream from L2.
constexpr int BWD_W_BYTES = 2 * NF3T * 1024;
constexpr int BWD_DG_BYTES = T3 * 4 * 64 * 16;
constexpr int BWD_A_BYTES = T1 * 4 * 64 * 16;
constexpr int BWD_C_SLOTS = 8;
constexpr int BWD_LDS_BYTES = BWD_W_BYTES + BWD_DG_BYTES + BWD_A_BYTES + H2 * 4 + BWD_C_SLOTS * H1 * 4;
#ifndef MPG_PF
#define MPG_PF 4
#endif
constexpr int RED_DA_BYTES = 4 * T1 * 16 * 64 * 4;

// staging: block blk = (b*RB + rb)*N + j ; plane part (0 hi, 1 lo) ; row fi = fragment-order feature
// index ((tile*2 + s)*16 + h*8 + jj) ; column = receiver lane & 31.   16-bit elements.
MPG_DEV size_t stage_off(size_t blk, int part, int fi) { return ((blk * 2 + part) * (size_t)H2 + fi) * 32; }

template <typename V>
MPG_DEV void stage_frag(void* base, size_t blk, int part, int tile, int s, int h, int r, const V f) {
    uint16_t* out = reinterpret_cast<uint16_t*>(base) + stage_off(blk, part, (tile * 2 + s) * 16 + h * 8) + r;
    typedef typename ElemOf<V>::type E;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const E x = f[jj];
        out[jj * 32] = *reinterpret_cast<const uint16_t*>(&x);
    }
}

template <int DROP, bool F16, bool NEEDW>  // DROP: 0 off, 1 byte mode, 2 bit mode (see common.h)
__global__ __launch_bounds__(256, 1) void edge_bwd_kernel(const MpgEdgeBwd p) {
    typedef typename FragT<F16>::type V;  // forward-recomputation operands; gradient operands are bf16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    const __amdgpu_buffer_rsrc_t r2 = img_rsrc(p.W2img, 2 * NF2);     // W2 hi | lo (forward image)
    const __amdgpu_buffer_rsrc_t r2t = img_rsrc(p.W2Timg, 2 * NF2T);  // W2^T hi | lo (bf16)
    const int lane16 = lane * 16;
    const bf16x8* t3g = reinterpret_cast<const bf16x8*>(p.W3Timg);
    bf16x8* l3thi = reinterpret_cast<bf16x8*>(smem);
    bf16x8* l3tlo = l3thi + NF3T * 64;
    float4* ldg = reinterpret_cast<float4*>(smem + BWD_W_BYTES);                 // [(m*4+g)][lane]
    float4* la = reinterpret_cast<float4*>(smem + BWD_W_BYTES + BWD_DG_BYTES);   // [(q*2+s)*2+u][lane]
    float* lb2 = reinterpret_cast<float*>(smem + BWD_W_BYTES + BWD_DG_BYTES + BWD_A_BYTES);
    float* lc = lb2 + H2;
    copy_to_lds(l3thi, t3g, 2 * NF3T * 64, tid);
    for (int t = tid; t < H2; t += 256) lb2[t] = p.b2[t];
    // per-receiver tiles shared by the four waves: upstream gradient dagg (scaled) and the layer-1
    // receiver term a, both in the register order the chain layout wants (zeros for padding lanes)
    for (int t = tid; t < T3 * 4 * 64; t += 256) {
        const int ln = t & 63, mg = t >> 6, rr = ln & 31, hh = ln >> 5, ii = rb * 32 + rr;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ii < p.N) {
            const float* di = p.dagg + (size_t)(b * p.N + ii) * p.ld_dagg + 32 * (mg >> 2) + 8 * (mg & 3) + 4 * hh;
            v4 = make_float4(di[0] * p.agg_scale, di[1] * p.agg_scale, di[2] * p.agg_scale, di[3] * p.agg_scale);
        }
        ldg[t] = v4;
    }
    for (int t = tid; t < T1 * 4 * 64; t += 256) {
        const int ln = t & 63, qsu = t >> 6, rr = ln & 31, hh = ln >> 5, ii = rb * 32 + rr;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ii < p.N) v4 = ld4(p.a + (size_t)(b * p.N + ii) * H1 + 8 * qsu + 4 * hh);
        la[t] = v4;
    }

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    float dacc[T1][2][8];
#pragma unroll
    for (int q = 0; q < T1; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) dacc[q][s][k] = 0.f;

    for (int j0 = jbeg; j0 < jend; j0 += BWD_C_SLOTS) {
    const int j1 = min(jend, j0 + BWD_C_SLOTS);
    __syncthreads();
    for (int t = tid; t < (j1 - j0) * (H1 / 4); t += 256)
        reinterpret_cast<float4*>(lc)[t] = reinterpret_cast<const float4*>(p.c + (size_t)(b * p.N + j0) * H1)[t];
    __syncthreads();
    for (int j = j0 + w; j < j1; j += 4) {
        const float mj = p.mask ? p.mask[b * p.N + j] : 1.f;
        const float mjs = mj * p.dscale;
        const uint32_t erow = (uint32_t)((b * p.N + i) * p.N + j);
        const size_t blk = (size_t)(b * RB + rb) * p.N + j;
        const float* cj = lc + (j - j0) * H1;
        uint32_t sw[T3 / 2];  // this lane's 96 sign bits of Z3 (requested now, used after layer 2)
#pragma unroll
        for (int q = 0; q < T3 / 2; ++q) sw[q] = p.sign3[blk * (T3 * 32) + q * 64 + lane];

        // ---- recompute layer 1
        V e1hi[T1][2], e1lo[T1][2];
#pragma unroll
        for (int q = 0; q < T1; ++q)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 c4 = ld4(cj + 32 * q + 16 * s + 8 * u + 4 * h);
                    const float4 a4 = la[((q * 2 + s) * 2 + u) * 64 + lane];
                    const float cc[4] = {c4.x + a4.x, c4.y + a4.y, c4.z + a4.z, c4.w + a4.w};
                    const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, q, 4 * s + 2 * u + h, h);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        v[4 * u + t] = drop_apply<DROP>(lrelu(cc[t], p.alpha), wd, 16 * s + 8 * u + t, t, p.thr);
                }
                split8(v, e1hi[q][s], e1lo[q][s]);
            }

        // ---- recompute layer 2 (E2 feeds dW3 and its zeros/signs give keep2 * phi'(Z2)); W2 hi+lo
        //      stream from L2 through a 5-deep fragment ring that runs across the five tiles
        V e2hi[T2][2], e2lo[T2][2];
        {
            constexpr int KS = T1 * 2, TOT = T2 * KS, PF = MPG_PF;
            V rh[PF + 1], rl[PF + 1];
#pragma unroll
            for (int k = 0; k < PF; ++k) { rh[k] = img_frag<V>(r2, lane16, k); rl[k] = img_frag<V>(r2, lane16, NF2 + k); }
            f32x16 accs[2];
            float v2[16];
            auto epi2 = [&](int mm, int g) {
                const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow, mm, 2 * g + h, h);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    v2[4 * g + t] = drop_apply<DROP>(lrelu(accs[mm & 1][4 * g + t], p.alpha), wd, 8 * g + t, t, p.thr);
                if (g == 1 || g == 3) {
                    const int s = g >> 1;
                    split8(v2 + 8 * s, e2hi[mm][s], e2lo[mm][s]);
                    if (NEEDW) {
                        stage_frag(p.stageE2, blk, 0, mm, s, h, r, e2hi[mm][s]);
                        stage_frag(p.stageE2, blk, 1, mm, s, h, r, e2lo[mm][s]);
                    }
                }
            };
#pragma unroll
            for (int kk = 0; kk < TOT; ++kk) {
                const int m = kk / KS, k = kk % KS;
                if (kk + PF < TOT) {
                    rh[(kk + PF) % (PF + 1)] = img_frag<V>(r2, lane16, kk + PF);
                    rl[(kk + PF) % (PF + 1)] = img_frag<V>(r2, lane16, NF2 + kk + PF);
                }
                f32x16& acc = accs[m & 1];
                if (k == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = ld4(lb2 + 32 * m + 8 * g + 4 * h);
                        acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
                    }
                }
                acc = mfma3(rh[kk % (PF + 1)], rl[kk % (PF + 1)], e1hi[k >> 1][k & 1], e1lo[k >> 1][k & 1], acc);
                if (m > 0 && k >= 1 && k <= 4) epi2(m - 1, k - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) epi2(T2 - 1, g);
        }

        // ---- dZ3 = m_j * dagg * keep3 * phi'(Z3), phi' from the forward's sign words
        bf16x8 z3hi[T3][2], z3lo[T3][2];
        {
#pragma unroll
            for (int m = 0; m < T3; ++m) {
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, m, 2 * g + h, h);
                    const float4 d4 = ldg[(m * 4 + g) * 64 + lane];
                    const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        // sign bit of this lane's Z3 register (tile m, 4g+t): see the forward's epi3
                        const int neg = __builtin_amdgcn_sbfe((int)sw[m >> 1], 31 - (16 * (m & 1) + 4 * g + t), 1);
                        const float d = mjs * dd[t], da = d * p.alpha;
                        const float sel = __builtin_bit_cast(float, (neg & __builtin_bit_cast(int, da)) | (~neg & __builtin_bit_cast(int, d)));
                        v[4 * g + t] = drop_apply<DROP>(sel, wd, 8 * g + t, t, p.thr);
                    }
                }
                split8(v, z3hi[m][0], z3lo[m][0]);
                split8(v + 8, z3hi[m][1], z3lo[m][1]);
            }
        }

        // ---- dE2 = W3'^T dZ3 (W3^T resident in LDS) ; dZ2 = dE2 * keep2 * phi'(Z2)
        bf16x8 z2hi[T2][2], z2lo[T2][2];
        {
            f32x16 accs[2];
            float v2[16];
#pragma unroll
            for (int m = 0; m <= T2; ++m) {
                const int mm = m - 1;
                auto epi4 = [&](int g) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int k = 4 * g + t;
                        // e2 of register k = element (k&7) of fragment k>>3: >0 -> 1, <0 -> alpha,
                        // ==0 -> dropped (dropout on) or z == 0 (dropout off: slope alpha, as torch)
                        const float e = (float)e2hi[mm][k >> 3][k & 7];
                        const float gt = e > 0.f ? 1.f : (e < 0.f ? p.alpha : (DROP ? 0.f : p.alpha));
                        v2[k] = accs[(m + 1) & 1][k] * gt;
                    }
                    if (g == 1 || g == 3) {
                        const int s = g >> 1;
                        split8(v2 + 8 * s, z2hi[mm][s], z2lo[mm][s]);
                        if (NEEDW) {
                            stage_frag(p.stageZ2, blk, 0, mm, s, h, r, z2hi[mm][s]);
                            stage_frag(p.stageZ2, blk, 1, mm, s, h, r, z2lo[mm][s]);
                        }
                    }
                };
                if (m < T2) {
                    f32x16& acc = accs[m & 1];
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
                    tile_chain<T3 * 2, bf16x8>(
                        acc, z3hi, z3lo,
                        [&](int k) { return l3thi[(m * T3 * 2 + k) * 64 + lane]; },
                        [&](int k) { return l3tlo[(m * T3 * 2 + k) * 64 + lane]; },
                        [&](int k) { if (m > 0 && (k & 1) && k < 8) epi4(k >> 1); });
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) epi4(g);
                }
            }
        }

        // ---- dE1 = W2'^T dZ2 (W2^T streamed from L2 through a 5-deep fragment ring across the
        //      three tiles) ; dZ1 = dE1 * keep1 * phi'(Z1) ; da_i += dZ1 ; dc_j = sum_i dZ1
        {
            float* dcj = p.dc + ((size_t)rb * p.B * p.N + (size_t)(b * p.N + j)) * H1;
            constexpr int KS = T2 * 2, TOT = T1 * KS, PF = MPG_PF;
            bf16x8 rh[PF + 1], rl[PF + 1];
#pragma unroll
            for (int k = 0; k < PF; ++k) { rh[k] = img_frag<bf16x8>(r2t, lane16, k); rl[k] = img_frag<bf16x8>(r2t, lane16, NF2T + k); }
            f32x16 accs[2];
            // dc_j = sum over the 32 receivers (lanes of one half) of dZ1: 16 values per tile and lane.  Halving
            // reduction: at each step a lane keeps half of its values and hands the other half to its partner
            // (DPP), so 16 values cost 15 exchanges instead of 80 and lane l ends with the total of value l & 15
            // (= accumulator register 8s + 4u + t  <->  feature 32 mm + 16 s + 8 u + 4 h + t).
            const bool lb0 = lane & 1, lb1 = lane & 2, lb2 = lane & 4, lb3 = lane & 8;
            float ured[4];
            auto epi5 = [&](int mm, int su) {  // slice (s,u) of tile mm
                const int s = su >> 1, u = su & 1;
                const float4 c4 = ld4(cj + 32 * mm + 16 * s + 8 * u + 4 * h);
                const float4 a4 = la[((mm * 2 + s) * 2 + u) * 64 + lane];
                const float cc[4] = {c4.x + a4.x, c4.y + a4.y, c4.z + a4.z, c4.w + a4.w};
                const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, mm, 4 * s + 2 * u + h, h);
                float dz[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float gt = drop_apply<DROP>(lrelu_grad(cc[t], p.alpha), wd, 16 * s + 8 * u + t, t, p.thr);
                    // accumulator register 8s + 4u + t  <->  element 4u+t of k-step s.  Padding lanes (i >= N) carry
                    // exact zeros all the way down: their dagg and a tiles are zero-filled.
                    dz[t] = accs[mm & 1][8 * s + 4 * u + t] * gt;
                    dacc[mm][s][4 * u + t] += dz[t];
                }
                const float w0 = halve_add<0xB1>(lb0, dz[0], dz[1]), w1 = halve_add<0xB1>(lb0, dz[2], dz[3]);
                ured[su] = halve_add<0x4E>(lb1, w0, w1);
                if (su == 3) {
                    const float x0 = halve_add<0x124>(lb2, ured[0], ured[1]), x1 = halve_add<0x124>(lb2, ured[2], ured[3]);
                    float y = halve_add<0x128>(lb3, x0, x1);
                    y += __shfl_xor(y, 16, 64);
                    const int e = lane & 15;
                    if (!(lane & 16)) dcj[32 * mm + 16 * (e >> 3) + 8 * ((e >> 2) & 1) + 4 * h + (e & 3)] = y;
                }
            };
#pragma unroll
            for (int kk = 0; kk < TOT; ++kk) {
                const int m = kk / KS, k = kk % KS;
                if (kk + PF < TOT) {
                    rh[(kk + PF) % (PF + 1)] = img_frag<bf16x8>(r2t, lane16, kk + PF);
                    rl[(kk + PF) % (PF + 1)] = img_frag<bf16x8>(r2t, lane16, NF2T + kk + PF);
                }
                f32x16& acc = accs[m & 1];
                if (k == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
                }
                acc = mfma3(rh[kk % (PF + 1)], rl[kk % (PF + 1)], z2hi[k >> 1][k & 1], z2lo[k >> 1][k & 1], acc);
                if (m > 0 && k >= 1 && k <= 4) epi5(m - 1, k - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int su = 0; su < 4; ++su) epi5(T1 - 1, su);
        }
    }
    }

    // ---- da: reduce over the four waves (disjoint sender subsets)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int q = 0; q < T1; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) red[((w * T1 + q) * 16 + 8 * s + k) * 64 + lane] = dacc[q][s][k];
    __syncthreads();
    float* out = p.da + ((size_t)sc * p.B + b) * p.N * H1;
    for (int e = tid; e < T1 * 16 * 64; e += 256) {
        const int ln = e & 63, k = (e >> 6) & 15, q = e >> 10;
        const float sum = red[e] + red[e + T1 * 1024] + red[e + 2 * T1 * 1024] + red[e + 3 * T1 * 1024];
        const int ii = rb * 32 + (ln & 31);
        const int f = 32 * q + 16 * (k >> 3) + 8 * ((k >> 2) & 1) + 4 * (ln >> 5) + (k & 3);
        if (ii < p.N) out[(size_t)ii * H1 + f] = sum;
    }
}


// ------------------------------------------------------------------------------------------------
// Weight gradients of fe.net.1 / fe.net.2.  One workgroup walks a contiguous range of 32-receiver
// blocks; per block it builds, in LDS, the four operand tiles [feature (fragment order)][receiver]
// as bf16 hi/lo planes -- dZ2 copied and E2 converted from the staged planes, dZ3 and E1 rebuilt --
// and its four waves accumulate their share of the 30 (dW3) + 15 (dW2) output tiles with
// A = dZ rows, B = E rows, contraction over the 32 receivers (2 k-steps).
constexpr int DW_LD = 40;  // tile row stride in 16-bit elements (80 B: conflict-free ds_read_b128)
constexpr int DW_ROWS = H3 + H2 + H2 + H1;  // Z3 | E2 | Z2 | E1
constexpr int DW_TILE_BYTES = DW_ROWS * 2 * DW_LD * 2;  // 97,280
constexpr int DW_TLD = 36;  // row stride (floats) of the dagg^T / a^T tiles: 32 would put every row on the same banks
constexpr int DW_LDS_BYTES = DW_TILE_BYTES + H3 * DW_TLD * 4 + H1 * DW_TLD * 4 + H1 * 4 + T3 * 16 * 8;  // + dagg^T + a^T + c_j + sign words

struct DwTile { int prod, m, n; };  // prod 0: dW3 (A = Z3 tile m, B = E2 tile n); 1: dW2 (A = Z2, B = E1)
__device__ constexpr DwTile DW_TILES[45] = {
    {0,0,0},{0,0,1},{0,0,2},{0,0,3},{0,0,4},{0,1,0},{0,1,1},{0,1,2},{0,1,3},{0,1,4},{1,0,0},{1,0,1},
    {0,2,0},{0,2,1},{0,2,2},{0,2,3},{0,2,4},{0,3,0},{0,3,1},{0,3,2},{0,3,3},{0,3,4},{1,0,2},{1,1,0},
    {0,4,0},{0,4,1},{0,4,2},{0,4,3},{0,4,4},{1,1,1},{1,1,2},{1,2,0},{1,2,1},{1,2,2},{1,3,0},
    {0,5,0},{0,5,1},{0,5,2},{0,5,3},{0,5,4},{1,3,1},{1,3,2},{1,4,0},{1,4,1},{1,4,2}};

MPG_DEV int feat_of_fi(int fi) {  // fragment-order index -> feature
    const int ms = fi >> 4, hh = (fi >> 3) & 1, jj = fi & 7;
    return 32 * (ms >> 1) + 16 * (ms & 1) + 8 * (jj >> 2) + 4 * hh + (jj & 3);
}

template <int BEGIN, int END>
MPG_DEV void dw_mfma(f32x16* acc, const __bf16* tiles, int lane) {
    const int rr = lane & 31, hh = lane >> 5;
    const __bf16* Z3h = tiles;                          // plane order: hi rows then lo rows per tensor
    const __bf16* Z3l = Z3h + H3 * DW_LD;
    const __bf16* E2h = Z3l + H3 * DW_LD;
    const __bf16* E2l = E2h + H2 * DW_LD;
    const __bf16* Z2h = E2l + H2 * DW_LD;
    const __bf16* Z2l = Z2h + H2 * DW_LD;
    const __bf16* E1h = Z2l + H2 * DW_LD;
    const __bf16* E1l = E1h + H1 * DW_LD;
#pragma unroll
    for (int t = BEGIN; t < END; ++t) {
        const DwTile d = DW_TILES[t];
        const __bf16* Ah = d.prod == 0 ? Z3h : Z2h;
        const __bf16* Al = d.prod == 0 ? Z3l : Z2l;
        const __bf16* Bh = d.prod == 0 ? E2h : E1h;
        const __bf16* Bl = d.prod == 0 ? E2l : E1l;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ao = (32 * d.m + rr) * DW_LD + 16 * s + 8 * hh;
            const int bo = (32 * d.n + rr) * DW_LD + 16 * s + 8 * hh;
            acc[t - BEGIN] = mfma3(*reinterpret_cast<const bf16x8*>(Ah + ao), *reinterpret_cast<const bf16x8*>(Al + ao),
                                   *reinterpret_cast<const bf16x8*>(Bh + bo), *reinterpret_cast<const bf16x8*>(Bl + bo),
                                   acc[t - BEGIN]);
        }
    }
}

template <int BEGIN, int END>
MPG_DEV void dw_store(const f32x16* acc, float* part3, float* part2, int lane) {
    const int cc = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = BEGIN; t < END; ++t) {
        const DwTile d = DW_TILES[t];
        float* dst = d.prod == 0 ? part3 : part2;
        const int ncol = d.prod == 0 ? H2 : H1;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = 32 * d.m + 8 * (k >> 2) + 4 * hh + (k & 3);
            dst[(size_t)row * ncol + 32 * d.n + cc] = acc[t - BEGIN][k];
        }
    }
}

// The whole per-workgroup loop is instantiated once per wave role (BEGIN..END = that wave's output
// tiles): a run-time branch per block around the MFMA section would make the accumulators merge at
// every join (hundreds of register moves and scratch spills).
//
// Build phase: work units are (tile row, piece of 8 receivers) = one 16-byte LDS write.  The staged
// dZ2 / E2 planes of block n+1 are requested into registers before the MFMAs of block n are issued
// (one workgroup per CU: nothing else would hide the HBM latency) and written to LDS afterwards.
constexpr int DW_PIECES = H2 * 4;  // (row, piece) units of one 160-row plane

template <int DROP, bool F16, int BEGIN, int END>
MPG_DEV void edge_dw_body(const MpgEdgeDw& p, char* smem, const int w) {
    typedef typename FragT<F16>::type E2V;
    __bf16* tiles = reinterpret_cast<__bf16*>(smem);
    __bf16* Z3h = tiles;
    __bf16* Z3l = Z3h + H3 * DW_LD;
    __bf16* E2h = Z3l + H3 * DW_LD;
    __bf16* E2l = E2h + H2 * DW_LD;
    __bf16* Z2h = E2l + H2 * DW_LD;
    __bf16* Z2l = Z2h + H2 * DW_LD;
    __bf16* E1h = Z2l + H2 * DW_LD;
    __bf16* E1l = E1h + H1 * DW_LD;
    float* dgT = reinterpret_cast<float*>(smem + DW_TILE_BYTES);  // [feature][32]
    float* aT = dgT + H3 * DW_TLD;                                // [feature][32 (+4 pad)]
    float* cj = aT + H1 * DW_TLD;                                 // [feature]
    uint32_t* lsg = reinterpret_cast<uint32_t*>(cj + H1);  // [3][64] sign words of the block
    const int tid = threadIdx.x, lane = tid & 63;
    const int RB = (p.N + 31) / 32;
    const int nblk = p.B * RB * p.N;
    const int per = (nblk + gridDim.x - 1) / gridDim.x;
    const int blk0 = blockIdx.x * per, blk1 = min(nblk, blk0 + per);

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    f32x16 acc[END - BEGIN];
#pragma unroll
    for (int t = 0; t < END - BEGIN; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float db3[3] = {0.f, 0.f, 0.f}, db2[3] = {0.f, 0.f, 0.f};  // unit n of this thread: row (tid>>2) + 64 n

    // staged planes of one block, as this thread's 16-byte pieces
    bf16x8 pz[5];      // dZ2: piece id = tid + 256 n over [part][row][pc]
    E2V peh[3], pel[3];  // E2: (row, pc) id = tid + 256 n, hi and lo planes
    auto prefetch = [&](int blk) {
        const bf16x8* sZ = reinterpret_cast<const bf16x8*>(p.stageZ2) + (size_t)blk * 2 * DW_PIECES;
        const E2V* sE = reinterpret_cast<const E2V*>(p.stageE2) + (size_t)blk * 2 * DW_PIECES;
#pragma unroll
        for (int n = 0; n < 5; ++n) pz[n] = sZ[tid + 256 * n];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int id = tid + 256 * n;
            if (id < DW_PIECES) { peh[n] = sE[id]; pel[n] = sE[DW_PIECES + id]; }
        }
    };
    if (blk0 < blk1) prefetch(blk0);

    int cur_brb = -1;
    for (int blk = blk0; blk < blk1; ++blk) {
        const int j = blk % p.N, brb = blk / p.N, rb = brb % RB, b = brb / RB;
        __syncthreads();  // previous block's MFMAs are done with the tiles
        if (brb != cur_brb) {  // new (jet, receiver block): refresh dagg^T and a^T
            cur_brb = brb;
            for (int t = tid; t < H3 * 32; t += 256) {
                const int f = t >> 5, ii = rb * 32 + (t & 31);
                dgT[f * DW_TLD + (t & 31)] = ii < p.N ? p.dagg[(size_t)(b * p.N + ii) * p.ld_dagg + f] * p.agg_scale : 0.f;
            }
            for (int t = tid; t < H1 * 32; t += 256) {
                const int f = t >> 5, ii = rb * 32 + (t & 31);
                aT[f * DW_TLD + (t & 31)] = ii < p.N ? p.a[(size_t)(b * p.N + ii) * H1 + f] : 0.f;
            }
        }
        if (tid < H1) cj[tid] = p.c[(size_t)(b * p.N + j) * H1 + tid];
        if (tid >= 64) lsg[tid - 64] = p.sign3[(size_t)blk * (T3 * 32) + tid - 64];
        // staged dZ2 pieces -> tile (bf16 planes as they are); bias sums
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int id = tid + 256 * n, part = id / DW_PIECES, row = (id % DW_PIECES) >> 2, pc = id & 3;
            *reinterpret_cast<bf16x8*>((part ? Z2l : Z2h) + row * DW_LD + 8 * pc) = pz[n];
        }
        // staged E2 pieces (fp16 or bf16 hi/lo) -> bf16 hi/lo
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int id = tid + 256 * n;
            if (id < DW_PIECES) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (float)peh[n][k] + (float)pel[n][k];
                bf16x8 hh, ll;
                split8(v, hh, ll);
                *reinterpret_cast<bf16x8*>(E2h + (id >> 2) * DW_LD + 8 * (id & 3)) = hh;
                *reinterpret_cast<bf16x8*>(E2l + (id >> 2) * DW_LD + 8 * (id & 3)) = ll;
            }
        }
        __syncthreads();  // dgT / aT / cj / lsg visible
        const float mj = p.mask ? p.mask[b * p.N + j] : 1.f;
        const float mjs = mj * p.dscale;
        // db2 from the dZ2 tile rows just written by this thread's own pieces is done below via LDS
        // dZ3 units: 192 rows x 4 pieces = 768 -> 3 per thread
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int id = tid + 256 * n, fi = id >> 2, pc = id & 3;
            const int f = feat_of_fi(fi);
            const int m = f >> 5, fl = f & 31;
            const int reg = 4 * (fl >> 3) + (fl & 3), hb = (fl >> 2) & 1;
            // the 8 receivers of this piece are lanes 32 hb + 8 pc + k; each holds the bit at the same position
            const int sh = 31 - (16 * (m & 1) + reg);
            const uint4 w0 = *reinterpret_cast<const uint4*>(lsg + (m >> 1) * 64 + 32 * hb + 8 * pc);
            const uint4 w1 = *reinterpret_cast<const uint4*>(lsg + (m >> 1) * 64 + 32 * hb + 8 * pc + 4);
            const uint32_t ws[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            float v[8];
            float s = 0.f;
            const float4 d0 = *reinterpret_cast<const float4*>(dgT + f * DW_TLD + 8 * pc);
            const float4 d1 = *reinterpret_cast<const float4*>(dgT + f * DW_TLD + 8 * pc + 4);
            const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int rcv = 8 * pc + k, ii = rb * 32 + rcv;
                float gt = ((ws[k] >> sh) & 1u) ? p.alpha : 1.f;
                if (DROP) {
                    const uint32_t erow = (uint32_t)((b * p.N + ii) * p.N + j);
                    if (!drop_keep_f(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, f, p.thr)) gt = 0.f;
                }
                const float x = ii < p.N ? mjs * dd[k] * gt : 0.f;
                v[k] = x;
                s += x;
            }
            db3[n] += s;
            bf16x8 hh, ll;
            split8(v, hh, ll);
            *reinterpret_cast<bf16x8*>(Z3h + fi * DW_LD + 8 * pc) = hh;
            *reinterpret_cast<bf16x8*>(Z3l + fi * DW_LD + 8 * pc) = ll;
        }
        // E1 units: 96 rows x 4 pieces = 384
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int id = tid + 256 * n;
            if (id < H1 * 4) {
                const int fi = id >> 2, pc = id & 3;
                const int f = feat_of_fi(fi);
                float v[8];
                const float4 a0 = *reinterpret_cast<const float4*>(aT + f * DW_TLD + 8 * pc);
                const float4 a1 = *reinterpret_cast<const float4*>(aT + f * DW_TLD + 8 * pc + 4);
                const float aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int rcv = 8 * pc + k, ii = rb * 32 + rcv;
                    float x = lrelu(aa[k] + cj[f], p.alpha);
                    if (DROP) {
                        const uint32_t erow = (uint32_t)((b * p.N + ii) * p.N + j);
                        if (!drop_keep_f(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, f, p.thr)) x = 0.f;
                    }
                    v[k] = x;
                }
                bf16x8 hh, ll;
                split8(v, hh, ll);
                *reinterpret_cast<bf16x8*>(E1h + fi * DW_LD + 8 * pc) = hh;
                *reinterpret_cast<bf16x8*>(E1l + fi * DW_LD + 8 * pc) = ll;
            }
        }
        // db2: this thread's dZ2 pieces (hi + lo planes hold the same rows at n and n + 2.5 -> sum both)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int id = tid + 256 * n;
            if (id < DW_PIECES) {
                const bf16x8 zh = *reinterpret_cast<const bf16x8*>(Z2h + (id >> 2) * DW_LD + 8 * (id & 3));
                const bf16x8 zl = *reinterpret_cast<const bf16x8*>(Z2l + (id >> 2) * DW_LD + 8 * (id & 3));
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += (float)zh[k] + (float)zl[k];
                db2[n] += s;
            }
        }
        __syncthreads();
        if (blk + 1 < blk1) prefetch(blk + 1);
        dw_mfma<BEGIN, END>(acc, tiles, lane);
    }

    // ---- per-workgroup partials (fragment-order indices; edge_dw_reduce undoes the permutation)
    float* part = p.part + (size_t)blockIdx.x * (H3 * H2 + H2 * H1 + H3 + H2);
    float* part3 = part, *part2 = part + H3 * H2, *pb3 = part2 + H2 * H1, *pb2 = pb3 + H3;
    dw_store<BEGIN, END>(acc, part3, part2, lane);
    // bias sums: add the 4 pieces of a row (4 adjacent threads)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        float x = db3[n], y = db2[n];
        x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64);
        y += __shfl_xor(y, 1, 64); y += __shfl_xor(y, 2, 64);
        const int row = (tid >> 2) + 64 * n;
        if ((tid & 3) == 0) {
            pb3[row] = x;
            if (row < H2) pb2[row] = y;
        }
    }
}

template <int DROP, bool F16>
__global__ __launch_bounds__(256, 1) void edge_dw_kernel(const MpgEdgeDw p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (w == 0) edge_dw_body<DROP, F16, 0, 12>(p, smem, w);
    else if (w == 1) edge_dw_body<DROP, F16, 12, 24>(p, smem, w);
    else if (w == 2) edge_dw_body<DROP, F16, 24, 35>(p, smem, w);
    else edge_dw_body<DROP, F16, 35, 45>(p, smem, w);
}

// out = scale * sum over workgroup partials, feature indices mapped back from fragment order.
// 32 outputs x 8 partial-slices per block: the 256 partials of an output are read by 8 threads.
__global__ __launch_bounds__(256) void edge_dw_reduce(const float* __restrict__ part, int nwg, float scale,
                                                      float* __restrict__ dW3, float* __restrict__ dW2,
                                                      float* __restrict__ db3, float* __restrict__ db2) {
    __shared__ float red[8][32];
    constexpr int PER = H3 * H2 + H2 * H1 + H3 + H2;
    const int ix = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + ix;
    float s = 0.f;
    if (idx < PER)
        for (int g = sl; g < nwg; g += 8) s += part[(size_t)g * PER + idx];
    red[sl][ix] = s;
    __syncthreads();
    if (sl != 0 || idx >= PER) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][ix];
    if (idx < H3 * H2) {
        const int r3 = idx / H2, c2 = idx % H2;
        dW3[feat_of_fi(r3) * H2 + feat_of_fi(c2)] = s * scale;
    } else if (idx < H3 * H2 + H2 * H1) {
        const int k = idx - H3 * H2, r2 = k / H1, c1 = k % H1;
        dW2[feat_of_fi(r2) * H1 + feat_of_fi(c1)] = s * scale;
    } else if (idx < H3 * H2 + H2 * H1 + H3) {
        db3[feat_of_fi(idx - H3 * H2 - H2 * H1)] = s;
    } else {
        db2[feat_of_fi(idx - H3 * H2 - H2 * H1 - H3)] = s;
    }
}

}  // namespace

extern "C" int mpg_edge_bwd(const MpgEdgeBwd* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->SC <= 0) return -1;
    if (p->sign3 == nullptr) return -3;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f)) return -4;
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB * p->SC), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool needw = p->stageE2 != nullptr && p->stageZ2 != nullptr;
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
#define MPG_BWD_ONE(D, H, W)                                                                                      \
    do {                                                                                                          \
        static bool done = false;                                                                                 \
        if (!done) {                                                                                              \
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_bwd_kernel<D, H, W>,                              \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS_BYTES));        \
            done = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((edge_bwd_kernel<D, H, W>), grid, block, BWD_LDS_BYTES, st, *p);                       \
    } while (0)
#define MPG_BWD_W(D, H)                                                                                           \
    do { if (needw) MPG_BWD_ONE(D, H, true); else MPG_BWD_ONE(D, H, false); } while (0)
#define MPG_BWD_H(D)                                                                                              \
    do { if (p->f16) MPG_BWD_W(D, true); else MPG_BWD_W(D, false); } while (0)
    if (dm == 0) MPG_BWD_H(0);
    else if (dm == 1) MPG_BWD_H(1);
    else MPG_BWD_H(2);
#undef MPG_BWD_H
#undef MPG_BWD_W
#undef MPG_BWD_ONE
    return (int)hipGetLastError();
}

extern "C" int mpg_edge_dw(const MpgEdgeDw* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->nwg <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p->nwg), block(256);
    const int dm = p->thr == 0 ? 0 : 1;  // per-element decisions: drop_keep_f picks the bit form itself
#define MPG_DW_ONE(D, H)                                                                                          \
    do {                                                                                                          \
        static bool done = false;                                                                                 \
        if (!done) {                                                                                              \
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)edge_dw_kernel<D, H>,                                  \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES));         \
            done = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL((edge_dw_kernel<D, H>), grid, block, DW_LDS_BYTES, st, *p);                            \
    } while (0)
    if (dm && p->f16) MPG_DW_ONE(1, true);
    else if (dm) MPG_DW_ONE(1, false);
    else if (p->f16) MPG_DW_ONE(0, true);
    else MPG_DW_ONE(0, false);
#undef MPG_DW_ONE
    constexpr int PER = H3 * H2 + H2 * H1 + H3 + H2;
    hipLaunchKernelGGL(edge_dw_reduce, dim3((PER + 31) / 32), dim3(256), 0, st, p->part, p->nwg, p->dscale, p->dW3,
                       p->dW2, p->db3, p->db2);
    return (int)hipGetLastError();
}
