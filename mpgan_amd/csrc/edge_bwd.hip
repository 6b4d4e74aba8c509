// Backward of the fused edge network (see edge.hip for the forward and the chain layout).
//
//   dZ3 = m_j * dagg_i * keep3 * phi'(Z3)      phi'(Z3) from the forward's saved sign words
//   dE2 = W3'^T dZ3 ;  dZ2 = dE2 * keep2 * phi'(Z2)     (Z2/E2 recomputed: one 90-MFMA layer)
//   dE1 = W2'^T dZ2 ;  dZ1 = dE1 * keep1 * phi'(Z1) ;  da_i = sum_j dZ1 ;  dc_j = sum_i dZ1
//   dW3 = s * sum_e dZ3 E2^T ;  dW2 = s * sum_e dZ2 E1^T ;  db3 = sum_e dZ3 ;  db2 = sum_e dZ2
//
// Two kernels.  edge_bwd_kernel walks the data-gradient chain exactly like the forward kernel (a wave
// per (jet, 32 receivers), senders in a loop, W3^T hi+lo resident in LDS, the rest streamed from L2) and, when
// weight gradients are wanted, parks E2 and dZ2 in memory as the 16-bit hi/lo B fragments the lanes hold (one
// coalesced 16-byte store per lane and fragment).  edge_dw_kernel then streams those fragments into
// [receiver][feature] LDS images, rebuilds the cheap operands (E1 from a_i + c_j, dZ3 from dagg and the sign
// words), fetches MFMA operands with transposing LDS reads and accumulates dW3/dW2/db3/db2 in registers over
// its share of the edges; the per-workgroup partials are summed by a last small kernel that also undoes the
// fragment-order permutation of the feature indices.
#include "edge_common.h"

#ifndef MPG_BEXP
#define MPG_BEXP 0  // experiment bits (tools/ubench/bwd_bench.hip): 1 streamed fragments all from index 0 (L1 hits), 2 no staging stores,
                    // 4 no dZ3 build, 8 no dZ2 epilogue, 16 no dZ1 epilogue, 32 no E2 epilogue
#endif
#ifndef MPG_DW_EXP
#define MPG_DW_EXP 0  // experiment bits (tools/ubench/dw_bench.hip): 1 consumers idle, 2 builders idle, 4 no staged loads, 8 no LDS writes
#endif

namespace {

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

// staging: block blk = (b*RB + rb)*N + j ; plane part (0 hi, 1 lo) ; the 10 B-operand fragments (tile, k-step) of
// the 160-feature tensor exactly as the lanes hold them: one coalesced 16-byte store per lane and fragment.
// In "fragment order" a lane's 8 elements are features fi = 16 frag + 8 h + jj of receiver r = lane & 31.
template <typename V>
MPG_DEV void stage_frag(void* base, size_t blk, int part, int frag, int lane, const V f) {
    reinterpret_cast<V*>(base)[((blk * 2 + part) * NFR2 + frag) * 64 + lane] = f;
}

template <int DROP, bool F16, bool NEEDW>  // DROP: 0 off, 1 byte mode, 2 bit mode (see common.h)
__global__ __launch_bounds__(256, 1) void edge_bwd_v1_kernel(const MpgEdgeBwd p) {
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
    for (int t = tid; t < H2; t += 256) lb2[t] = p.b2[t] * SC_E2;
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
        if (ii < p.N) v4 = ld4(p.a + (size_t)(b * p.N + ii) * (p.ld_ac ? p.ld_ac : H1) + 8 * qsu + 4 * hh);
        la[t] = make_float4(v4.x * SC_A, v4.y * SC_A, v4.z * SC_A, v4.w * SC_A);
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
    for (int t = tid; t < (j1 - j0) * (H1 / 4); t += 256) {
        const float4 c4 = ld4(p.c + (size_t)(b * p.N + j0 + t / (H1 / 4)) * (p.ld_ac ? p.ld_ac : H1) + 4 * (t % (H1 / 4)));
        reinterpret_cast<float4*>(lc)[t] = make_float4(c4.x * SC_A, c4.y * SC_A, c4.z * SC_A, c4.w * SC_A);
    }
    __syncthreads();
    for (int j = j0 + w; j < j1; j += 4) {
        const float mj = p.mask ? p.mask[b * p.N + j] : 1.f;
        if (mj == 0.f) {  // masked sender: every gradient through these edges is exactly zero (mpg_edge_dw skips the block too)
            if (lane < H1 / 4)
                reinterpret_cast<float4*>(p.dc + ((size_t)rb * p.B * p.N + (size_t)(b * p.N + j)) * H1)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
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
            for (int k = 0; k < PF; ++k) { rh[k] = img_frag<V>(r2, lane16, (MPG_BEXP & 1) ? 0 : k); rl[k] = img_frag<V>(r2, lane16, NF2 + ((MPG_BEXP & 1) ? 0 : k)); }
            f32x16 accs[2];
            float v2[16];
            auto epi2 = [&](int mm, int g) {
                if (MPG_BEXP & 32) { if (g == 1 || g == 3) { e2hi[mm][g >> 1] = e1hi[mm % T1][g >> 1]; e2lo[mm][g >> 1] = e1lo[mm % T1][g >> 1]; } return; }
                const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow, mm, 2 * g + h, h);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    v2[4 * g + t] = drop_apply<DROP>(lrelu(accs[mm & 1][4 * g + t], p.alpha), wd, 8 * g + t, t, p.thr);
                if (g == 1 || g == 3) {
                    const int s = g >> 1;
                    split8(v2 + 8 * s, e2hi[mm][s], e2lo[mm][s]);
                    if (NEEDW && !(MPG_BEXP & 2)) {
                        stage_frag(p.stageE2, blk, 0, mm * 2 + s, lane, e2hi[mm][s]);
                        stage_frag(p.stageE2, blk, 1, mm * 2 + s, lane, e2lo[mm][s]);
                    }
                }
            };
#pragma unroll
            for (int kk = 0; kk < TOT; ++kk) {
                const int m = kk / KS, k = kk % KS;
                if (kk + PF < TOT) {
                    rh[(kk + PF) % (PF + 1)] = img_frag<V>(r2, lane16, (MPG_BEXP & 1) ? 0 : kk + PF);
                    rl[(kk + PF) % (PF + 1)] = img_frag<V>(r2, lane16, NF2 + ((MPG_BEXP & 1) ? 0 : kk + PF));
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
        if (MPG_BEXP & 4) {
#pragma unroll
            for (int m = 0; m < T3; ++m)
#pragma unroll
                for (int s = 0; s < 2; ++s) { z3hi[m][s] = __builtin_bit_cast(bf16x8, e2hi[m % T2][s]); z3lo[m][s] = __builtin_bit_cast(bf16x8, e2lo[m % T2][s]); }
        } else {
            const int slope_pos = __builtin_bit_cast(int, mjs), slope_neg = __builtin_bit_cast(int, mjs * p.alpha);
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
                        // sign bit of this lane's Z3 register (tile m, 4g+t): see the forward's epi3.  The slope
                        // (alpha | 1) times m_j * dscale is one select between two wave-uniform constants.
                        const int neg = __builtin_amdgcn_sbfe((int)sw[m >> 1], 31 - (16 * (m & 1) + 4 * g + t), 1);
                        const float sel = dd[t] * __builtin_bit_cast(float, (neg & slope_neg) | (~neg & slope_pos));
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
                    if (MPG_BEXP & 8) { if (g == 1 || g == 3) { z2hi[mm][g >> 1] = z3hi[mm][g >> 1]; z2lo[mm][g >> 1] = z3lo[mm][g >> 1]; } return; }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int k = 4 * g + t;
                        // e2 of register k = element (k&7) of fragment k>>3, read off the packed 16-bit hi part:
                        // sign clear -> slope 1, set -> alpha; all-zero magnitude = dropped (dropout on) -> 0
                        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                        const unsigned int wd2 = __builtin_bit_cast(u32x4_t, e2hi[mm][k >> 3])[(k & 7) >> 1];
                        const int neg = __builtin_amdgcn_sbfe((int)wd2, 16 * (k & 1) + 15, 1);
                        float gt = __builtin_bit_cast(float, (neg & __builtin_bit_cast(int, p.alpha)) | (~neg & 0x3f800000));
                        if (DROP && __builtin_amdgcn_ubfe(wd2, 16 * (k & 1), 15) == 0u) gt = 0.f;
                        v2[k] = accs[(m + 1) & 1][k] * gt;
                    }
                    if (g == 1 || g == 3) {
                        const int s = g >> 1;
                        split8(v2 + 8 * s, z2hi[mm][s], z2lo[mm][s]);
                        if (NEEDW && !(MPG_BEXP & 2)) {
                            stage_frag(p.stageZ2, blk, 0, mm * 2 + s, lane, z2hi[mm][s]);
                            stage_frag(p.stageZ2, blk, 1, mm * 2 + s, lane, z2lo[mm][s]);
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
            for (int k = 0; k < PF; ++k) { rh[k] = img_frag<bf16x8>(r2t, lane16, (MPG_BEXP & 1) ? 0 : k); rl[k] = img_frag<bf16x8>(r2t, lane16, NF2T + ((MPG_BEXP & 1) ? 0 : k)); }
            f32x16 accs[2];
            // dc_j = sum over the 32 receivers (lanes of one half) of dZ1: 16 values per tile and lane.  Halving
            // reduction: at each step a lane keeps half of its values and hands the other half to its partner
            // (DPP), so 16 values cost 15 exchanges instead of 80 and lane l ends with the total of value l & 15
            // (= accumulator register 8s + 4u + t  <->  feature 32 mm + 16 s + 8 u + 4 h + t).
            const bool lb0 = lane & 1, lb1 = lane & 2, lb2 = lane & 4, lb3 = lane & 8;
            float ured[4];
            auto epi5 = [&](int mm, int su) {  // slice (s,u) of tile mm
                const int s = su >> 1, u = su & 1;
                if (MPG_BEXP & 16) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) dacc[mm][s][4 * u + t] += accs[mm & 1][8 * s + 4 * u + t];
                    return;
                }
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
                    rh[(kk + PF) % (PF + 1)] = img_frag<bf16x8>(r2t, lane16, (MPG_BEXP & 1) ? 0 : kk + PF);
                    rl[(kk + PF) % (PF + 1)] = img_frag<bf16x8>(r2t, lane16, NF2T + ((MPG_BEXP & 1) ? 0 : kk + PF));
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
// Weight gradients of fe.net.1 / fe.net.2:  dW3 = sum_e dZ3 E2^T,  dW2 = sum_e dZ2 E1^T  (+ bias sums).
// The contraction runs over edges = (receiver, sender): per block (32 receivers of one jet, one sender) the
// four operand tensors are needed as [feature][receiver], but the backward (and any rebuild) naturally
// produces [receiver][8 features] pieces.  So the block's operands are laid down in LDS as plain
// [receiver][feature] bf16 images and the MFMA fragments are fetched with gfx950's transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads 4 receivers x 16 features and each lane receives one feature's
// 4 receivers).  Row strides are odd multiples of 64 B, which makes those reads bank-conflict free.
//
// A workgroup is 8 waves on 4 SIMDs: waves 0-3 are CONSUMERS (each owns 10-12 of the 45 output tiles in
// registers and only issues LDS reads + MFMAs), waves 4-7 are BUILDERS (VALU only: copy the staged dZ2, convert
// the staged E2 from fp16 to bf16 hi/lo, rebuild dZ3 from dagg and the sign words and E1 from a_i + c_j, sum
// the biases).  The images are double buffered (2 x 80 KiB = all of the LDS): builders fill block n+1 while
// consumers multiply block n, one barrier per block.  Every builder thread owns fixed (receiver, feature
// chunk) pieces, keeps what it needs of dagg / a in registers and reloads each staged piece for the block
// after next right after using it, so no builder ever waits on another.
constexpr int DW_RS3 = 448, DW_RS2 = 320, DW_RS1 = 192;  // image row strides (bytes)
constexpr int DW_Z3H = 0, DW_Z3L = DW_Z3H + 32 * DW_RS3, DW_E2H = DW_Z3L + 32 * DW_RS3, DW_E2L = DW_E2H + 32 * DW_RS2,
              DW_Z2H = DW_E2L + 32 * DW_RS2, DW_Z2L = DW_Z2H + 32 * DW_RS2, DW_E1H = DW_Z2L + 32 * DW_RS2,
              DW_E1L = DW_E1H + 32 * DW_RS1, DW_BUF = DW_E1L + 32 * DW_RS1;
constexpr int DW_LDS_BYTES = 2 * DW_BUF;  // 163,840
static_assert(DW_LDS_BYTES <= 163840, "dW images must fit the LDS twice");

struct DwTile { int prod, m, n; };  // prod 0: dW3 (A = Z3 tile m, B = E2 tile n); 1: dW2 (A = Z2, B = E1)
// consumer wave w owns tiles [0,12) [12,23) [23,34) [34,45)
__device__ constexpr DwTile DW_TILES[45] = {
    {0,0,0},{0,0,1},{0,0,2},{0,0,3},{0,0,4},{0,1,0},{0,1,1},{0,1,2},{0,1,3},{0,1,4},{1,0,0},{1,0,1},
    {0,2,0},{0,2,1},{0,2,2},{0,2,3},{0,2,4},{0,3,0},{0,3,1},{0,3,2},{0,3,3},{0,3,4},{1,0,2},
    {0,4,0},{0,4,1},{0,4,2},{0,4,3},{0,4,4},{0,5,0},{0,5,1},{0,5,2},{0,5,3},{0,5,4},{1,1,0},
    {1,1,1},{1,1,2},{1,2,0},{1,2,1},{1,2,2},{1,3,0},{1,3,1},{1,3,2},{1,4,0},{1,4,1},{1,4,2}};

constexpr bool dw_same_rows(int t, int u) { return DW_TILES[t].prod == DW_TILES[u].prod && DW_TILES[t].m == DW_TILES[u].m; }
constexpr bool dw_leader(int t, int begin) { return t == begin || !dw_same_rows(t, t - 1); }
constexpr int dw_group_end(int t, int end) {
    int e = t + 1;
    while (e < end && dw_same_rows(t, e)) ++e;
    return e;
}

MPG_DEV int feat_of_fi(int fi) {  // fragment-order index -> feature
    const int ms = fi >> 4, hh = (fi >> 3) & 1, jj = fi & 7;
    return 32 * (ms >> 1) + 16 * (ms & 1) + 8 * (jj >> 2) + 4 * hh + (jj & 3);
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// fragment (32 features starting at byte column `col`, receivers 16s .. 16s+15) of an image: two transposed reads
MPG_DEV bf16x8 dw_frag(uint32_t lane_addr, int off, int rs) {
    typedef __attribute__((address_space(3))) s16x4* P;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P)(lane_addr + (uint32_t)off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P)(lane_addr + (uint32_t)(off + 4 * rs)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int BEGIN, int END>
MPG_DEV void dw_consume(f32x16* acc, uint32_t buf, int lane) {
    // lane 4q+p of 16-lane group g supplies row (8 (g>>1) + q), feature columns 16 (g&1) + 4p .. +3 of the block
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = 8 * (g >> 1) + q, col = (16 * (g & 1) + 4 * pp) * 2;
    // one opaque base per image family, everything else is an immediate offset
    uint32_t bz3 = buf + DW_Z3H + row * DW_RS3 + col, be2 = buf + DW_E2H + row * DW_RS2 + col;
    uint32_t bz2 = buf + DW_Z2H + row * DW_RS2 + col, be1 = buf + DW_E1H + row * DW_RS1 + col;
    asm volatile("" : "+v"(bz3), "+v"(be2), "+v"(bz2), "+v"(be1));
    // tiles sharing their A rows (same product and m) form a group: per k-step the A fragments are read once
    // and only the B fragments change -- 8 + 8 fragment registers live beside the 160-192 accumulators
    static_for<BEGIN, END>([&](auto tc) {
        MPG_CI(t, tc);
        if constexpr (dw_leader(t, BEGIN)) {
            constexpr DwTile d = DW_TILES[t];
            constexpr int ge = dw_group_end(t, END);
            constexpr int rsa = d.prod == 0 ? DW_RS3 : DW_RS2, rsb = d.prod == 0 ? DW_RS2 : DW_RS1;
            constexpr int alo = d.prod == 0 ? DW_Z3L - DW_Z3H : DW_Z2L - DW_Z2H, blo = d.prod == 0 ? DW_E2L - DW_E2H : DW_E1L - DW_E1H;
            uint32_t ba = d.prod == 0 ? bz3 : bz2, bb = d.prod == 0 ? be2 : be1;
            // a fresh (opaque) base per group: otherwise the B fragments of a whole product (80 registers) are kept
            // for the next group and the accumulators spill
            asm volatile("" : "+v"(ba), "+v"(bb));
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 ah = dw_frag(ba, 16 * s * rsa + 64 * d.m, rsa), al = dw_frag(ba, alo + 16 * s * rsa + 64 * d.m, rsa);
                static_for<t, ge>([&](auto uc) {
                    MPG_CI(u, uc);
                    constexpr int n = DW_TILES[u].n;
                    const bf16x8 bh = dw_frag(bb, 16 * s * rsb + 64 * n, rsb), bl = dw_frag(bb, blo + 16 * s * rsb + 64 * n, rsb);
                    acc[u - BEGIN] = mfma3(ah, al, bh, bl, acc[u - BEGIN]);
                });
            }
        }
    });
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

// A workgroup walks at most 64 blocks (the launcher sizes the grid for that); their valid-sender bits are one
// ballot taken at kernel start, so stepping to the next unmasked block is pure scalar arithmetic -- no memory
// access and no loop inside the pipelined loops (either would make the compiler drain vmcnt there).
// The blocks of a workgroup are STRIDED over the launch (slot t of workgroup g is block g + t * gridDim.x): a contiguous
// range would be one jet's senders, and a launch would last as long as its fullest jet (masks sorted to the end of a
// jet left whole workgroups idle at N = 150); strided, every workgroup samples all jets.
MPG_DEV int dw_block(int t) { return (int)blockIdx.x + t * (int)gridDim.x; }
MPG_DEV unsigned long long dw_valid_bits(const MpgEdgeDw& p, int blk0, int blk1) {   // over the slots [blk0, blk1)
    const int RB = (p.N + 31) / 32, t = blk0 + (int)(threadIdx.x & 63), x = dw_block(t);
    bool ok = t < blk1;
    if (ok && p.mask != nullptr) ok = p.mask[((x / p.N) / RB) * p.N + x % p.N] != 0.f;
    return __ballot(ok);
}
MPG_DEV int dw_next_valid(unsigned long long bits, int blk0, int blk, int blk1) {
    const int d = blk - blk0;
    const unsigned long long rem = d < 64 ? bits >> d : 0ull;
    return rem ? blk + __builtin_ctzll(rem) : blk1;
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load
// (vmcnt(0)), which would expose the latency of the staged pieces requested a block ahead.
MPG_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The per-workgroup loop of one consumer wave (its output tiles BEGIN..END of DW_TILES).  Instantiated per
// role: a run-time branch around the MFMA section would make the accumulators merge at every join.
template <int BEGIN, int END>
MPG_DEV void dw_consumer(const MpgEdgeDw& p, int blk0, int blk1, unsigned long long vbits) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[END - BEGIN];
#pragma unroll
    for (int t = 0; t < END - BEGIN; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    int cur = dw_next_valid(vbits, blk0, blk0, blk1), it = 0;
    lds_barrier();  // block `cur` is in buffer 0
    while (cur < blk1) {
        if (!(MPG_DW_EXP & 1)) dw_consume<BEGIN, END>(acc, lds0 + (it & 1) * DW_BUF, lane);
        lds_barrier();
        cur = dw_next_valid(vbits, blk0, cur + 1, blk1);
        ++it;
    }
    float* part = p.part + (size_t)blockIdx.x * (H3 * H2 + H2 * H1 + H3 + H2);
    dw_store<BEGIN, END>(acc, part, part + H3 * H2, lane);
}

// keep bits (bit k = element k) of one 8-feature chunk: features 32 tile + f0 + {0..3, 8..11} of edge row `erow`
template <int DM>
MPG_DEV uint32_t dw_chunk_keep(uint32_t seed_lo, uint32_t seed_hi, uint32_t tag, uint32_t erow, int tile, int f0, uint32_t thr) {
    if constexpr (DM == 2) {
        const uint32_t w = drop_word(seed_lo, seed_hi, tag, erow, DROP_BIT_GRP + (uint32_t)tile) >> f0;
        return (w & 0xfu) | ((w >> 4) & 0xf0u);  // bits f0..f0+3 and f0+8..f0+11
    } else if constexpr (DM == 1) {
        uint32_t m = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t w = drop_word(seed_lo, seed_hi, tag, erow, (uint32_t)(8 * tile + (f0 >> 2) + 2 * u));
#pragma unroll
            for (int t = 0; t < 4; ++t) m |= (drop_keep(w, t, thr) ? 1u : 0u) << (4 * u + t);
        }
        return m;
    } else {
        return 0xffu;
    }
}

template <int DROP, bool F16>
MPG_DEV void dw_builder(const MpgEdgeDw& p, char* smem, int blk0, int blk1, unsigned long long vbits) {
    typedef typename FragT<F16>::type E2V;
    const int bt = threadIdx.x - 256;     // builder thread 0..255
    const int r = bt & 31, cg = bt >> 5;  // receiver row of the images, chunk group: chunks cg, cg + 8, cg + 16
    const int RB = (p.N + 31) / 32;
    // chunk c = 2 frag + h holds fragment-order features 8c .. 8c+7 = registers 8s .. 8s+7 of tile (c >> 2) of
    // lane (r, h):  s = (c >> 1) & 1 and h = c & 1 are the same for all chunks of this thread
    const int cs = (cg >> 1) & 1, ch = cg & 1;
    const int f0 = 16 * cs + 4 * ch;  // features of chunk c: 32 (c >> 2) + f0 + {0..3, 8..11}

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    float dreg[3][8], areg[2][8];   // dagg / a of this thread's Z3 / E1 chunks, current jet (raw)
    float db3[3][8], db2[3][8];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) { db3[n][k] = 0.f; db2[n][k] = 0.f; }
    uint32_t sw[3];                 // sign words of the Z3 chunks' lanes
    float4 cv[2][2];                // c_j of the E1 chunks
    E2V eh[3], el[3];               // staged E2 pieces (hi, lo)
    bf16x8 zh[3], zl[3];            // staged dZ2 pieces

    const bool third = cg < 4;  // chunk groups 0..3 own a third 160-feature piece and a second E1 chunk
    auto e1tile = [&](int n) { return n == 0 || third ? 2 * n + (cg >> 2) : (cg >> 2); };  // 0..2
    // Threads of chunk groups 4..7 have no third piece of the 160-feature tensors (and no second E1 chunk): they
    // redo their previous piece instead (same data to the same place), which keeps the whole build free of
    // branches -- inside a branch the compiler waits for ALL outstanding loads, i.e. for the prefetches too.
    auto chunk160 = [&](int n) { return n < 2 || third ? cg + 8 * n : cg + 8; };

    // Every global read is a raw buffer load: resource in SGPRs, block-dependent part as scalar offset, one
    // thread-constant VGPR offset per stream (plain pointers cost two VGPRs of address per load in flight).
    const int nblk = p.B * RB * p.N;
    const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageE2), 0, nblk * (2 * NFR2 * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageZ2), 0, nblk * (2 * NFR2 * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(p.sign3), 0, nblk * (T3 * 32 * 4), 0x00020000);
    const int ldac = p.ld_ac ? p.ld_ac : H1;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c), 0, p.B * p.N * ldac * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.B * p.N * ldac * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dagg), 0, p.B * p.N * p.ld_dagg * 4, 0x00020000);
    int vo160[3];  // byte offset of this thread's piece n inside a 10 KiB plane
#pragma unroll
    for (int n = 0; n < 3; ++n) vo160[n] = (chunk160(n) * 32 + r) * 16;
    const int voS = (32 * ch + r) * 4;
    int voE1[2];   // byte offset of E1 chunk n's first feature inside a 96-float row of a / c
#pragma unroll
    for (int n = 0; n < 2; ++n) voE1[n] = (32 * e1tile(n) + f0) * 4;
    auto ldb4 = [&](__amdgpu_buffer_rsrc_t rs, int vo, int so) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
    };

    // dagg / a of the (jet, receiver block) of block `blk`, RAW: nothing may be computed from a prefetched value
    // before the block that needs it (a use right behind the load would make every iteration wait for its
    // youngest load, i.e. drain the whole prefetch queue).  Padding receivers read row 0 and get scale 0.
    float dscl = 0.f;  // agg_scale * dscale, or 0 for a padding receiver -- belongs to the jet in dreg
    auto load_jet = [&](int blk) {
        const int brb = blk / p.N, rb = brb % RB, b = brb / RB, ii = rb * 32 + r;
        const bool ok = ii < p.N;
        dscl = ok ? p.agg_scale * p.dscale : 0.f;
        const int rowD = (ok ? ii : 0) * p.ld_dagg * 4 + (32 * (cg >> 2) + f0) * 4, soD = b * p.N * p.ld_dagg * 4;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const float4 u = ldb4(rD, rowD + 256 * n, soD), v = ldb4(rD, rowD + 256 * n + 32, soD);
            dreg[n][0] = u.x; dreg[n][1] = u.y; dreg[n][2] = u.z; dreg[n][3] = u.w;
            dreg[n][4] = v.x; dreg[n][5] = v.y; dreg[n][6] = v.z; dreg[n][7] = v.w;
        }
        const int rowA = (ok ? ii : 0) * ldac * 4, soA = b * p.N * ldac * 4;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float4 u = ldb4(rA, rowA + voE1[n], soA), v = ldb4(rA, rowA + voE1[n] + 32, soA);
            areg[n][0] = u.x; areg[n][1] = u.y; areg[n][2] = u.z; areg[n][3] = u.w;
            areg[n][4] = v.x; areg[n][5] = v.y; areg[n][6] = v.z; areg[n][7] = v.w;
        }
    };
    unsigned int nbw = 0xffffffffu;  // this receiver's neighbour word holding the block's sender (k-NN graphs)
    const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(p.nbr), 0, p.nbr ? p.B * p.N * ((p.N + 31) >> 5) * 4 : 0, 0x00020000);
    auto load_nb = [&](int blk) {  // (with no graph the resource is empty: the load returns 0 and is ignored below)
        const int j = blk % p.N, brb = blk / p.N, rb = brb % RB, b = brb / RB, ii = rb * 32 + r;
        nbw = __builtin_amdgcn_raw_buffer_load_b32(rN, ((b * p.N + (ii < p.N ? ii : 0)) * ((p.N + 31) >> 5) + (j >> 5)) * 4, 0, 0);
    };
    auto load_sw = [&](int blk) {  // word (tile >> 1) = n of lane (r, h)
#pragma unroll
        for (int n = 0; n < 3; ++n) sw[n] = __builtin_amdgcn_raw_buffer_load_b32(rS, voS, blk * (T3 * 32 * 4) + n * 256, 0);
    };
    auto load_c = [&](int blk) {
        const int j = blk % p.N, b = (blk / p.N) / RB, so = (b * p.N + j) * ldac * 4;
#pragma unroll
        for (int n = 0; n < 2; ++n) { cv[n][0] = ldb4(rC, voE1[n], so); cv[n][1] = ldb4(rC, voE1[n] + 32, so); }
    };
    // staged pieces: chunk c of receiver r is element c * 32 + r of a plane of 640 16-byte pieces
    auto load_e2 = [&](int blk, int n) {
        eh[n] = __builtin_bit_cast(E2V, __builtin_amdgcn_raw_buffer_load_b128(rE, vo160[n], blk * (2 * NFR2 * 1024), 0));
        el[n] = __builtin_bit_cast(E2V, __builtin_amdgcn_raw_buffer_load_b128(rE, vo160[n], blk * (2 * NFR2 * 1024) + NFR2 * 1024, 0));
    };
    auto load_z2 = [&](int blk, int n) {
        zh[n] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rZ, vo160[n], blk * (2 * NFR2 * 1024), 0));
        zl[n] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rZ, vo160[n], blk * (2 * NFR2 * 1024) + NFR2 * 1024, 0));
    };

    // build block `blk` into buffer `buf`; right after a staged piece is used, request the one of `pre`
    auto build = [&](int blk, char* buf, int pre_) {
        const bool exp_noload = (MPG_DW_EXP & 4) && p.N != 12345, exp_nowrite = (MPG_DW_EXP & 8) && p.N != 12345;
        const int pre = (MPG_DW_EXP & 16) ? blk0 : pre_;
        const int j = blk % p.N, brb = blk / p.N, rb = brb % RB, b = brb / RB, ii = rb * 32 + r;
        const uint32_t erow = (uint32_t)((b * p.N + ii) * p.N + j);
        // dZ2: copy, bias sums
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int c = chunk160(n);
            if (!exp_nowrite) {
                *reinterpret_cast<bf16x8*>(buf + DW_Z2H + r * DW_RS2 + c * 16) = zh[n];
                *reinterpret_cast<bf16x8*>(buf + DW_Z2L + r * DW_RS2 + c * 16) = zl[n];
            }
            const float take = n < 2 || third ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) db2[n][k] += take * ((float)zh[n][k] + (float)zl[n][k]);
            if (!exp_noload) load_z2(pre, n);
        }
        // E2: staged fp16 (or bf16) hi/lo -> bf16 hi/lo
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int c = chunk160(n);
            bf16x8 hh, ll;
            if constexpr (F16) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (float)eh[n][k] + (float)el[n][k];
                split8(v, hh, ll);
            } else {
                hh = eh[n]; ll = el[n];
            }
            if (!exp_nowrite) {
                *reinterpret_cast<bf16x8*>(buf + DW_E2H + r * DW_RS2 + c * 16) = hh;
                *reinterpret_cast<bf16x8*>(buf + DW_E2L + r * DW_RS2 + c * 16) = ll;
            } else if (hh[0] == (__bf16)123.f) db2[0][0] += (float)ll[1];
            if (!exp_noload) load_e2(pre, n);
        }
        // dZ3 = dagg * slope(sign bit) * keep3
        const float in_set = (p.nbr == nullptr || ((nbw >> (j & 31)) & 1u)) ? 1.f : 0.f;
        const float dscl_1 = dscl * in_set, dscl_a = dscl * p.alpha * in_set;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int m = 2 * n + (cg >> 2), c = cg + 8 * n;
            float v[8];
            const uint32_t keep = dw_chunk_keep<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, m, f0, p.thr);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t neg = (sw[n] >> (31 - (16 * (m & 1) + 8 * cs + k))) & 1u;
                float x = dreg[n][k] * (neg ? dscl_a : dscl_1);
                if (DROP && !((keep >> k) & 1u)) x = 0.f;
                v[k] = x;
                db3[n][k] += x;
            }
            bf16x8 hh, ll;
            split8(v, hh, ll);
            if (!exp_nowrite) {
                *reinterpret_cast<bf16x8*>(buf + DW_Z3H + r * DW_RS3 + c * 16) = hh;
                *reinterpret_cast<bf16x8*>(buf + DW_Z3L + r * DW_RS3 + c * 16) = ll;
            } else if (hh[0] == (__bf16)123.f) db2[0][0] += (float)ll[1];
        }
        if (!exp_noload) { load_sw(pre); load_nb(pre); }
        // E1 = keep1 * lrelu(a_i + c_j)   (chunk groups 4..7: the second chunk repeats the first)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int q = e1tile(n), c = 4 * q + (cg & 3);
            const float cc[8] = {cv[n][0].x, cv[n][0].y, cv[n][0].z, cv[n][0].w, cv[n][1].x, cv[n][1].y, cv[n][1].z, cv[n][1].w};
            float v[8];
            const uint32_t keep = dw_chunk_keep<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, q, f0, p.thr);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float x = lrelu(areg[n][k] + cc[k], p.alpha);
                if (DROP && !((keep >> k) & 1u)) x = 0.f;
                v[k] = x;
            }
            bf16x8 hh, ll;
            split8(v, hh, ll);
            if (!exp_nowrite) {
                *reinterpret_cast<bf16x8*>(buf + DW_E1H + r * DW_RS1 + c * 16) = hh;
                *reinterpret_cast<bf16x8*>(buf + DW_E1L + r * DW_RS1 + c * 16) = ll;
            } else if (hh[0] == (__bf16)123.f) db2[0][0] += (float)ll[1];
        }
        if (!exp_noload) { load_c(pre); load_jet(pre); }
    };

    int cur = dw_next_valid(vbits, blk0, blk0, blk1), it = 0;
    int nxt = dw_next_valid(vbits, blk0, cur + 1, blk1);
    // `pre` is clamped to the last block of the range: past the end the prefetches fetch that block again, unused
    if (cur < blk1) {
        const int b0 = dw_block(cur);
        load_jet(b0); load_sw(b0); load_nb(b0); load_c(b0);
#pragma unroll
        for (int n = 0; n < 3; ++n) { load_e2(b0, n); load_z2(b0, n); }
        build(b0, smem, dw_block(min(nxt, blk1 - 1)));
    }
    lds_barrier();
    while (cur < blk1) {
        const int nxt2 = dw_next_valid(vbits, blk0, nxt + 1, blk1);
        if (nxt < blk1 && !(MPG_DW_EXP & 2)) build(dw_block(nxt), smem + ((it + 1) & 1) * DW_BUF, dw_block(min(nxt2, blk1 - 1)));
        lds_barrier();
        cur = nxt;
        nxt = nxt2;
        ++it;
    }

    // bias sums: add the 32 receivers (one half-wave per chunk group), fragment-order index fi = 8 c + k
    float* part = p.part + (size_t)blockIdx.x * (H3 * H2 + H2 * H1 + H3 + H2);
    float* pb3 = part + H3 * H2 + H2 * H1, *pb2 = pb3 + H3;
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float x = db3[n][k], y = db2[n][k];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
            const int c = cg + 8 * n;
            if (r == 0) {
                pb3[8 * c + k] = x;
                if (c < 2 * NFR2) pb2[8 * c + k] = y;  // (the repeated third piece added zeros)
            }
        }
}

template <int DROP, bool F16>
__global__ __launch_bounds__(512, 1) void edge_dw_kernel(const MpgEdgeDw p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RB = (p.N + 31) / 32;
    const int nblk = p.B * RB * p.N;
    const int blk0 = 0, blk1 = (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // slots of this workgroup
    const unsigned long long vbits = dw_valid_bits(p, blk0, blk1);
    if (w == 0) dw_consumer<0, 12>(p, blk0, blk1, vbits);
    else if (w == 1) dw_consumer<12, 23>(p, blk0, blk1, vbits);
    else if (w == 2) dw_consumer<23, 34>(p, blk0, blk1, vbits);
    else if (w == 3) dw_consumer<34, 45>(p, blk0, blk1, vbits);
    else dw_builder<DROP, F16>(p, smem, blk0, blk1, vbits);
}

// out = scale * sum over workgroup partials, feature indices mapped back from fragment order.
// 32 outputs x 8 partial-slices per block: the 256 partials of an output are read by 8 threads.
__global__ __launch_bounds__(256) void edge_dw_reduce(const float* __restrict__ part, int nwg, float scale3, float scale, int accumulate,
                                                      float* __restrict__ dW3, float* __restrict__ dW2,
                                                      float* __restrict__ db3, float* __restrict__ db2) {
    __shared__ float red[8][32];
    constexpr int PER = H3 * H2 + H2 * H1 + H3 + H2;
    const int ix = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + ix;
    float s = 0.f;
    if (idx < PER) {   // (four independent partial sums: a thread's loads are in flight together; fixed summation order)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int g = sl;
        for (; g + 24 < nwg; g += 32) {
            s0 += part[(size_t)g * PER + idx]; s1 += part[(size_t)(g + 8) * PER + idx];
            s2 += part[(size_t)(g + 16) * PER + idx]; s3 += part[(size_t)(g + 24) * PER + idx];
        }
        for (; g < nwg; g += 8) s0 += part[(size_t)g * PER + idx];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][ix] = s;
    __syncthreads();
    if (sl != 0 || idx >= PER) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][ix];
    if (idx < H3 * H2) {
        const int r3 = idx / H2, c2 = idx % H2;
        float* d = dW3 + feat_of_fi(r3) * H2 + feat_of_fi(c2);
        *d = s * scale3 + (accumulate ? *d : 0.f);
    } else if (idx < H3 * H2 + H2 * H1) {
        const int k = idx - H3 * H2, r2 = k / H1, c1 = k % H1;
        float* d = dW2 + feat_of_fi(r2) * H1 + feat_of_fi(c1);
        *d = s * scale + (accumulate ? *d : 0.f);
    } else if (idx < H3 * H2 + H2 * H1 + H3) {
        float* d = db3 + feat_of_fi(idx - H3 * H2 - H2 * H1);
        *d = s + (accumulate ? *d : 0.f);
    } else {
        float* d = db2 + feat_of_fi(idx - H3 * H2 - H2 * H1 - H3);
        *d = s + (accumulate ? *d : 0.f);
    }
}

}  // namespace

// first-generation data-gradient kernel (one sender per wave and pass): kept for A/B timing in tools/ubench only
extern "C" int mpg_edge_bwd_v1(const MpgEdgeBwd* p, void* stream) {
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
        MPG_ENSURE_LDS((edge_bwd_v1_kernel<D, H, W>), BWD_LDS_BYTES);                                                \
        hipLaunchKernelGGL((edge_bwd_v1_kernel<D, H, W>), grid, block, BWD_LDS_BYTES, st, *p);                       \
    } while (0)
#define MPG_BWD_W(D, H)                                                                                           \
    do { if (needw) MPG_BWD_ONE(D, H, true); else MPG_BWD_ONE(D, H, false); } while (0)
#define MPG_BWD_H(D)                                                                                              \
    do { if (p->f16) MPG_BWD_W(D, true); else MPG_BWD_W(D, false); } while (0)
#ifdef MPG_SINGLE_VARIANT
    MPG_BWD_W(MPG_SINGLE_VARIANT, true);
#else
    if (dm == 0) MPG_BWD_H(0);
    else if (dm == 1) MPG_BWD_H(1);
    else MPG_BWD_H(2);
#endif
#undef MPG_BWD_H
#undef MPG_BWD_W
#undef MPG_BWD_ONE
    return (int)hipGetLastError();
}

extern "C" int mpg_edge_dw(const MpgEdgeDw* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->nwg <= 0) return -1;
    {
        const int nblk = p->B * ((p->N + 31) / 32) * p->N;
        if ((nblk + p->nwg - 1) / p->nwg > 64) return -5;  // a workgroup walks at most 64 blocks (one ballot of valid bits)
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p->nwg), block(512);
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
#define MPG_DW_ONE(D, H)                                                                                          \
    do {                                                                                                          \
        MPG_ENSURE_LDS((edge_dw_kernel<D, H>), DW_LDS_BYTES);                                                     \
        hipLaunchKernelGGL((edge_dw_kernel<D, H>), grid, block, DW_LDS_BYTES, st, *p);                            \
    } while (0)
#define MPG_DW_H(D) do { if (p->f16) MPG_DW_ONE(D, true); else MPG_DW_ONE(D, false); } while (0)
#ifdef MPG_SINGLE_VARIANT  // tools/ubench/dw_bench.hip: one instantiation
    MPG_DW_ONE(MPG_SINGLE_VARIANT, true);
#else
    if (dm == 0) MPG_DW_H(0);
    else if (dm == 1) MPG_DW_H(1);
    else MPG_DW_H(2);
#endif
#undef MPG_DW_H
#undef MPG_DW_ONE
    constexpr int PER = H3 * H2 + H2 * H1 + H3 + H2;
    // (the staged E2 carries the forward's operand scale SC_E2)
    hipLaunchKernelGGL(edge_dw_reduce, dim3((PER + 31) / 32), dim3(256), 0, st, p->part, p->nwg, p->dscale / SC_E2, p->dscale, p->accumulate, p->dW3,
                       p->dW2, p->db3, p->db2);
    return (int)hipGetLastError();
}
