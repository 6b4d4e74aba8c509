// NOT PART OF THE PRODUCT LIBRARY (kept as a measured experiment; harness: w2_bench.hip).
// Grouped weight gradients dW[n, k] = sum_m dy[m, n] x[m, k]:
// one-term fp16 products, fp32 accumulation, split over the rows m.
//
// Why not the generic 64x64x32 split-K GEMM of gemm.hip: there both operands are k-major, so every thread fetches them
// dword by dword, splits 16 values into hi/lo planes and meets two barriers per six MFMAs; an output strip is re-read
// four to five times (134 MB per launch) and the launch takes 35 us for 6 GFLOP.  Here
//   * a workgroup owns a 128 x 128 tile of one dW for a slice of the rows; its four waves each hold a 64 x 64 quarter
//     (2 x 2 MFMA tiles) in registers;
//   * the rows arrive as they lie in memory (feature-contiguous, float4 per thread, two whole rows per wave instruction),
//     are rounded ONCE to fp16 and laid down as plain [row][feature] images; the MFMA fragments (8 consecutive rows of
//     one feature per lane) come out of gfx950's transposing LDS read, exactly as in edge_dw.hip;
//   * gradients have no natural scale, fp16 has 30 binades: the dy rows of a CHUNK (4 stages = 128 rows) are fetched
//     into registers first, their largest magnitude decides the chunk's power of two (max lands in [2^13, 2^14)), and
//     the accumulators -- fp32, so this is exact -- are rescaled when the unit changes from one chunk to the next.
//     Nothing about the magnitude of a gradient is assumed and no other kernel has to report one;
//   * the bias gradient (column sums of dy) is added up from the fp32 values the threads hold anyway.
// Products are dy (11 bits) x x (11 bits): the contraction runs over thousands of rows with independent roundings
// (measured errors: DESIGN.md section 2).  x is an activation (|x| < 65504 as everywhere on the fp16 forward path).
#include "../../mpgan_amd/csrc/common.h"
#include "../../mpgan_amd/csrc/gemm.h"
#include "wgrad2.h"

#ifndef MPG_W2_EXP   // experiments (tools/ubench/w2_bench.hip): 1 no MFMAs, 2 no LDS traffic, 4 no global loads, 8 no partial stores
#define MPG_W2_EXP 0
#endif

namespace {

constexpr int W2_ROWS = 32;                    // rows of the contraction per stage (two MFMA k-steps)
constexpr int W2_RS = 320;                     // image row stride in bytes: 128 fp16 + 64 (odd multiple of 64 B)
constexpr int W2_IMG = W2_ROWS * W2_RS;        // 10,240 per operand
constexpr int W2_STAGE = 2 * W2_IMG;           // dy image, x image
constexpr int W2_NBUF = 2;
constexpr int W2_CHUNK = 4;                    // stages per chunk (dy rows held in registers: 16 float4)
constexpr int W2_RED = W2_NBUF * W2_STAGE;     // scratch: 4 wave maxima; later the column sums [8][128]
constexpr int W2_LDS = W2_RED + 8 * 128 * 4;

typedef unsigned int w2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int w2_u32x2 __attribute__((ext_vector_type(2)));
typedef short w2_s16x4 __attribute__((ext_vector_type(4)));
typedef short w2_s16x8 __attribute__((ext_vector_type(8)));

MPG_DEV void w2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fragment (32 features from byte column `off` of the lane's base, rows 16s .. 16s+15 folded into `off`): two transposing reads
MPG_DEV f16x8 w2_frag(uint32_t lane_addr, int off) {
    typedef __attribute__((address_space(3))) w2_s16x4* P;
    const w2_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P)(uintptr_t)(lane_addr + (uint32_t)off));
    const w2_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P)(uintptr_t)(lane_addr + (uint32_t)(off + 4 * W2_RS)));
    const w2_s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}

MPG_DEV uint32_t w2_pk(float a, float b) {   // two floats -> packed fp16 (round to nearest even)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}

__global__ __launch_bounds__(256, 1) void wgrad2_kernel(const W2Group G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int q = 0;
    while ((int)blockIdx.x >= G.wg0[q + 1]) ++q;
    const W2Job& J = G.j[q];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int local = blockIdx.x - G.wg0[q];
    const int tn = (J.N + 127) / 128, tk = (J.K + 127) / 128;
    const int n0 = 128 * (local % tn), k0 = 128 * ((local / tn) % tk), z = local / (tn * tk);
    const int nst = (J.M + W2_ROWS - 1) / W2_ROWS, per = (nst + G.splitk[q] - 1) / G.splitk[q];
    const int st0 = z * per, st1 = min(nst, st0 + per);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    float* const red = reinterpret_cast<float*>(smem + W2_RED);

    // staging role: float4 column group cg of rows r0 + 8j (j < 4) of a stage, for both operands
    const int cg = tid & 31, r0 = tid >> 5;
    const int ncol = n0 + 4 * cg, kcol = k0 + 4 * cg;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.dy), 0, (int)((size_t)J.M * J.ldy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(J.x), 0, (int)((size_t)J.M * J.ldx * 4), 0x00020000);
    // (rows past M and column groups past the operand's width read through an out-of-range offset: zeros; a group that
    // straddles the width -- K not a multiple of 4 -- stays inside its row, the launcher checks ld, and is masked below)
    const bool ngrp = ncol < J.N, kgrp = kcol < J.K;
    // (32-bit offsets -- the launcher bounds the operands' sizes -- and the selects kept as selects: with a 64-bit product
    // behind them the compiler turns every one into a branch around its load)
    const int rlim = min(J.M, st1 * W2_ROWS), ldy4 = J.ldy * 4, ldx4 = J.ldx * 4, ncol4 = ncol * 4, kcol4 = kcol * 4;
    auto dy_off = [&](int row) { const int o = row * ldy4 + ncol4; return (ngrp && row < rlim) ? o : 0x7ffffff0; };
    auto x_off = [&](int row) { const int o = row * ldx4 + kcol4; return (kgrp && row < rlim) ? o : 0x7ffffff0; };
    bool nel[4], kel[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { nel[e] = ncol + e < J.N; kel[e] = kcol + e < J.K; }

    // fragment role: lane 4q+p of 16-lane group g addresses row 8 (g >> 1) + q, features 16 (g & 1) + 4p .. +3 of a tile
    const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
    const uint32_t frag0 = (uint32_t)((8 * (fg >> 1) + fq) * W2_RS + (16 * (fg & 1) + 4 * fp) * 2);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    int ecur = 0;          // the accumulators hold 2^ecur * (sum so far)
    bool first = true;
    int it = 0;            // stages done: LDS buffer = it & 1

    // A chunk's rows of both operands are requested a whole chunk ahead, into the register set the chunk before it has
    // just finished with (vector memory completes in order: nothing younger is ever waited for behind them).
    f32x4 dyv[2][W2_CHUNK][4], xv[2][W2_CHUNK][4];
    auto request = [&](auto bc, int c0) {
        MPG_CI(b, bc);
        const int rbase = (MPG_W2_EXP & 4) ? 0x40000000 : c0 * W2_ROWS + r0;   // (rows past the slice read zeros: dy_off, x_off)
        if (J.dy_vec) {
            static_for<0, W2_CHUNK>([&](auto sc) {
                MPG_CI(s, sc);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dyv[b][s][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, dy_off(rbase + s * W2_ROWS + 8 * j), 0, 0));
            });
        } else {   // rows that are not float4 groups (a generator's 3 output features): element by element
            static_for<0, W2_CHUNK>([&](auto sc) {
                MPG_CI(s, sc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = rbase + s * W2_ROWS + 8 * j, o = row * ldy4 + ncol4;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dyv[b][s][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rdy, (row < rlim && nel[e]) ? o + 4 * e : 0x7ffffff0, 0, 0));
                }
            });
        }
        static_for<0, W2_CHUNK>([&](auto sc) {
            MPG_CI(s, sc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xv[b][s][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, x_off(rbase + s * W2_ROWS + 8 * j), 0, 0));
        });
    };
    auto chunk = [&](auto bc, int c0) {
        MPG_CI(b, bc);
        const int ns = min(W2_CHUNK, st1 - c0);
        // ---- the chunk's unit
        float mx = 0.f;
        static_for<0, W2_CHUNK>([&](auto sc) {
            MPG_CI(s, sc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = nel[e] ? dyv[b][s][j][e] : 0.f;
                    dyv[b][s][j][e] = v;
                    mx = fmaxf(mx, fabsf(v));
                }
        });
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) red[w] = mx;
        w2_barrier();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        // exponent e with mx * 2^e in [2^13, 2^14), from the bits (mx = 0, inf or nan: unit 1)
        const uint32_t mb = __builtin_bit_cast(uint32_t, mx) >> 23;
        int e = (mb == 0u || mb >= 255u) ? 0 : (127 + 13) - (int)mb;
        e = max(-100, min(100, e));
        e = __builtin_amdgcn_readfirstlane(e);
        const float unit = __builtin_bit_cast(float, (uint32_t)(127 + e) << 23);
        if (!first && e != ecur) {
            const float rs = __builtin_bit_cast(float, (uint32_t)(127 + max(-126, min(127, e - ecur))) << 23);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][bb][i] *= rs;
        }
        ecur = e;
        first = false;
        // ---- the next chunk's rows (everything this chunk still needs is in registers: no wait below this point)
        request(std::integral_constant<int, 1 - b>{}, c0 + W2_CHUNK);

        // ---- stages: images of stage s, barrier, fragments + MFMAs
        static_for<0, W2_CHUNK>([&](auto sc) {
            MPG_CI(s, sc);
            if (s < ns) {
                const uint32_t buf = lds0 + (uint32_t)((it & 1) * W2_STAGE);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 d = dyv[b][s][j];
                    bsum[0] += d[0]; bsum[1] += d[1]; bsum[2] += d[2]; bsum[3] += d[3];
                    const w2_u32x2 pd = {w2_pk(d[0] * unit, d[1] * unit), w2_pk(d[2] * unit, d[3] * unit)};
                    const f32x4 xx = xv[b][s][j];
                    const w2_u32x2 px = {w2_pk(kel[0] ? xx[0] : 0.f, kel[1] ? xx[1] : 0.f), w2_pk(kel[2] ? xx[2] : 0.f, kel[3] ? xx[3] : 0.f)};
                    const uint32_t a = buf + (uint32_t)((r0 + 8 * j) * W2_RS + 8 * cg);
                    if (!(MPG_W2_EXP & 2)) {
                        *reinterpret_cast<__attribute__((address_space(3))) w2_u32x2*>((uintptr_t)a) = pd;
                        *reinterpret_cast<__attribute__((address_space(3))) w2_u32x2*>((uintptr_t)(a + W2_IMG)) = px;
                    } else {
                        bsum[0] += __builtin_bit_cast(float, pd[0] ^ px[1]); bsum[1] += __builtin_bit_cast(float, pd[1] ^ px[0]);
                    }
                }
                w2_barrier();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint32_t fa = buf + frag0 + (uint32_t)(16 * ks * W2_RS + 128 * wr);
                    const uint32_t fb = buf + W2_IMG + frag0 + (uint32_t)(16 * ks * W2_RS + 128 * wc);
                    f16x8 a0, a1, b0, b1;
                    if (!(MPG_W2_EXP & 2)) { a0 = w2_frag(fa, 0); a1 = w2_frag(fa, 64); b0 = w2_frag(fb, 0); b1 = w2_frag(fb, 64); }
                    else { a0 = a1 = b0 = b1 = __builtin_bit_cast(f16x8, w2_u32x4{fa, fb, fa ^ fb, fa + fb}); }
                    if (!(MPG_W2_EXP & 1)) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
                    } else { acc[0][0][0] += (float)a0[0] + (float)b0[0] + (float)a1[1] + (float)b1[1]; }
                }
                ++it;
            }
        });
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    request(B0{}, st0);
    for (int c0 = st0; c0 < st1; c0 += 2 * W2_CHUNK) {
        chunk(B0{}, c0);
        if (c0 + W2_CHUNK < st1) chunk(B1{}, c0 + W2_CHUNK);
    }

    // ---- partial tile: part[z][n][k] = out_scale * 2^-ecur * acc (D layout: register 4g+t = row 8g + 4h + t, lane & 31 = column)
    {
        const float sc = J.out_scale * __builtin_bit_cast(float, (uint32_t)(127 - ecur) << 23);
        float* part = J.part + (size_t)z * J.split_stride;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)((size_t)J.N * J.ldp * 4), 0x00020000);
        const int c = lane & 31, h = lane >> 5;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int k = k0 + 64 * wc + 32 * b + c;
                const int nb = n0 + 64 * wr + 32 * a + 4 * h;
                const bool kok = k < J.K;
                int obase = (nb * J.ldp + k) * 4;
                asm volatile("" : "+v"(obase));   // (a select between two ready values stays a select; with the product behind it: a branch per store)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int dn = 8 * (i >> 2) + (i & 3);
                    const int o = obase + dn * (J.ldp * 4);
                    const bool ok = (nb + dn < J.N) & kok & !((MPG_W2_EXP & 8) && i + a + b > 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, acc[a][b][i] * sc), rp, ok ? o : 0x7ffffff0, 0, 0);
                }
            }
        // bias gradient: column sums of dy, from the workgroups of tile column 0
        if (J.hb && k0 == 0) {
            w2_barrier();
#pragma unroll
            for (int e = 0; e < 4; ++e) red[r0 * 128 + 4 * cg + e] = bsum[e];
            w2_barrier();
            if (tid < 128 && n0 + tid < J.N) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) s += red[r * 128 + tid];
                part[(size_t)(n0 + tid) * J.ldp + J.K] = s * J.out_scale;
            }
        }
    }
}

}  // namespace

int mpg_wgrad2_launch(const W2Group* G, hipStream_t st) {
    MPG_ENSURE_LDS(wgrad2_kernel, W2_LDS);
    hipLaunchKernelGGL(wgrad2_kernel, dim3(G->wg0[G->n]), dim3(256), W2_LDS, st, *G);
    return (int)hipGetLastError();
}
