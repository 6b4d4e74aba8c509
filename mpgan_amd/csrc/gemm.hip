// Generic fp32-in / fp32-out GEMM on the bf16 matrix cores (3-term split), with the fused
// epilogues the node network (fn), the layer-1 node terms (a, c) and GAPT's projections need.
//
//   C[M,N] = epi( sum_k A(m,k) * B(k,n) )
//   A(m,k) = AK ? A[m*lda + k] : A[k*lda + m]        (k >= K1 reads the optional 2nd segment A2)
//   B(k,n) = BK ? B[n*ldb + k] : B[k*ldb + n]
//
//   (AK,BK) = (1,1): Y = X W^T          (nn.Linear forward,      mpgan/model.py:78)
//   (AK,BK) = (1,0): dX = dY W          (its input gradient)
//   (AK,BK) = (0,0): dW = dY^T X        (its weight gradient; split-K over gridDim.z)
//
// Tile 64x64x32, 256 threads = 4 waves in a 2x2 grid of 32x32 MFMA tiles.  Operands are
// split to bf16 hi/lo while being staged into LDS in MFMA-fragment order (16 B per lane,
// conflict-free ds_read_b128).
#include "common.h"
#include "reduce_group.h"
#include "gemm.h"

namespace {

// Staging is split in two so the loop can overlap: tile_load() puts the next k-tile's 8 floats per
// thread and operand into registers BEFORE the MFMAs of the current tile, tile_store() converts them to
// 16-bit hi/lo and writes the LDS fragments after the barrier (one workgroup has only a handful of
// k-tiles and few co-resident workgroups, so an un-pipelined loop pays a full HBM/L2 round trip per tile).
// `ones_row` (>= 0): that row of the operand is the constant 1 (virtual ones column: C gets the column
// sums of the other operand, i.e. the bias gradient, for free).
template <bool KC>
MPG_DEV void tile_load(float (&v)[8], const float* __restrict__ P, int ld, const float* __restrict__ P2, int ld2,
                       int K1, int row0, int nrows, int kt, int kend, int tid, bool vec_ok, int ones_row) {
    if constexpr (KC) {
        const int row = tid >> 2, kc = tid & 3;
        const int gr = row0 + row;
        const int k0 = kt + 8 * kc;
        if (vec_ok && gr < nrows && gr != ones_row && k0 + 8 <= kend && P2 == nullptr) {
            const float4 u0 = *reinterpret_cast<const float4*>(P + (size_t)gr * ld + k0);
            const float4 u1 = *reinterpret_cast<const float4*>(P + (size_t)gr * ld + k0 + 4);
            v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w;
            v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float x = 0.f;
                if (gr < nrows && k < kend) {
                    if (gr == ones_row) x = 1.f;
                    else x = (P2 != nullptr && k >= K1) ? P2[(size_t)gr * ld2 + (k - K1)] : P[(size_t)gr * ld + k];
                }
                v[e] = x;
            }
        }
    } else {
        // rows (m or n) contiguous in memory: lane = row (a wave reads 256 contiguous bytes per k), each
        // thread takes 8 consecutive k of its row = exactly one fragment's elements
        const int row = tid & 63, kg = tid >> 6;
        const int gr = row0 + row;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kt + 8 * kg + e;
            float x = 0.f;
            if (gr < nrows && k < kend) x = gr == ones_row ? 1.f : P[(size_t)k * ld + gr];
            v[e] = x;
        }
    }
}

template <bool KC, typename V>
MPG_DEV void tile_store(const float (&v)[8], V (*hi)[2][64], V (*lo)[2][64], int tid) {
    typedef typename ElemOf<V>::type E;
    if constexpr (KC) {
        const int row = tid >> 2, kc = tid & 3;
        V h8, l8;
        split8(v, h8, l8);
        const int blk = row >> 5, r = row & 31, s = kc >> 1, h = kc & 1;
        hi[blk][s][h * 32 + r] = h8;
        lo[blk][s][h * 32 + r] = l8;
    } else {
        const int row = tid & 63, kg = tid >> 6;
        V h8, l8;
        split8(v, h8, l8);
        const int blk = row >> 5, r = row & 31, s = kg >> 1, h = kg & 1;
        hi[blk][s][h * 32 + r] = h8;
        lo[blk][s][h * 32 + r] = l8;
    }
}

template <bool AK, bool BK, bool F16>
MPG_DEV void gemm_body(const MpgGemm& g_, const int bx, const int by, const int bz, const int nz, const bool fold_ones = false) {
    typedef typename FragT<F16>::type V;
    __shared__ V As_hi[2][2][64], As_lo[2][2][64], Bs_hi[2][2][64], Bs_lo[2][2][64];
    // fold_ones (the grouped weight-gradient launch, when the ones column would be alone in a 64-wide tile column of its
    // own -- K a multiple of 64): the products run on the N - 1 real columns and the workgroups of tile column 0 add up
    // the rows of their A tiles, which they have in registers anyway, into column N - 1 (the bias gradient)
    MpgGemm g = g_;
    if (fold_ones) { g.N -= 1; g.ones_col = 0; }
    float rsum = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int m0 = by * 64, n0 = bx * 64;
    // split-K range of this z-slice (multiples of 32)
    const int ktiles = (g.K + 31) / 32;
    const int per = (ktiles + nz - 1) / nz;
    const int kbeg = bz * per * 32;
    const int kend_z = min(g.K, (bz + 1) * per * 32);
    const bool a_vec = AK && (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    const bool b_vec = BK && (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int b_ones = g.ones_col ? g.N - 1 : -1;  // last "row" of the B operand is the ones column
    float va[8], vb[8];
    if (kbeg < kend_z) {
        tile_load<AK>(va, g.A, g.lda, g.A2, g.lda2, g.K1, m0, g.M, kbeg, kend_z, tid, a_vec, -1);
        tile_load<BK>(vb, g.B, g.ldb, nullptr, 0, 0, n0, g.N, kbeg, kend_z, tid, b_vec, b_ones);
    }
    for (int kt = kbeg; kt < kend_z; kt += 32) {
        if (fold_ones && bx == 0) rsum += ((va[0] + va[1]) + (va[2] + va[3])) + ((va[4] + va[5]) + (va[6] + va[7]));
        tile_store<AK>(va, As_hi, As_lo, tid);
        tile_store<BK>(vb, Bs_hi, Bs_lo, tid);
        __syncthreads();
        if (kt + 32 < kend_z) {  // next tile's global loads fly during the MFMAs
            tile_load<AK>(va, g.A, g.lda, g.A2, g.lda2, g.K1, m0, g.M, kt + 32, kend_z, tid, a_vec, -1);
            tile_load<BK>(vb, g.B, g.ldb, nullptr, 0, 0, n0, g.N, kt + 32, kend_z, tid, b_vec, b_ones);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
            acc = mfma3(As_hi[wr][s][lane], As_lo[wr][s][lane], Bs_hi[wc][s][lane], Bs_lo[wc][s][lane], acc);
        __syncthreads();
    }

    if constexpr (!AK) {
        if (fold_ones && bx == 0) {   // thread (row = tid & 63, k group = tid >> 6) holds its row's sum over its 8 k of every tile
            float* red = reinterpret_cast<float*>(&As_hi[0][0][0]);   // (the loop ended with a barrier: the buffers are free)
            red[tid] = rsum;
            __syncthreads();
            const int m = m0 + tid;
            if (tid < 64 && m < g.M)
                (g.C + (size_t)bz * g.split_stride)[(size_t)m * g.ldc + g.N] =
                    ((red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192])) * g.out_scale;
        }
    }
    // ---------------- epilogue (D layout: reg 4g+t -> row 8g+4h+t, lane&31 -> col)
    const int c = lane & 31, h = lane >> 5;
    const int n = n0 + 32 * wc + c;
    if (n >= g.N) return;
    uint32_t seed_lo = 0, seed_hi = 0;
    if (g.seed != nullptr) { const uint64_t sd = *g.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }
    const float bias = (g.bias != nullptr && bz == 0) ? g.bias[n] : 0.f;
    float* Cz = g.C + (size_t)bz * g.split_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wr + 8 * (r >> 2) + 4 * h + (r & 3);
        if (m >= g.M) continue;
        float v = acc[r] * g.out_scale + bias;
        if (g.act == 1) v = lrelu(v, g.alpha);
        if (g.drop_thr) {  // forward dropout on the output element (m, n)
            v = drop_keep_f(seed_lo, seed_hi, g.drop_tag, (uint32_t)m, n, g.drop_thr) ? v * g.drop_scale : 0.f;
        }
        if (g.gateH != nullptr) {  // backward through (dropout o leaky-relu) of the layer that produced H
            const float hv = g.gateH[(size_t)m * g.ldh + n];
            float gt = g.gate_act ? lrelu_grad(hv, g.alpha) : 1.f;
            if (g.gate_thr) {
                gt = drop_keep_f(seed_lo, seed_hi, g.gate_tag, (uint32_t)m, n, g.gate_thr) ? gt * g.gate_scale : 0.f;
            }
            v *= gt;
        }
        if (g.resid != nullptr) v += g.resid[(size_t)m * g.ldr + n];
        float* dst = Cz + (size_t)m * g.ldc + n;
        if (g.accumulate) v += *dst;
        *dst = v;
    }
}

template <bool AK, bool BK, bool F16>
__global__ __launch_bounds__(256) void gemm_kernel(const MpgGemm g) {
    gemm_body<AK, BK, F16>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z);
}

// Several independent weight-gradient GEMMs (dW = dY^T X, split-K) in ONE launch: each of these is a few hundred
// short workgroups that cannot fill the chip or hide their own latencies; side by side they do.
// (only the fields the weight-gradient form uses travel as kernel arguments: 16 whole MpgGemm would not fit the 4 KiB)
// the ones column would be the only column of the last 64-wide tile column
__host__ __device__ inline bool wgrad_fold_ones(int N, int ones_col) { return ones_col && N > 1 && (N - 1) % 64 == 0; }
struct GemmSlim { const float* A; const float* B; float* C; long long split_stride; int lda, ldb, ldc, M, N, K; float out_scale; int ones_col; };
struct GemmGroup { GemmSlim g[MPG_GROUP_MAX]; int splitk[MPG_GROUP_MAX]; int wg0[MPG_GROUP_MAX + 1]; int n; };
template <bool F16>
__global__ __launch_bounds__(256) void gemm_group_kernel(const GemmGroup G) {
    int q = 0;
    while ((int)blockIdx.x >= G.wg0[q + 1]) ++q;
    const GemmSlim& e = G.g[q];
    MpgGemm g = {};
    g.A = e.A; g.B = e.B; g.C = e.C; g.split_stride = e.split_stride;
    g.lda = e.lda; g.ldb = e.ldb; g.ldc = e.ldc; g.M = e.M; g.N = e.N; g.K = e.K;
    g.out_scale = e.out_scale; g.ones_col = e.ones_col; g.alpha = 0.2f; g.f16 = F16;
    const int local = blockIdx.x - G.wg0[q];
    const bool fold = wgrad_fold_ones(g.N, g.ones_col);
    const int tx = (g.N - (fold ? 1 : 0) + 63) / 64, ty = (g.M + 63) / 64;
    gemm_body<false, false, F16>(g, local % tx, (local / tx) % ty, local / (tx * ty), G.splitk[q], fold);
}

__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const ReduceGroup R) { splitk_reduce_group_body(R, (int)blockIdx.x); }

// out[m,n] = in[m,n] * gate(H[m,n]) : the elementwise "backward through dropout (and leaky
// relu)" step applied to an upstream gradient before it enters a GEMM.
__global__ void gate_kernel(const float* __restrict__ in, int ldi, const float* __restrict__ H, int ldh,
                            float* __restrict__ out, int ldo, int M, int N, int gate_act, float alpha,
                            const uint64_t* seed, uint32_t tag, uint32_t thr, float scale) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * N) return;
    const int m = idx / N, n = idx % N;
    uint32_t seed_lo = 0, seed_hi = 0;
    if (seed != nullptr) { const uint64_t sd = *seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }
    float gt = 1.f;
    if (gate_act && H != nullptr) gt = lrelu_grad(H[(size_t)m * ldh + n], alpha);
    if (thr) {
        gt = drop_keep_f(seed_lo, seed_hi, tag, (uint32_t)m, n, thr) ? gt * scale : 0.f;
    }
    out[(size_t)m * ldo + n] = in[(size_t)m * ldi + n] * gt;
}

// out[m, 0:cols] = sum_s A[s][m][:],  out[m, cols:2 cols] = sum_s B[s][m][:]  -- the partial da (one slab per sender chunk) and
// dc (one per receiver block) of a sender-chunked data-gradient launch, added slab by slab in index order; one float4 per thread.
__global__ void slab_sums_kernel(const float* __restrict__ A, int slabsA, size_t strideA, const float* __restrict__ B, int slabsB,
                                 size_t strideB, float* __restrict__ out, int M, int cols) {
    const int q = cols / 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * 2 * q) return;
    const int m = (int)(idx / (2 * q)), r = (int)(idx % (2 * q));
    const bool second = r >= q;
    const float* src = (second ? B : A) + (size_t)m * cols + 4 * (second ? r - q : r);
    const int n = second ? slabsB : slabsA;
    const size_t st = second ? strideB : strideA;
    float4 acc = *reinterpret_cast<const float4*>(src);
    for (int sl = 1; sl < n; ++sl) {
        const float4 t = *reinterpret_cast<const float4*>(src + sl * st);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)m * 2 * cols + 4 * r) = acc;
}

// out[n, col0 + k] = sum_z part[z][n][k]  (k < K);  bias[n] = sum_z part[z][n][K] when the GEMM carried a
// ones column (ld of the partials = K + has_bias).  One launch replaces torch's sum + strided copy (+ bias sum).
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, int N, int K, int has_bias,
                                     float* __restrict__ out, int ldo, float* __restrict__ bias) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ldp = K + has_bias;
    if (idx >= N * ldp) return;
    const int n = idx / ldp, k = idx % ldp;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += part[(size_t)z * N * ldp + idx];
    if (k < K) out[(size_t)n * ldo + k] = s;
    else if (bias != nullptr) bias[n] = s;
}

// keep-mask materialiser for tests: mask[row, f] in {0,1} exactly as the kernels decide it
__global__ void drop_mask_kernel(float* __restrict__ out, size_t rows, int F, const uint64_t* seed, uint32_t tag,
                                 uint32_t thr) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * (size_t)F) return;
    const size_t row = idx / F;
    const int f = idx % F;
    const uint64_t sd = *seed;
    out[idx] = (thr == 0 || drop_keep_f((uint32_t)sd, (uint32_t)(sd >> 32), tag, (uint32_t)row, f, thr)) ? 1.f : 0.f;
}

}  // namespace

extern "C" int mpg_gemm(const MpgGemm* g, int ak, int bk, int splitk, void* stream) {
    if (g->M <= 0 || g->N <= 0) return 0;
    if (g->K < 0 || splitk < 1) return -1;
    dim3 grid((g->N + 63) / 64, (g->M + 63) / 64, splitk), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MPG_LAUNCH(AKV, BKV)                                                                              \
    do {                                                                                                  \
        if (g->f16) hipLaunchKernelGGL((gemm_kernel<AKV, BKV, true>), grid, block, 0, st, *g);           \
        else hipLaunchKernelGGL((gemm_kernel<AKV, BKV, false>), grid, block, 0, st, *g);                 \
    } while (0)
    if (ak && bk) MPG_LAUNCH(true, true);
    else if (ak && !bk) MPG_LAUNCH(true, false);
    else if (!ak && !bk) MPG_LAUNCH(false, false);
    else MPG_LAUNCH(false, true);
#undef MPG_LAUNCH
    return (int)hipGetLastError();
}

extern "C" int mpg_gemm_wgrad_group(const MpgGemm* g, const int* splitk, int n, void* stream) {
    if (n < 1 || n > MPG_GROUP_MAX) return -1;
    GemmGroup G;
    G.n = n;
    G.wg0[0] = 0;
    for (int q = 0; q < n; ++q) {
        if (g[q].M <= 0 || g[q].N <= 0 || g[q].K < 0 || splitk[q] < 1 || g[q].f16 != g[0].f16) return -1;
        if (g[q].A2 != nullptr || g[q].bias != nullptr || g[q].act || g[q].gateH != nullptr || g[q].resid != nullptr ||
            g[q].accumulate || g[q].drop_thr) return -2;   // the grouped launch is the plain dW = scale * A^T B form
        GemmSlim& e = G.g[q];
        e.A = g[q].A; e.B = g[q].B; e.C = g[q].C; e.split_stride = g[q].split_stride;
        e.lda = g[q].lda; e.ldb = g[q].ldb; e.ldc = g[q].ldc; e.M = g[q].M; e.N = g[q].N; e.K = g[q].K;
        e.out_scale = g[q].out_scale; e.ones_col = g[q].ones_col;
        G.splitk[q] = splitk[q];
        const int ncol = g[q].N - (wgrad_fold_ones(g[q].N, g[q].ones_col) ? 1 : 0);
        G.wg0[q + 1] = G.wg0[q] + ((ncol + 63) / 64) * ((g[q].M + 63) / 64) * splitk[q];
    }
    if (g[0].f16) hipLaunchKernelGGL((gemm_group_kernel<true>), dim3(G.wg0[n]), dim3(256), 0, (hipStream_t)stream, G);
    else hipLaunchKernelGGL((gemm_group_kernel<false>), dim3(G.wg0[n]), dim3(256), 0, (hipStream_t)stream, G);
    return (int)hipGetLastError();
}

extern "C" int mpg_splitk_reduce_group(const MpgReduceJob* jobs, int n, void* stream) {
    if (n < 1) return -1;
    ReduceGroup R;
    const int nb = make_reduce_group(jobs, n, R);
    if (nb < 0) return -1;
    if (nb == 0) return 0;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, R);
    return (int)hipGetLastError();
}

extern "C" int mpg_splitk_reduce(const float* part, int S, int N, int K, int has_bias, float* out, int ldo,
                                 float* bias, void* stream) {
    const int tot = N * (K + has_bias);
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, S, N, K,
                       has_bias, out, ldo, bias);
    return (int)hipGetLastError();
}

extern "C" int mpg_gate(const float* in, int ldi, const float* H, int ldh, float* out, int ldo, int M, int N,
                        int gate_act, float alpha, const uint64_t* seed, uint32_t tag, uint32_t thr, float scale,
                        void* stream) {
    const size_t tot = (size_t)M * N;
    if (tot == 0) return 0;
    hipLaunchKernelGGL(gate_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, ldi, H, ldh, out,
                       ldo, M, N, gate_act, alpha, seed, tag, thr, scale);
    return (int)hipGetLastError();
}

extern "C" int mpg_slab_sums(const float* A, int slabsA, uint64_t strideA, const float* B, int slabsB, uint64_t strideB,
                             float* out, int M, int cols, void* stream) {
    if (M <= 0) return 0;
    if (cols <= 0 || cols % 4 != 0 || slabsA < 1 || slabsB < 1 || strideA % 4 != 0 || strideB % 4 != 0) return -2;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)out) & 15) return -5;
    const size_t tot = (size_t)M * (cols / 2);
    hipLaunchKernelGGL(slab_sums_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, slabsA, (size_t)strideA, B,
                       slabsB, (size_t)strideB, out, M, cols);
    return (int)hipGetLastError();
}

extern "C" int mpg_dropout_mask(float* out, uint64_t rows, int F, const uint64_t* seed, uint32_t tag, uint32_t thr,
                                void* stream) {
    const size_t tot = (size_t)rows * F;
    if (tot == 0) return 0;
    hipLaunchKernelGGL(drop_mask_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, (size_t)rows,
                       F, seed, tag, thr);
    return (int)hipGetLastError();
}
