// The small per-jet pieces around the message-passing layers, one launch each instead of a dozen elementwise ATen
// kernels (the "glue" of an iteration):
//   mpg_rank_mask        MPGenerator._get_mask, mask_c branch (mpgan/model.py:689-699): the n = int(label * N)
//                        lowest-noise particles of a jet are real
//   mpg_gen_tail_fwd/bwd MPNet._final_activation (tanh, :533-538) + MPGenerator._final_mask (:741-757): the
//                        generator's output rows (tanh(y) | mask - 0.5), written wherever the caller wants them
//                        (e.g. straight into the second half of the discriminator's real+generated batch)
//   mpg_disc_head_fwd/bwd MPDiscriminator._post_mp (masked sum / mean pooling, :812-829) + fnd_layer (one Linear
//                        + Dropout) + the final sigmoid -- and, when a loss is named, the per-jet loss terms of
//                        calc_D_loss / calc_G_loss (train.py:331-395, :465-476) with their gradient, so that the
//                        backward starts from the last MPLayer's output without any autograd node in between
#include "common.h"
#include "../../include/mpgan_amd.h"

namespace {

// rank of particle i among its jet's first features (ties by index) by counting; one workgroup per jet
__global__ __launch_bounds__(256) void rank_mask_kernel(const float* __restrict__ x, int ld_jet, int ld_part,
                                                        const float* __restrict__ labels, int ld_lab, int N,
                                                        float* __restrict__ mask, float* __restrict__ ignore) {
    extern __shared__ float xs[];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += blockDim.x) xs[i] = x[(size_t)b * ld_jet + (size_t)i * ld_part];
    __syncthreads();
    const int n_minus_1 = (int)(labels[(size_t)b * ld_lab] * (float)N) - 1;  // (labels * N).int() - 1
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float xi = xs[i];
        int rank = 0;
        for (int j = 0; j < N; ++j) {
            const float xj = xs[j];
            rank += (xj < xi) || (xj == xi && j < i);
        }
        mask[(size_t)b * N + i] = rank <= n_minus_1 ? 1.f : 0.f;
        if (ignore != nullptr) ignore[(size_t)b * N + i] = rank <= n_minus_1 ? 0.f : 1.f;   // the attention blocks' key mask: 1 - mask
    }
}

// Jets in order of decreasing multiplicity (ties by index): counting sort in one workgroup.  keys[] and the histogram live in LDS.
// The mask is read coalesced (thread = element), the ranks among equal keys wave by wave from ballots: a jet's position is
// (jets with a smaller key) + (jets with its key in earlier waves) + (lanes below it in its wave with its key).
__global__ __launch_bounds__(1024) void jet_order_kernel(const float* __restrict__ mask, int B, int N, int* __restrict__ order) {
    extern __shared__ int jo[];
    int* keys = jo;                  // [B]: first the number of unmasked particles, then N - that (0 = fullest)
    int* hist = jo + B;              // [N + 2]: hist[k + 1] = jets with key k, then its prefix sums
    int* wcnt = jo + B + N + 2;      // [waves][N + 1]: jets with key k in wave w, then in the waves before w
    const int nw = (B + 63) / 64, lane = threadIdx.x & 63;
    for (int t = threadIdx.x; t < N + 2; t += blockDim.x) hist[t] = 0;
    for (int t = threadIdx.x; t < nw * (N + 1); t += blockDim.x) wcnt[t] = 0;
    __syncthreads();
    // a thread counts its jet's particles itself: the N loads of a thread are independent (all in flight together) and the
    // threads of a wave walk one contiguous 64 N-float window -- one round trip instead of B N / 1024 rounds of LDS atomics,
    // thirty to an address
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* mb = mask + (size_t)b * N;
        int c = 0;
#pragma unroll 8
        for (int i = 0; i < N; ++i) c += mb[i] != 0.f;
        const int k = N - c;
        keys[b] = k;
        atomicAdd(&hist[k + 1], 1);
        atomicAdd(&wcnt[(b >> 6) * (N + 1) + k], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 1; k < N + 2; ++k) hist[k] += hist[k - 1];   // hist[k] = jets with a smaller key
    for (int k = threadIdx.x; k <= N; k += blockDim.x) {           // exclusive scan over the waves, per key
        int run = 0;
        for (int w = 0; w < nw; ++w) { const int c = wcnt[w * (N + 1) + k]; wcnt[w * (N + 1) + k] = run; run += c; }
    }
    __syncthreads();
    for (int b0 = (threadIdx.x >> 6) * 64; b0 < B; b0 += blockDim.x) {   // whole waves: ballots need every lane
        const int b = b0 + lane;
        const int k = b < B ? keys[b] : -1;
        // lanes below this one with the same key: compare against every lane's key through the wave
        int below = 0;
        for (int l = 0; l < 64; ++l) {
            const int kl = __shfl(k, l, 64);
            below += (l < lane && kl == k);
        }
        if (b < B) order[hist[k] + wcnt[(b >> 6) * (N + 1) + k] + below] = b;
    }
}

__global__ __launch_bounds__(256) void gen_tail_fwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ mask,
                                                           float* __restrict__ out, int ldo, int V, int F, int act) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    for (int f = 0; f < F; ++f) {
        const float t = y[(size_t)v * ldy + f];
        out[(size_t)v * ldo + f] = act == 1 ? tanhf(t) : (act == 2 ? 1.f / (1.f + expf(-t)) : t);
    }
    if (mask != nullptr) out[(size_t)v * ldo + F] = mask[v] - 0.5f;
}

__global__ __launch_bounds__(256) void gen_tail_bwd_kernel(const float* __restrict__ dout, int ldd, const float* __restrict__ out, int ldo,
                                                           float* __restrict__ dy, int ldy, int V, int F, int act) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    for (int f = 0; f < F; ++f) {
        const float o = out[(size_t)v * ldo + f], g = dout[(size_t)v * ldd + f];
        dy[(size_t)v * ldy + f] = act == 1 ? g * (1.f - o * o) : (act == 2 ? g * o * (1.f - o) : g);
    }
}

// one wave per jet; lane = (particle parity, feature) for F <= 32, features looped beyond
MPG_DEV float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

MPG_DEV void disc_head_bwd_jet(const MpgDiscHead& p, const int b, const int lane, const float out, const float aux0);

template <bool FUSED>
__global__ __launch_bounds__(256) void disc_head_fwd_kernel(const MpgDiscHead p) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= p.B) return;
    const float* yb = p.y + (size_t)b * p.N * p.ldy;
    float z = 0.f, msum = 0.f;
    for (int f0 = 0; f0 < p.F; f0 += 32) {
        const int f = f0 + (lane & 31);
        // (four particles of the lane's parity in flight at a time: one dependent load after the other, the 75 steps of a
        // 150-particle jet took 37 us)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        auto term = [&](int i) {
            return i < p.N ? (p.mask ? p.mask[(size_t)b * p.N + i] : 1.f) * yb[(size_t)i * p.ldy + f] : 0.f;
        };
        if (f < p.F)
            for (int i = lane >> 5; i < p.N; i += 8) { a0 += term(i); a1 += term(i + 2); a2 += term(i + 4); a3 += term(i + 6); }
        float acc = (a0 + a1) + (a2 + a3);
        acc += __shfl_xor(acc, 32, 64);              // both particle parities
        if (p.pooled != nullptr && lane < 32 && f < p.F) p.pooled[(size_t)b * p.F + f] = acc;   // (un-normalised sum)
        z += (lane < 32 && f < p.F) ? acc * p.w[f] : 0.f;
    }
    if (p.mean) {
        for (int i = lane; i < p.N; i += 64) msum += p.mask ? p.mask[(size_t)b * p.N + i] : 1.f;
        msum = wave_sum(msum);
    }
    z = wave_sum(z);
    const float pool_scale = p.mean ? 1.f / (p.mask ? msum + 1e-12f : (float)p.N) : 1.f;
    z = z * pool_scale + (p.bias ? p.bias[0] : 0.f);
    float keep = 1.f;
    if (p.thr) {
        const uint64_t sd = *p.seed;
        keep = drop_keep_f((uint32_t)sd, (uint32_t)(sd >> 32), p.tag, (uint32_t)b, 0, p.thr) ? p.dscale : 0.f;
    }
    z *= keep;
    const float out = p.sigmoid ? 1.f / (1.f + expf(-z)) : z;
    if (lane == 0) {
        p.out[b] = out;
        if (p.aux != nullptr) { p.aux[2 * b] = keep * pool_scale; p.aux[2 * b + 1] = 0.f; }
    }
    // with a loss named (mpg_disc_head_loss) the jet's backward follows at once: its loss gradient needs nothing but its own output
    if (FUSED) disc_head_bwd_jet(p, b, lane, out, keep * pool_scale);
}

// dL/dout of jet b for the named loss (and its loss term); t = 1 for a jet scored against the "real" target
MPG_DEV float loss_grad(int loss, float out, float t, float& term) {
    switch (loss) {
    case 0: term = (out - t) * (out - t); return 2.f * (out - t);                       // ls: mse
    case 1: {                                                                           // og: nn.BCELoss
        const float l1 = fmaxf(logf(out), -100.f), l0 = fmaxf(logf(1.f - out), -100.f);
        term = -(t * l1 + (1.f - t) * l0);
        return (out - t) / fmaxf((1.f - out) * out, 1e-12f);                            // (torch's backward formula)
    }
    case 2: term = t > 0.5f ? -out : out; return t > 0.5f ? -1.f : 1.f;                 // w
    default: {                                                                          // hinge (D); G uses the w form
        const float s = t > 0.5f ? -1.f : 1.f, m = 1.f + s * out;
        term = fmaxf(m, 0.f);
        return m > 0.f ? s : 0.f;
    }
    }
}

// the backward of jet b (one wave), given its output: dL/dout from the named loss (or the upstream gradient), through the
// final activation, the dropout and the pooling down to dy; the per-jet loss term and d/d(pre-dropout z) are left for the reduction
MPG_DEV void disc_head_bwd_jet(const MpgDiscHead& p, const int b, const int lane, const float out, const float aux0) {
    float g;  // dL/dout
    if (p.loss >= 0) {
        // D step: jets [0, n_real) are scored against 1, the rest against 0 (hinge: margins); G step (n_real = B with
        // loss_g): every jet against 1, hinge in its generator form (= w)
        const float t = b < p.n_real ? 1.f : 0.f;
        float term;
        g = loss_grad((p.gen_step && p.loss == 3) ? 2 : p.loss, out, t, term) * p.inv_count;
        if (lane == 0) p.terms[b] = term * p.inv_count;
    } else {
        g = p.gout[b];
    }
    const float gz = g * (p.sigmoid ? out * (1.f - out) : 1.f);        // through the sigmoid
    const float gp = gz * aux0;                                        // through dropout and the pooling normalisation
    if (lane == 0) p.aux[2 * b + 1] = gz * (aux0 != 0.f ? (p.thr ? p.dscale : 1.f) : 0.f);  // d/d(pre-dropout z): for db, dw
    if (p.dy == nullptr) return;
    float* db_ = p.dy + (size_t)b * p.N * p.ld_dy;
    for (int f0 = 0; f0 < p.F; f0 += 32) {
        const int f = f0 + (lane & 31);
        if (f >= p.F) continue;
        const float wf = p.w[f] * gp;
        for (int i = lane >> 5; i < p.N; i += 2) db_[(size_t)i * p.ld_dy + f] = (p.mask ? p.mask[(size_t)b * p.N + i] : 1.f) * wf;
    }
}

__global__ __launch_bounds__(256) void disc_head_bwd_kernel(const MpgDiscHead p) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= p.B) return;
    disc_head_bwd_jet(p, b, lane, p.out[b], p.aux[2 * b]);
}

// workgroup 0: the loss value (sum of the per-jet terms, fixed order); workgroups 1..: the head's own parameter gradients,
// four outputs each, one wave per output striding over the jets with four loads in flight per lane (fixed summation order;
// with 32 lanes per output and one dependent load after the other the 1,024 jets of a GAPT D step took 14 us)
__global__ __launch_bounds__(256) void disc_head_reduce_kernel(const MpgDiscHead p) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        if (p.loss >= 0 && p.loss_out != nullptr) {
            float s = 0.f;
            for (int b = tid; b < p.B; b += 256) s += p.terms[b];
            red[tid] = s;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
            if (tid == 0) *p.loss_out = red[0];
        }
        return;
    }
    if (p.dw == nullptr) return;
    // dw_f = sum_b gzpre_b * pool_scale_b * pooled[b, f] ;  db = sum_b gzpre_b
    const int f = (blockIdx.x - 1) * 4 + (tid >> 6), lane = tid & 63;   // (f == F: the bias)
    if (f > p.F) return;
    const float ds = p.thr ? p.dscale : 1.f;
    auto term = [&](int b) {
        if (b >= p.B) return 0.f;
        const float gzp = p.aux[2 * b + 1];
        if (f >= p.F) return gzp;
        const float a0 = p.aux[2 * b];
        return gzp * (a0 != 0.f ? a0 / ds : 0.f) * p.pooled[(size_t)b * p.F + f];   // pooling normalisation alone
    };
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int b = lane; b < p.B; b += 256) {
        s0 += term(b); s1 += term(b + 64); s2 += term(b + 128); s3 += term(b + 192);
    }
    float s = (s0 + s1) + (s2 + s3);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane == 0) {
        if (f < p.F) p.dw[f] = s + (p.accumulate ? p.dw[f] : 0.f);
        else if (p.db != nullptr) p.db[0] = s + (p.accumulate ? p.db[0] : 0.f);
    }
}

// ---- k-nearest-neighbour sets (MPLayer._getA_knn, mpgan/model.py:319-381) as bit masks; one workgroup per jet
__global__ __launch_bounds__(256) void knn_sets_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mask, int N, int F,
                                                       int k, int first, unsigned int* __restrict__ nbr) {
    extern __shared__ float dist[];   // [N][N]: d(i, j)
    const int b = blockIdx.x, NW = (N + 31) >> 5;
    const float* xb = x + (size_t)b * N * ldx;
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int i = e / N, j = e % N;
        // ((1 - mul) mask + mul), mul = 1e4 (mpgan/model.py:333-335), as the reference evaluates it: continuous in the mask
        // (1 for a real sender, 1e4 for a zero-masked one, in between for the soft masks of mask_exp / learnt masks)
        const float sj = mask != nullptr ? (1.f - 1e4f) * mask[(size_t)b * N + j] + 1e4f : 1.f;
        float acc = 0.f;
        for (int f = 0; f < F; ++f) {
            const float d = sj * xb[(size_t)j * ldx + f] - xb[(size_t)i * ldx + f] + 1e-12f;
            acc += d * d;
        }
        dist[e] = sqrtf(acc);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * NW; e += blockDim.x) {
        const int i = e / NW, wq = e % NW;
        unsigned int bits = 0u;
        for (int jj = 0; jj < 32; ++jj) {
            const int j = wq * 32 + jj;
            if (j >= N) break;
            const float dij = dist[i * N + j];
            int rank = 0;
            for (int l = 0; l < N; ++l) {
                const float dil = dist[i * N + l];
                rank += (dil < dij) || (dil == dij && l < j);
            }
            if (rank >= first && rank < first + k) bits |= 1u << jj;
        }
        nbr[((size_t)b * N + i) * NW + wq] = bits;
    }
}

// ---- LayerNorm over the last dimension (GAPT's MAB with layer_norm: gapt/model.py:118-120, :131-136)
// one wave per row; lane = feature (looped for E > 64); mean / rstd saved for the backward
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y, int ldy,
                                                            float* __restrict__ stats, int M, int E, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int f = lane; f < E; f += 64) s += xr[f];
    const float mean = wave_sum(s) / (float)E;
    float v = 0.f;
    for (int f = lane; f < E; f += 64) { const float d = xr[f] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)E + eps);
    for (int f = lane; f < E; f += 64) y[(size_t)row * ldy + f] = (xr[f] - mean) * rstd * w[f] + b[f];
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// dx per row; the waves of a workgroup walk rows (stride = all waves of the grid) and keep their share of
// dw = sum_rows g * xhat and db = sum_rows g in registers: partial sums [wave][2][E], reduced in fixed order afterwards
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w, const float* __restrict__ stats,
                                                            float* __restrict__ dx, int lddx, float* __restrict__ part,
                                                            int M, int E) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), nwv = gridDim.x * 4;
    float aw[16], ab[16];   // E <= 1024
#pragma unroll
    for (int k = 0; k < 16; ++k) { aw[k] = 0.f; ab[k] = 0.f; }
    for (int row = wv; row < M; row += nwv) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        const float* gr = g + (size_t)row * ldg;
        const float* xr = x + (size_t)row * ldx;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int f = lane + 64 * k;
            if (f < E) {
                const float xh = (xr[f] - mean) * rstd, gw = gr[f] * w[f];
                s1 += gw; s2 += gw * xh;
                aw[k] += gr[f] * xh; ab[k] += gr[f];
            }
        }
        s1 = wave_sum(s1) / (float)E; s2 = wave_sum(s2) / (float)E;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int f = lane + 64 * k;
            if (f < E) {
                const float xh = (xr[f] - mean) * rstd;
                dx[(size_t)row * lddx + f] = rstd * (gr[f] * w[f] - s1 - xh * s2);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int f = lane + 64 * k;
        if (f < E) { part[((size_t)wv * 2 + 0) * E + f] = aw[k]; part[((size_t)wv * 2 + 1) * E + f] = ab[k]; }
    }
}

__global__ __launch_bounds__(256) void layernorm_reduce_kernel(const float* __restrict__ part, int nwv, int E, float* __restrict__ dw,
                                                               float* __restrict__ db, int accumulate) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= 2 * E) return;
    const int which = f / E, ff = f % E;
    float s = 0.f;
    for (int v = 0; v < nwv; ++v) s += part[((size_t)v * 2 + which) * E + ff];
    float* dst = which == 0 ? dw : db;
    dst[ff] = s + (accumulate ? dst[ff] : 0.f);
}

}  // namespace

extern "C" int mpg_knn_sets(const float* x, int ldx, const float* mask, int B, int N, int F, int k, int self_loops,
                            unsigned int* nbr, void* stream) {
    if (B <= 0 || N <= 0 || F <= 0 || k <= 0 || N > 192) return -1;
    const int lds = N * N * (int)sizeof(float);
    MPG_ENSURE_LDS(knn_sets_kernel, lds);
    hipLaunchKernelGGL(knn_sets_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, x, ldx, mask, N, F, k, self_loops ? 0 : 1, nbr);
    return (int)hipGetLastError();
}

extern "C" int mpg_layernorm_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, float* stats,
                                 int M, int E, float eps, void* stream) {
    if (M <= 0 || E <= 0) return -1;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, y, ldy, stats, M, E, eps);
    return (int)hipGetLastError();
}

extern "C" int mpg_layernorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* w, const float* stats, float* dx,
                                 int lddx, float* part, int nwaves, float* dw, float* db, int accumulate, int M, int E,
                                 void* stream) {
    if (M <= 0 || E <= 0 || E > 1024 || nwaves <= 0 || nwaves % 4 != 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nwaves / 4), dim3(256), 0, st, g, ldg, x, ldx, w, stats, dx, lddx, part, M, E);
    if (dw != nullptr)
        hipLaunchKernelGGL(layernorm_reduce_kernel, dim3((2 * E + 255) / 256), dim3(256), 0, st, part, nwaves, E, dw, db, accumulate);
    return (int)hipGetLastError();
}

extern "C" int mpg_rank_mask(const float* x, int ld_jet, int ld_part, const float* labels, int ld_lab, int B, int N,
                             float* mask, float* ignore, void* stream) {
    if (B <= 0 || N <= 0 || N > 8192) return -1;
    hipLaunchKernelGGL(rank_mask_kernel, dim3(B), dim3(N <= 64 ? 64 : 256), N * sizeof(float), (hipStream_t)stream, x, ld_jet, ld_part,
                       labels, ld_lab, N, mask, ignore);
    return (int)hipGetLastError();
}

extern "C" int mpg_jet_order(const float* mask, int B, int N, int* order, void* stream) {
    if (B <= 0 || N <= 0 || mask == nullptr || order == nullptr) return -1;
    const size_t lds = ((size_t)B + N + 2 + (size_t)((B + 63) / 64) * (N + 1)) * sizeof(int);
    if (lds > 64 * 1024) return -2;
    hipLaunchKernelGGL(jet_order_kernel, dim3(1), dim3(B <= 256 ? 256 : 1024), lds, (hipStream_t)stream, mask, B, N, order);
    return (int)hipGetLastError();
}

extern "C" int mpg_gen_tail_fwd(const float* y, int ldy, const float* mask, float* out, int ldo, int V, int F, int act,
                                void* stream) {
    if (V <= 0 || F <= 0) return -1;
    hipLaunchKernelGGL(gen_tail_fwd_kernel, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, ldy, mask, out, ldo, V, F, act);
    return (int)hipGetLastError();
}

extern "C" int mpg_gen_tail_bwd(const float* dout, int ldd, const float* out, int ldo, float* dy, int ldy, int V, int F,
                                int act, void* stream) {
    if (V <= 0 || F <= 0) return -1;
    hipLaunchKernelGGL(gen_tail_bwd_kernel, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, dout, ldd, out, ldo, dy, ldy, V, F, act);
    return (int)hipGetLastError();
}

extern "C" int mpg_disc_head_fwd(const MpgDiscHead* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->F <= 0 || p->out == nullptr || p->aux == nullptr) return -1;
    hipLaunchKernelGGL(disc_head_fwd_kernel<false>, dim3((p->B + 3) / 4), dim3(256), 0, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

extern "C" int mpg_disc_head_loss(const MpgDiscHead* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->F <= 0 || p->out == nullptr || p->aux == nullptr) return -1;
    if (p->loss < 0 || p->terms == nullptr) return -2;
    if (p->dw != nullptr && p->pooled == nullptr) return -2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(disc_head_fwd_kernel<true>, dim3((p->B + 3) / 4), dim3(256), 0, st, *p);
    if (p->loss_out != nullptr || p->dw != nullptr)
        hipLaunchKernelGGL(disc_head_reduce_kernel, dim3(1 + (p->dw != nullptr ? (p->F + 1 + 3) / 4 : 0)), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}

extern "C" int mpg_disc_head_bwd(const MpgDiscHead* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->F <= 0 || p->out == nullptr || p->aux == nullptr) return -1;
    if (p->loss >= 0 ? p->terms == nullptr : p->gout == nullptr) return -2;
    if (p->dw != nullptr && p->pooled == nullptr) return -2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(disc_head_bwd_kernel, dim3((p->B + 3) / 4), dim3(256), 0, st, *p);
    if ((p->loss >= 0 && p->loss_out != nullptr) || p->dw != nullptr)
        hipLaunchKernelGGL(disc_head_reduce_kernel, dim3(1 + (p->dw != nullptr ? (p->F + 1 + 3) / 4 : 0)), dim3(256), 0, st, *p);
    return (int)hipGetLastError();
}
