// Fused optimiser steps over one flat parameter buffer (one launch per network per step).
//
// Replace torch.optim.{RMSprop, Adam, Adadelta}(...).step() as the reference configures them
// (setup_training.py:1511-1523):
//   rmsprop  (default)  lr only: alpha = 0.99, eps = 1e-8, no momentum, not centered, no weight decay
//                       v = alpha v + (1 - alpha) g^2 ;  p -= lr * g / (sqrt(v) + eps)
//   adam                lr, betas = (beta1, beta2), weight_decay = 5e-4 (L2, added to the gradient), eps = 1e-8
//                       m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
//   adadelta            lr, rho = 0.9, eps = 1e-6
//                       v = rho v + (1-rho) g^2 ; d = sqrt(u + eps)/sqrt(v + eps) * g ; u = rho u + (1-rho) d^2 ; p -= lr d
// `gscale` multiplies the gradient first (1/world_size after a summing all-reduce).  Adam's step count lives in
// device memory (a captured hipGraph must see it advance on every replay).  `zero_grad` != 0: the gradient buffer is
// cleared behind its last use (the next backward accumulates into zeros: optimizer.zero_grad() of train.py:419 / :494
// without a launch of its own).
#include "common.h"
#include "../../include/mpgan_amd.h"

namespace {
__global__ void rmsprop_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ v, size_t n,
                               float lr, float alpha, float eps, float gscale, int zero_grad, uint64_t* __restrict__ counter, uint64_t counter_add) {
    if (counter != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *counter += counter_add;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float vi = alpha * v[i] + (1.f - alpha) * gi * gi;
        v[i] = vi;
        p[i] -= lr * gi / (sqrtf(vi) + eps);
        if (zero_grad) g[i] = 0.f;
    }
}

__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ step, size_t n, float lr, float b1, float b2, float eps, float wd,
                            float gscale, int zero_grad, uint64_t* __restrict__ counter, uint64_t counter_add) {
    if (counter != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *counter += counter_add;
    const float t = *step + 1.f;
    const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float pi = p[i];
        const float gi = g[i] * gscale + wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) / bc2s + eps);
        if (zero_grad) g[i] = 0.f;
    }
}
__global__ void step_inc_kernel(float* step) { *step += 1.f; }

__global__ void adadelta_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ v, float* __restrict__ u,
                                size_t n, float lr, float rho, float eps, float gscale, int zero_grad, uint64_t* __restrict__ counter, uint64_t counter_add) {
    if (counter != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *counter += counter_add;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float vi = rho * v[i] + (1.f - rho) * gi * gi;
        const float d = sqrtf(u[i] + eps) / sqrtf(vi + eps) * gi;
        v[i] = vi;
        u[i] = rho * u[i] + (1.f - rho) * d * d;
        p[i] -= lr * d;
        if (zero_grad) g[i] = 0.f;
    }
}

// out[i] = mean + std * z_i, z ~ N(0, 1): counter-based (two hashes of (seed, tag, pair index) -> Box-Muller, one pair of
// values per thread), so a captured hipGraph draws fresh values on every replay from the device-resident seed alone
__global__ void normal_kernel(float* __restrict__ out, size_t n, const uint64_t* __restrict__ seed, uint32_t tag, float mean, float std) {
    const uint64_t sd = *seed;
    const uint32_t lo = (uint32_t)sd, hi = (uint32_t)(sd >> 32);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x, npair = (n + 1) / 2;
    for (; i < npair; i += stride) {
        // (the pair index goes through the mixer twice, the second time with the words swapped: two independent 32-bit streams)
        uint32_t a = drop_word(lo, hi, tag, (uint32_t)i, (uint32_t)(i >> 32));
        uint32_t b = drop_word(hi ^ 0x9E3779B9u, lo, tag + 0x7F4A7C15u, (uint32_t)i ^ a, (uint32_t)(i >> 32) + 1u);
        const float u1 = ((float)(a >> 8) + 0.5f) * (1.f / 16777216.f);      // (0, 1)
        const float u2 = ((float)(b >> 8) + 0.5f) * (1.f / 16777216.f);
        const float r = sqrtf(-2.f * logf(u1)) * std;
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        out[2 * i] = mean + r * cs;
        if (2 * i + 1 < n) out[2 * i + 1] = mean + r * sn;
    }
}

// The generator's input noise AND its jets' masks (mask_c, mpgan/model.py:689-699: the n = int(label N) particles with the
// smallest first noise feature) in one launch: the ranking reads nothing but the noise just drawn.  One workgroup per jet;
// the values are mpg_normal's (same pair indices over the flat [B, N, L] tensor), the mask mpg_rank_mask's.
__global__ __launch_bounds__(256) void normal_rank_mask_kernel(float* __restrict__ out, int N, int L, const uint64_t* __restrict__ seed,
                                                               uint32_t tag, float mean, float std, const float* __restrict__ labels,
                                                               int ld_lab, float* __restrict__ mask, float* __restrict__ ignore) {
    extern __shared__ float first[];   // [N]: the first feature of every particle
    const uint64_t sd = *seed;
    const uint32_t lo = (uint32_t)sd, hi = (uint32_t)(sd >> 32);
    const int b = blockIdx.x;
    const size_t per_jet = (size_t)N * L, pair0 = (size_t)b * per_jet / 2, npair = per_jet / 2;   // (N L even: checked by the launcher)
    for (size_t q = threadIdx.x; q < npair; q += blockDim.x) {
        const size_t i = pair0 + q;
        uint32_t a = drop_word(lo, hi, tag, (uint32_t)i, (uint32_t)(i >> 32));
        uint32_t c = drop_word(hi ^ 0x9E3779B9u, lo, tag + 0x7F4A7C15u, (uint32_t)i ^ a, (uint32_t)(i >> 32) + 1u);
        const float u1 = ((float)(a >> 8) + 0.5f) * (1.f / 16777216.f);
        const float u2 = ((float)(c >> 8) + 0.5f) * (1.f / 16777216.f);
        const float r = sqrtf(-2.f * logf(u1)) * std;
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        const float v0 = mean + r * cs, v1 = mean + r * sn;
        out[2 * i] = v0;
        out[2 * i + 1] = v1;
        const size_t e0 = 2 * q;                       // element index within the jet
        if (e0 % L == 0) first[e0 / L] = v0;
        if ((e0 + 1) % L == 0) first[(e0 + 1) / L] = v1;   // (L == 1 only)
    }
    __syncthreads();
    const int n_minus_1 = (int)(labels[(size_t)b * ld_lab] * (float)N) - 1;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float xi = first[i];
        int rank = 0;
        for (int j = 0; j < N; ++j) {
            const float xj = first[j];
            rank += (xj < xi) || (xj == xi && j < i);
        }
        mask[(size_t)b * N + i] = rank <= n_minus_1 ? 1.f : 0.f;
        if (ignore != nullptr) ignore[(size_t)b * N + i] = rank <= n_minus_1 ? 0.f : 1.f;
    }
}

inline int nblocks(uint64_t n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }
}  // namespace

extern "C" int mpg_rmsprop(float* p, float* g, float* v, uint64_t n, float lr, float alpha, float eps,
                           float gscale, int zero_grad, uint64_t* counter, uint64_t counter_add, void* stream) {
    if (n == 0) return counter != nullptr ? -1 : 0;
    hipLaunchKernelGGL(rmsprop_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, v, (size_t)n, lr, alpha,
                       eps, gscale, zero_grad, counter, counter_add);
    return (int)hipGetLastError();
}

extern "C" int mpg_adam(float* p, float* g, float* m, float* v, float* step, uint64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float gscale, int zero_grad, uint64_t* counter, uint64_t counter_add,
                        void* stream) {
    if (n == 0) return counter != nullptr ? -1 : 0;
    if (step == nullptr) return -1;
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, step, (size_t)n, lr,
                       beta1, beta2, eps, weight_decay, gscale, zero_grad, counter, counter_add);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    return (int)hipGetLastError();
}

extern "C" int mpg_adadelta(float* p, float* g, float* v, float* u, uint64_t n, float lr, float rho, float eps,
                            float gscale, int zero_grad, uint64_t* counter, uint64_t counter_add, void* stream) {
    if (n == 0) return counter != nullptr ? -1 : 0;
    hipLaunchKernelGGL(adadelta_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, v, u, (size_t)n, lr, rho,
                       eps, gscale, zero_grad, counter, counter_add);
    return (int)hipGetLastError();
}

extern "C" int mpg_normal(float* out, uint64_t n, const uint64_t* seed, uint32_t tag, float mean, float std, void* stream) {
    if (n == 0) return 0;
    if (seed == nullptr || !(std >= 0.f)) return -1;
    hipLaunchKernelGGL(normal_kernel, dim3(nblocks((n + 1) / 2)), dim3(256), 0, (hipStream_t)stream, out, (size_t)n, seed, tag, mean, std);
    return (int)hipGetLastError();
}

extern "C" int mpg_normal_rank_mask(float* out, int B, int N, int L, const uint64_t* seed, uint32_t tag, float mean, float std,
                                    const float* labels, int ld_lab, float* mask, float* ignore, void* stream) {
    if (B <= 0 || N <= 0 || L <= 0 || N > 8192 || ((size_t)N * L) % 2) return -1;
    if (seed == nullptr || labels == nullptr || mask == nullptr || !(std >= 0.f)) return -1;
    hipLaunchKernelGGL(normal_rank_mask_kernel, dim3(B), dim3(256), N * sizeof(float), (hipStream_t)stream, out, N, L, seed, tag, mean,
                       std, labels, ld_lab, mask, ignore);
    return (int)hipGetLastError();
}
