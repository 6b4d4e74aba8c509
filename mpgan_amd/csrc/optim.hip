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
// device memory (a captured hipGraph must see it advance on every replay).
#include "common.h"
#include "../../include/mpgan_amd.h"

namespace {
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v, size_t n,
                               float lr, float alpha, float eps, float gscale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float vi = alpha * v[i] + (1.f - alpha) * gi * gi;
        v[i] = vi;
        p[i] -= lr * gi / (sqrtf(vi) + eps);
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ step, size_t n, float lr, float b1, float b2, float eps, float wd,
                            float gscale) {
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
    }
}
__global__ void step_inc_kernel(float* step) { *step += 1.f; }

__global__ void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v, float* __restrict__ u,
                                size_t n, float lr, float rho, float eps, float gscale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float vi = rho * v[i] + (1.f - rho) * gi * gi;
        const float d = sqrtf(u[i] + eps) / sqrtf(vi + eps) * gi;
        v[i] = vi;
        u[i] = rho * u[i] + (1.f - rho) * d * d;
        p[i] -= lr * d;
    }
}

inline int nblocks(uint64_t n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }
}  // namespace

extern "C" int mpg_rmsprop(float* p, const float* g, float* v, uint64_t n, float lr, float alpha, float eps,
                           float gscale, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(rmsprop_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, v, (size_t)n, lr, alpha,
                       eps, gscale);
    return (int)hipGetLastError();
}

extern "C" int mpg_adam(float* p, const float* g, float* m, float* v, float* step, uint64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float gscale, void* stream) {
    if (n == 0) return 0;
    if (step == nullptr) return -1;
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, step, (size_t)n, lr,
                       beta1, beta2, eps, weight_decay, gscale);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    return (int)hipGetLastError();
}

extern "C" int mpg_adadelta(float* p, const float* g, float* v, float* u, uint64_t n, float lr, float rho, float eps,
                            float gscale, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(adadelta_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, v, u, (size_t)n, lr, rho,
                       eps, gscale);
    return (int)hipGetLastError();
}
