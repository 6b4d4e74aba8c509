// Fused RMSprop over one flat parameter buffer (one launch per network per step).
//
// Replaces torch.optim.RMSprop(...).step() as configured by the reference
// (setup_training.py:1511-1513: lr only, i.e. alpha = 0.99, eps = 1e-8, no momentum, not
// centered, no weight decay):   v = alpha v + (1 - alpha) g^2 ;  p -= lr * g / (sqrt(v) + eps).
// `gscale` multiplies the gradient first (1/world_size after a summing all-reduce).
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
}  // namespace

extern "C" int mpg_rmsprop(float* p, const float* g, float* v, uint64_t n, float lr, float alpha, float eps,
                           float gscale, void* stream) {
    if (n == 0) return 0;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, v, (size_t)n, lr, alpha,
                       eps, gscale);
    return (int)hipGetLastError();
}
