// nn.BatchNorm1d over the rows of x [M, F] -- the optional normalisation of LinearNet (mpgan/model.py:58-60, :80-81:
// Linear -> LeakyReLU -> BatchNorm1d -> Dropout).  Off in every published configuration and only reachable on the un-fused
// route (an edge network with batch norm normalises over all B*N*N edge rows of the batch), so these are plain
// bandwidth kernels: lane = feature, waves stride over the rows of a chunk, per-chunk partial sums in a fixed order.
//   stats : mean_f = sum_m x / M ;  var_f = sum_m (x - mean_f)^2 / M      (two passes: no cancellation)
//   apply : y = (x - mean) rsqrt(var + eps) w + b
//   bwd   : db = sum g ; dw = sum g xhat ; dx = w rstd (g - db / M - xhat dw / M)
#include "common.h"
#include "../../include/mpgan_amd.h"

namespace {

// part[c][f] (+ part2) = sum over the rows of chunk c of  v(m, f)
template <int MODE>   // 0: x ; 1: (x - mean)^2 ; 2: g and g * xhat
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg,
                                                         const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                         int M, int F, int nchunk, float* __restrict__ part) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane, c = blockIdx.y;
    const int per = (M + nchunk - 1) / nchunk, m0 = c * per, m1 = min(M, m0 + per);
    float s0 = 0.f, s1 = 0.f;
    if (f < F) {
        const float mu = MODE ? mean[f] : 0.f;
        const float rstd = MODE == 2 ? rsqrtf(var[f] + eps) : 0.f;
        for (int m = m0 + w; m < m1; m += 4) {
            const float xv = x[(size_t)m * ldx + f];
            if constexpr (MODE == 0) s0 += xv;
            if constexpr (MODE == 1) s0 += (xv - mu) * (xv - mu);
            if constexpr (MODE == 2) { const float gv = g[(size_t)m * ldg + f]; s0 += gv; s1 += gv * (xv - mu) * rstd; }
        }
    }
    red[0][w][lane] = s0; red[1][w][lane] = s1;
    __syncthreads();
    if (w == 0 && f < F) {
        part[(size_t)c * F + f] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        if constexpr (MODE == 2)
            part[(size_t)(nchunk + c) * F + f] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    }
}
// out[f] = scale * sum_c part[c][f]  (+ out[f] when accumulate)
__global__ void bn_finalize_kernel(const float* __restrict__ part, int nchunk, int F, float scale, int accumulate, float* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += part[(size_t)c * F + f];
    out[f] = s * scale + (accumulate ? out[f] : 0.f);
}
__global__ void bn_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean, const float* __restrict__ var,
                                const float* __restrict__ w, const float* __restrict__ b, float eps, float* __restrict__ y, int ldy,
                                int M, int F) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * F) return;
    const int m = (int)(i / F), f = (int)(i % F);
    y[(size_t)m * ldy + f] = (x[(size_t)m * ldx + f] - mean[f]) * rsqrtf(var[f] + eps) * (w ? w[f] : 1.f) + (b ? b[f] : 0.f);
}
__global__ void bn_dx_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                             const float* __restrict__ var, const float* __restrict__ w, float eps, const float* __restrict__ sums,
                             float* __restrict__ dx, int lddx, int M, int F) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * F) return;
    const int m = (int)(i / F), f = (int)(i % F);
    const float rstd = rsqrtf(var[f] + eps), xhat = (x[(size_t)m * ldx + f] - mean[f]) * rstd, inv = 1.f / (float)M;
    dx[(size_t)m * lddx + f] = (w ? w[f] : 1.f) * rstd * (g[(size_t)m * ldg + f] - sums[f] * inv - xhat * sums[F + f] * inv);
}

}  // namespace

extern "C" int mpg_batchnorm_stats(const float* x, int ldx, int M, int F, float* part, int nchunk, float* mean, float* var, void* stream) {
    if (M < 1 || F < 1 || nchunk < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((F + 63) / 64, nchunk), fin((F + 255) / 256);
    hipLaunchKernelGGL(bn_partial_kernel<0>, grid, dim3(256), 0, st, x, ldx, nullptr, 0, nullptr, nullptr, 0.f, M, F, nchunk, part);
    hipLaunchKernelGGL(bn_finalize_kernel, fin, dim3(256), 0, st, part, nchunk, F, 1.f / (float)M, 0, mean);
    hipLaunchKernelGGL(bn_partial_kernel<1>, grid, dim3(256), 0, st, x, ldx, nullptr, 0, mean, nullptr, 0.f, M, F, nchunk, part);
    hipLaunchKernelGGL(bn_finalize_kernel, fin, dim3(256), 0, st, part, nchunk, F, 1.f / (float)M, 0, var);
    return (int)hipGetLastError();
}

extern "C" int mpg_batchnorm_apply(const float* x, int ldx, const float* mean, const float* var, const float* w, const float* b, float eps,
                                   float* y, int ldy, int M, int F, void* stream) {
    if (M < 1 || F < 1) return -1;
    const size_t tot = (size_t)M * F;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, mean, var, w, b, eps, y,
                       ldy, M, F);
    return (int)hipGetLastError();
}

extern "C" int mpg_batchnorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* mean, const float* var, const float* w, float eps,
                                 float* part, int nchunk, float* sums, float* dx, int lddx, float* dw, float* db, int accumulate, int M, int F,
                                 void* stream) {
    if (M < 1 || F < 1 || nchunk < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((F + 63) / 64, nchunk), fin((F + 255) / 256);
    hipLaunchKernelGGL(bn_partial_kernel<2>, grid, dim3(256), 0, st, x, ldx, g, ldg, mean, var, eps, M, F, nchunk, part);
    hipLaunchKernelGGL(bn_finalize_kernel, fin, dim3(256), 0, st, part, nchunk, F, 1.f, 0, sums);                               // sum g
    hipLaunchKernelGGL(bn_finalize_kernel, fin, dim3(256), 0, st, part + (size_t)nchunk * F, nchunk, F, 1.f, 0, sums + F);     // sum g xhat
    if (db != nullptr) hipLaunchKernelGGL(bn_finalize_kernel, fin, dim3(256), 0, st, part, nchunk, F, 1.f, accumulate, db);
    if (dw != nullptr) hipLaunchKernelGGL(bn_finalize_kernel, fin, dim3(256), 0, st, part + (size_t)nchunk * F, nchunk, F, 1.f, accumulate, dw);
    if (dx != nullptr) {
        const size_t tot = (size_t)M * F;
        hipLaunchKernelGGL(bn_dx_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, g, ldg, x, ldx, mean, var, w, eps, sums, dx,
                           lddx, M, F);
    }
    return (int)hipGetLastError();
}
