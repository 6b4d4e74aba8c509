// Does the sustained MFMA rate depend on operand data (power throttling)?  Dependent chain of
// v_mfma_f32_32x32x16_f16 on all 256 CUs x 4 SIMDs with (a) constant small operands, (b) random operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ uint32_t hsh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int RANDOM>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    f16x8 a[4], b[4];
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            uint32_t r = hsh(threadIdx.x * 977 + blockIdx.x * 131 + q * 17 + j);
            a[q][j] = RANDOM ? (_Float16)(((int)(r & 0xffff) - 32768) / 32768.f) : (_Float16)1.0f;
            b[q][j] = RANDOM ? (_Float16)(((int)(r >> 16) - 32768) / 32768.f) : (_Float16)0.5f;
        }
    f32x16 acc0 = {0}, acc1 = {0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (r & 8) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r & 3], b[(r >> 2) & 3], acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r & 3], b[(r >> 2) & 3], acc0, 0, 0, 0);
        }
        if (RANDOM && (it & 15) == 15) { for (int j = 0; j < 16; ++j) { acc0[j] *= 1e-3f; acc1[j] *= 1e-3f; } }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int RANDOM> void run(int grid, int iters) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k<RANDOM>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<RANDOM>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 64;
    printf("%s operands, grid %4d, %6d MFMA/wave: %8.3f ms  %6.2f ns/MFMA  %6.2f clk/MFMA  -> %.0f TFLOP/s dense f16\n", RANDOM ? "random  " : "constant", grid,
           iters * 64, ms, ms * 1e6 / n, c / n, grid * 4 * n * 32768 / (ms * 1e-3) / 1e12);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>(256, 40); run<1>(256, 40);      // ~2500 MFMAs per wave: the length of one edge-kernel launch
    run<0>(256, 2000); run<1>(256, 2000);  // 2 ms of sustained work
    run<0>(64, 2000); run<1>(64, 2000);    // a quarter of the chip
    return 0;
}
