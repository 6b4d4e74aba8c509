// How fast does a workgroup get a weight image into LDS when every CU wants one at once?  (mab.hip's prologue: 80 KiB forward,
// 144 KiB backward.)  Variants: LDS-DMA (global_load_lds_dwordx4) vs registers + ds_write_b128; 4 / 8 / 16 waves issuing; the
// same image for every workgroup vs one image each; 256 / 128 / 64 workgroups.  Prints clocks (s_memtime) from the first
// issue to the barrier behind the last piece, workgroup 0 and the median workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o fill_rate fill_rate.hip && ./fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <bool DMA>
__global__ void fill_kernel(const char* src, long stride_wg, int bytes, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, nwav = blockDim.x >> 6, lane = threadIdx.x & 63;
    const char* s = src + stride_wg * blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int n = bytes / 1024;
    if constexpr (DMA) {
        for (int c = wave; c < n; c += nwav)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
    } else {
        for (int c0 = wave; c0 < n; c0 += nwav * 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int c = c0 + j * nwav; if (c < n) v[j] = *reinterpret_cast<const float4*>(s + c * 1024 + lane * 16); }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int c = c0 + j * nwav; if (c < n) *reinterpret_cast<float4*>(smem + c * 1024 + lane * 16) = v[j]; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t0; }
    if (sink != nullptr) sink[threadIdx.x] = reinterpret_cast<float*>(smem)[threadIdx.x * 7 % (bytes / 4)];
}

int main() {
    const int maxb = 144 * 1024;
    char* src; hipMalloc(&src, (size_t)maxb * 256); hipMemset(src, 1, (size_t)maxb * 256);
    unsigned long long* out; hipMalloc(&out, 256 * 16);
    float* sink; hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void*)fill_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
    hipFuncSetAttribute((const void*)fill_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
    std::vector<unsigned long long> h(512);
    for (int bytes : {80 * 1024, 144 * 1024})
        for (int dma = 1; dma >= 0; --dma)
            for (int shared = 1; shared >= 0; --shared)
                for (int wgs : {256, 128, 64, 8})
                    for (int threads : {256, 512, 1024}) {
                        unsigned long long best_issue = ~0ull, best_all = ~0ull, med_all = 0;
                        for (int rep = 0; rep < 5; ++rep) {
                            if (dma) hipLaunchKernelGGL(fill_kernel<true>, dim3(wgs), dim3(threads), bytes, 0, src, shared ? 0L : (long)maxb, bytes, out, sink);
                            else hipLaunchKernelGGL(fill_kernel<false>, dim3(wgs), dim3(threads), bytes, 0, src, shared ? 0L : (long)maxb, bytes, out, sink);
                            hipDeviceSynchronize();
                            hipMemcpy(h.data(), out, wgs * 16, hipMemcpyDeviceToHost);
                            std::vector<unsigned long long> all;
                            for (int i = 0; i < wgs; ++i) all.push_back(h[2 * i + 1]);
                            std::sort(all.begin(), all.end());
                            if (rep > 0 && all[wgs / 2] < best_all) { best_all = all[wgs / 2]; best_issue = h[0]; med_all = all[wgs - 1]; }
                        }
                        printf("%3d KiB %s %s wgs=%3d waves=%2d : median WG %6llu clk (%.1f B/clk/CU), slowest %6llu, wg0 issue %6llu\n", bytes / 1024,
                               dma ? "lds-dma " : "register", shared ? "one image " : "image each", wgs, threads / 64, best_all, (double)bytes / best_all, med_all, best_issue);
                    }
    return 0;
}
