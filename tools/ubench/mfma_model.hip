// Micro-benchmarks that pin the single-wave-per-SIMD machine model the fused edge kernels are
// scheduled against (gfx950): dependent MFMA chains, VALU issued in the MFMA shadow, LDS latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP 64
// NV = independent VALU ops after each MFMA; DEP = 1: one accumulator chain, 2: two alternating chains
template <int NV, int DEP>
__global__ __launch_bounds__(256, 1) void k_mfma(float* out, long long* cyc, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    f32x16 acc0 = {0}, acc1 = {0};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (DEP == 1 || (r & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[q % 8]) : "v"(v[(q + 1) % 8]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// LDS: chain of dependent ds_read_b128 (address from previous result) -> latency; 4 waves per CU active
__global__ __launch_bounds__(256, 1) void k_lds_lat(float* out, long long* cyc, int iters) {
    extern __shared__ uint4 sm[];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = make_uint4((i * 16 + 16 * 64) % (8192 * 16), 0, 0, 0);
    __syncthreads();
    uint32_t addr = threadIdx.x * 16;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            uint4 x;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr));
            addr = x.x;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = addr;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// LDS throughput: independent ds_read_b128 streams, all 4 waves
template <int NB>
__global__ __launch_bounds__(256, 1) void k_lds_bw(float* out, long long* cyc, int iters) {
    extern __shared__ uint4 sm[];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = make_uint4(i, 0, 0, 0);
    __syncthreads();
    uint32_t addr = threadIdx.x * 16;
    uint32_t s = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint4 x[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[r]) : "v"(addr), "n"(r * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < NB; ++r) s += x[r].x;
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, double units) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    launch(out, cyc); launch(out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); launch(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.3f ms  %10lld clk  %8.2f clk/unit  (%.1f ns/unit)\n", name, ms, c, c / units, ms * 1e6 / units);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int iters = 2000;
#define RUN_MFMA(NV, DEP) run("mfma NV=" #NV " DEP=" #DEP, [&](float* o, long long* c) { hipLaunchKernelGGL((k_mfma<NV, DEP>), dim3(256), dim3(256), 0, 0, o, c, iters); }, (double)iters * REP)
    RUN_MFMA(0, 1); RUN_MFMA(0, 2); RUN_MFMA(2, 1); RUN_MFMA(4, 1); RUN_MFMA(6, 1); RUN_MFMA(8, 1); RUN_MFMA(10, 1); RUN_MFMA(12, 1);
    RUN_MFMA(4, 2); RUN_MFMA(8, 2);
    run("lds latency (dependent b128)", [&](float* o, long long* c) { hipLaunchKernelGGL(k_lds_lat, dim3(256), dim3(256), 131072, 0, o, c, iters); }, (double)iters * 16);
    run("lds bw NB=4 (per b128/wave)", [&](float* o, long long* c) { hipLaunchKernelGGL((k_lds_bw<4>), dim3(256), dim3(256), 131072, 0, o, c, iters); }, (double)iters * 4);
    run("lds bw NB=16 (per b128/wave)", [&](float* o, long long* c) { hipLaunchKernelGGL((k_lds_bw<16>), dim3(256), dim3(256), 131072, 0, o, c, iters); }, (double)iters * 16);
    return 0;
}
