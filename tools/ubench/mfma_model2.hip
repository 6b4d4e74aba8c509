// Follow-up to mfma_model.hip: does VALU work hide behind an MFMA whose accumulator lives in AGPRs, when the
// VALU work itself reads OTHER AGPRs (v_accvgpr_read) or LDS?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP 64
// MODE 0: acc in AGPR, filler = v_fma on VGPRs; 1: acc in AGPR, filler = v_accvgpr_read of another AGPR tile
// 2: acc in VGPR, filler = v_fma; 3: acc AGPR, filler = v_accvgpr_read + v_mul + v_max (epilogue-like triple)
// 4: acc AGPR, filler = ds_read_b128 (NV of them) ; 5: acc in AGPR, filler = v_mov from VGPR tile
template <int NV, int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    extern __shared__ uint4 sm[];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = make_uint4(i, 0, 0, 0);
    __syncthreads();
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    f32x16 acc = {0}, other = {0};
    for (int j = 0; j < 16; ++j) other[j] = threadIdx.x + j;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    uint32_t addr = threadIdx.x * 16;
    uint4 ld[4] = {};
    if (MODE != 2) asm volatile("" : "+a"(acc), "+a"(other));
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (MODE == 0 || MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[q % 8]) : "v"(v[(q + 1) % 8]));
                else if (MODE == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[q % 8]) : "a"(other[q % 16]));
                else if (MODE == 3) {
                    if (q % 3 == 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[q % 8]) : "a"(other[q % 16]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[q % 8]) : "v"(v[(q + 1) % 8]));
                } else if (MODE == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[q % 4]) : "v"(addr), "n"((q % 4) * 1024));
            }
            if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (MODE != 2) asm volatile("" : "+a"(acc), "+a"(other));
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j] + other[j];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += ld[j].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename F>
static void run(const char* name, F launch, double units) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    launch(out, cyc); launch(out, cyc);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); launch(out, cyc); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms  %8.2f clk/mfma  (%.1f ns/mfma)\n", name, ms, c / units, ms * 1e6 / units);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    const int iters = 1000;
#define RUN(NV, MODE, label) run(label " NV=" #NV, [&](float* o, long long* c) { hipLaunchKernelGGL((k<NV, MODE>), dim3(256), dim3(256), 65536, 0, o, c, iters); }, (double)iters * REP)
    RUN(0, 0, "acc AGPR, no filler       "); RUN(0, 2, "acc VGPR, no filler       ");
    RUN(6, 0, "acc AGPR, v_fma           "); RUN(6, 2, "acc VGPR, v_fma           ");
    RUN(3, 1, "acc AGPR, accvgpr_read    "); RUN(6, 1, "acc AGPR, accvgpr_read    ");
    RUN(6, 3, "acc AGPR, read+fma+fma    "); RUN(9, 3, "acc AGPR, read+fma+fma    ");
    RUN(1, 4, "acc AGPR, ds_read_b128    "); RUN(2, 4, "acc AGPR, ds_read_b128    "); RUN(4, 4, "acc AGPR, ds_read_b128    ");
    return 0;
}
