// Issue cost of a few VALU opcodes on one wave per SIMD (s_memtime around 64 x 8 independent instructions).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define RUN(NAME, ASM)                                                                                   \
    __global__ void k_##NAME(unsigned* out, unsigned long long* t) {                                       \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const unsigned c = 0x9E3779B1u;                                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
        for (int i = 0; i < 64; ++i) {                                                                     \
            asm volatile(ASM " %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile(ASM " %0, %0, %1" : "+v"(a1) : "v"(c)); \
            asm volatile(ASM " %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile(ASM " %0, %0, %1" : "+v"(a3) : "v"(c)); \
            asm volatile(ASM " %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile(ASM " %0, %0, %1" : "+v"(a5) : "v"(c)); \
            asm volatile(ASM " %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile(ASM " %0, %0, %1" : "+v"(a7) : "v"(c)); \
        }                                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
        out[threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                          \
        if (threadIdx.x == 0) t[0] = t1 - t0;                                                              \
    }
RUN(mul_lo, "v_mul_lo_u32")
RUN(mul_u24, "v_mul_u32_u24")
RUN(xorb, "v_xor_b32")
RUN(add, "v_add_u32")
RUN(mulf, "v_mul_f32")
int main() {
    unsigned* o; unsigned long long* t; hipMalloc(&o, 256); hipMalloc(&t, 8);
    unsigned long long h;
#define GO(NAME) for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_##NAME, dim3(1), dim3(64), 0, 0, o, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); } \
    printf("%-10s %6.2f ticks per instruction (s_memtime ticks; 512 instructions)\n", #NAME, (double)h / 512.0);
    GO(mul_lo) GO(mul_u24) GO(xorb) GO(add) GO(mulf)
    return 0;
}
