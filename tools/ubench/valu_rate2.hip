// Issue cost of the opcodes mpg_edge_dw's builders are made of, one wave on a SIMD, eight independent chains each
// (s_memtime around 64 x 8 instructions).  hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define BODY8(STMT) STMT(a0) STMT(a1) STMT(a2) STMT(a3) STMT(a4) STMT(a5) STMT(a6) STMT(a7)
#define RUN(NAME, STMT)                                                                                  \
    __global__ void k_##NAME(unsigned* out, unsigned long long* t, unsigned c, unsigned d) {              \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
        for (int i = 0; i < 64; ++i) { BODY8(STMT) }                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
        out[threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                          \
        if (threadIdx.x == 0) t[0] = t1 - t0;                                                              \
    }
#define S_MULF(a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define S_FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
#define S_FMAC(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
#define S_BFE(a) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a));
#define S_BFI(a) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
#define S_CND(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c));
#define S_CVTPK(a) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define S_MIX(a) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(c), "v"(d));
#define S_PKMUL(a) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a) : "v"(c));
#define S_ADDF(a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define S_MAX(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(c));
#define S_XOR(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(c));
#define S_LSHR(a) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a));
#define S_MULLO(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c));
#define S_MULS(a) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "s"(c));
#define S_FMAS(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(c), "v"(d));
#define S_FMAK(a) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a) : "s"(c));
#define S_MULK(a) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a));
#define S_FMACS(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "s"(c), "v"(d));
#define S_MOV(a) asm volatile("v_mov_b32 %0, %0" : "+v"(a));
#define S_ADDS(a) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "s"(c));
#define S_PKFMA(a) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
RUN(mul_f32, S_MULF) RUN(fma_f32, S_FMA) RUN(fmac_f32, S_FMAC) RUN(bfe_i32, S_BFE) RUN(bfi_b32, S_BFI) RUN(cndmask, S_CND)
RUN(cvt_pk_f16, S_CVTPK) RUN(fma_mix, S_MIX) RUN(pk_mul_f16, S_PKMUL) RUN(add_f32, S_ADDF) RUN(max_f32, S_MAX) RUN(xor_b32, S_XOR)
RUN(lshr, S_LSHR) RUN(mul_lo_u32, S_MULLO) RUN(mul_f32_sgpr, S_MULS) RUN(fma_f32_sgpr, S_FMAS) RUN(fma_f32_s_k, S_FMAK) RUN(mul_f32_const, S_MULK) RUN(fmac_sgpr, S_FMACS) RUN(mov, S_MOV) RUN(add_f32_sgpr, S_ADDS) RUN(pk_fma_f16, S_PKFMA)
int main() {
    unsigned* o; unsigned long long* t; hipMalloc(&o, 256); hipMalloc(&t, 8);
    unsigned long long h;
#define GO(NAME) for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_##NAME, dim3(1), dim3(64), 0, 0, o, t, 3u, 5u); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); } \
    printf("%-12s %6.2f ticks per instruction (8 independent chains, 512 instructions)\n", #NAME, (double)h / 512.0);
    GO(mul_f32) GO(fma_f32) GO(fmac_f32) GO(bfe_i32) GO(bfi_b32) GO(cndmask) GO(cvt_pk_f16) GO(fma_mix) GO(pk_mul_f16) GO(add_f32) GO(max_f32) GO(xor_b32) GO(lshr) GO(mul_lo_u32) GO(mul_f32_sgpr) GO(fma_f32_sgpr) GO(fma_f32_s_k) GO(mul_f32_const) GO(fmac_sgpr) GO(mov) GO(add_f32_sgpr) GO(pk_fma_f16)
    printf("-- eight waves per workgroup (two per SIMD), ticks of wave 0 per instruction:\n");
#define GO8(NAME) for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k_##NAME, dim3(1), dim3(512), 0, 0, o, t, 3u, 5u); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); } \
    printf("%-12s %6.2f\n", #NAME, (double)h / 512.0);
    GO8(mul_f32) GO8(fma_f32) GO8(bfi_b32) GO8(cvt_pk_f16) GO8(fma_mix) GO8(mul_f32_sgpr) GO8(lshr)
    return 0;
}
