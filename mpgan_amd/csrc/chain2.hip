// Chained per-node MLP, the form the MPLayer shapes run (mpg_chain's description: include/mpgan_amd.h; the general
// kernel and the layouts: chain.hip).  Same arithmetic, different schedule:
//
//   * four waves, one per SIMD, each owning up to TWO 32-feature output tiles of a layer (tiles w and w + 4).  A whole
//     tile of weight fragments (up to 16 k-steps, hi and lo) sits in registers; the buffer a tile frees is refilled at
//     once with the same slot's tile of the NEXT layer, so a fragment is requested a full tile (>= 1,500 clk) before
//     its MFMA and nothing in a k loop ever waits for L2.
//   * the k loops are straight-line code cut into MFMA slots (common idiom of the edge kernels: one MFMA, a few VALU
//     instructions, a scheduling fence): the epilogue of a wave's first tile -- bias, LeakyReLU, dropout, gate, residual,
//     store, hi/lo split into the next layer's fragments -- runs in the slots of its second tile's MFMAs, element by
//     element.  Only the second tile's epilogue is exposed.  (Measured on the general kernel with MPG_CHSTAMP: per
//     layer 5k clk of MFMA loop + 5.5k clk of epilogue on two waves per SIMD in lock step, against 3k clk of MFMAs.)
//   * no branch inside a loop or an epilogue unit: options are uniform selects, stores go through buffer descriptors
//     whose bounds check drops what must not be written, the dropout mode is a template parameter.
//
// The schedule itself lives in chain2_impl.h (c2_body): this file is the stand-alone kernel -- rows staged from memory --
// and the entry point's shape table; edge_fwd2_impl.h runs the same body as the epilogue of the fused edge forward.
#include "chain2_impl.h"
#include "chain_int.h"
#include <stdlib.h>
#include <stdio.h>

#ifdef MPG_CHSTAMP  // diagnostic build (tools/chain_stamps.py): every wave of workgroup 0
__device__ unsigned long long g_c2_stamps[4 * 24];
#endif

namespace {

// GATES / RESID: bit l set = layer l multiplies by a gate operand / adds a residual (known per shape: no dead arithmetic)
// SL: the LAST layer's rows are not whole 16-byte groups (N or a row stride not a multiple of 4): its output and residual
// go element by element
template <bool F16, int KS0, int KS1, int KS2, int DROP, int GATES, int RESID, bool SL>
__global__ __launch_bounds__(256, 1) void chain2_kernel(const MpgChain p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 32;
#ifdef MPG_CHSTAMP
    unsigned long long c2_stamps[24] = {};
    unsigned long long* const c2_st = c2_stamps;
#else
    unsigned long long* const c2_st = nullptr;
#endif
    auto stage = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t seed_lo, const uint32_t seed_hi,
                     const float ascale) {
        c2_stage_rows<F16, KS0, DROP>(p, m0, 32, smem, first_tile, bias_request, bias_store, seed_lo, seed_hi, ascale, c2_st);
    };
    c2_body<F16, KS0, KS1, KS2, DROP, GATES, RESID, SL>(p, m0, 32, smem, smem + C2_FB, reinterpret_cast<float*>(smem + 2 * C2_FB), stage, c2_st);
#ifdef MPG_CHSTAMP
    if (blockIdx.x == 0 && (tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 24; ++i) g_c2_stamps[(tid >> 6) * 24 + i] = c2_st[i];
    }
#endif
}

template <bool F16, int A, int B, int C, int DROP, int GATES, int RESID, bool SL>
int c2_launch(const MpgChain* p, hipStream_t st) {
    MPG_ENSURE_LDS((chain2_kernel<F16, A, B, C, DROP, GATES, RESID, SL>), C2_LDS);
    hipLaunchKernelGGL((chain2_kernel<F16, A, B, C, DROP, GATES, RESID, SL>), dim3((p->M + 31) / 32), dim3(256), C2_LDS, st, *p);
    return (int)hipGetLastError();
}
template <bool F16, int A, int B, int C, int DROP, int GATES, int RESID>
int c2_launch_sl(const MpgChain* p, bool sl, hipStream_t st) {
    return sl ? c2_launch<F16, A, B, C, DROP, GATES, RESID, true>(p, st) : c2_launch<F16, A, B, C, DROP, GATES, RESID, false>(p, st);
}
template <bool F16, int A, int B, int C, int GATES, int RESID>
int c2_launch_drop(const MpgChain* p, int drop, bool sl, hipStream_t st) {
    if (drop == 2) return c2_launch_sl<F16, A, B, C, 2, GATES, RESID>(p, sl, st);
    if (drop == 1) return c2_launch_sl<F16, A, B, C, 1, GATES, RESID>(p, sl, st);
    return c2_launch_sl<F16, A, B, C, 0, GATES, RESID>(p, sl, st);
}

}  // namespace

#ifdef MPG_CHSTAMP
extern "C" int mpg_debug_chain2_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_c2_stamps), sizeof(unsigned long long) * 4 * 24);
}
#endif

#define C2_NA(why) do { if (dbg) fprintf(stderr, "mpg_chain: general kernel (%s; M %d, layers %d, K %d %d %d, N %d %d %d)\n", why, p->M, p->nlayers, \
        p->L[0].K, p->L[1].K, p->L[2].K, p->L[0].N, p->L[1].N, p->L[2].N); return MPG_CHAIN2_NA; } while (0)

int mpg_chain2_try(const MpgChain* p, hipStream_t st) {
    if (getenv("MPG_CHAIN_GENERAL")) return MPG_CHAIN2_NA;
    static const bool dbg = getenv("MPG_CHAIN_DEBUG") != nullptr;
    int ks[3] = {0, 0, 0};
    int drop = 0;   // 0 none, 1 byte mode, 2 bit mode -- one mode for every site of the call
    auto site = [&](uint32_t thr) {
        if (!thr) return true;
        const int mode = thr == 128u ? 2 : 1;
        if (drop && drop != mode) return false;
        drop = mode;
        return true;
    };
    if (!site(p->in_thr)) C2_NA("mixed dropout modes");
    bool sl = false;   // the last layer's rows are not whole 16-byte groups
    for (int l = 0; l < p->nlayers; ++l) {
        const MpgChainLayer& L = p->L[l];
        ks[l] = 2 * ((L.K + 31) / 32);
        if (L.N > 256) C2_NA("N");
        const bool vec = L.N % 4 == 0 && (L.out == nullptr || (L.ldo % 4 == 0 && ((uintptr_t)L.out & 15) == 0)) &&
                         (L.resid == nullptr || (L.ldr % 4 == 0 && ((uintptr_t)L.resid & 15) == 0));
        if (!vec) {
            if (l + 1 != p->nlayers) C2_NA("row groups of an inner layer");
            sl = true;
        }
        if (L.out != nullptr && (size_t)p->M * L.ldo * 4 >= 0x7fffffffull) C2_NA("out");
        if (L.gateH != nullptr && (L.ldh % 4 || ((uintptr_t)L.gateH & 15) || (size_t)p->M * L.ldh * 4 >= 0x7fffffffull)) C2_NA("gate");
        if (L.resid != nullptr && (size_t)p->M * L.ldr * 4 >= 0x7fffffffull) C2_NA("resid");
        if (!site(L.drop_thr) || !site(L.gateH != nullptr ? L.gate_thr : 0)) C2_NA("mixed dropout modes");
    }
    if (p->in_out != nullptr && (size_t)p->M * p->ld_in_out * 4 >= 0x7fffffffull) C2_NA("in_out");
    int gates = 0, resid = 0;
    for (int l = 0; l < p->nlayers; ++l) {
        gates |= (p->L[l].gateH != nullptr) << l;
        resid |= (p->L[l].resid != nullptr) << l;
    }
    if (p->f16) {
        if (p->nlayers == 3 && ks[0] == 14 && ks[1] == 16 && ks[2] == 16 && !gates && !resid)
            return c2_launch_drop<true, 14, 16, 16, 0, 0>(p, drop, sl, st);                                                   // fn forward
        if (p->nlayers == 1 && ks[0] == 2 && drop == 0 && !gates && !resid && !sl) return c2_launch<true, 2, 0, 0, 0, 0, 0, false>(p, st);  // a | c projection
        C2_NA("f16 shape");
    }
    if (p->nlayers == 3 && ks[0] == 2 && ks[1] == 16 && ks[2] == 16 && gates == 3 && !resid)
        return c2_launch_drop<false, 2, 16, 16, 3, 0>(p, drop, sl, st);                                                       // fn input gradients
    if (p->nlayers == 1 && ks[0] == 12 && drop == 0 && !gates && resid == 1) return c2_launch_sl<false, 12, 0, 0, 0, 0, 1>(p, sl, st);   // dx from da | dc
    C2_NA("shape");
}
