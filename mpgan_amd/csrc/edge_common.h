// Shared pieces of the fused edge-network kernels (forward: edge.hip, backward: edge_bwd.hip).
#pragma once
#include "common.h"
#include "../../include/mpgan_amd.h"

#ifndef MPG_CHAIN_PF
#define MPG_CHAIN_PF 1
#endif

namespace {


constexpr int H1 = 96, H2 = 160, H3 = 192;
constexpr int T1 = 3, T2 = 5, T3 = 6;  // 32-row tiles per layer width

// fragment fetch: from the LDS copy when present, else straight from the (L2-resident) image
template <bool LDS, typename V>
MPG_DEV V frag(const V* __restrict__ glb, const V* lds, int idx) {
    if constexpr (LDS) return lds[idx];
    else return glb[idx];
}

// LDS fragment reads as  ds_read_b128 v, base offset:imm  -- `base` is a per-lane byte address made opaque to the
// optimiser (otherwise it materialises one address register per fragment and parks them in AGPRs: an extra
// v_accvgpr_read in front of every read), `off` a compile-time byte offset below 64 KiB.
MPG_DEV uint32_t lds_base(const void* smem_ptr, int byte_off) {
    uint32_t b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem_ptr + (uint32_t)byte_off;
    asm volatile("" : "+v"(b));
    return b;
}
template <typename V>
MPG_DEV V lds_frag(uint32_t base, int off) {
    return *reinterpret_cast<__attribute__((address_space(3))) const V*>(base + (uint32_t)off);
}

MPG_DEV float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Streamed weight fragments go through buffer loads: one VGPR (lane * 16) + a scalar fragment
// offset.  Plain pointer arithmetic makes hipcc keep a 64-bit VGPR address per fragment (hundreds of
// registers hoisted out of the sender loop, spilled to scratch -- and every scratch reload drains
// vmcnt, which serialises the whole prefetch pipeline).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
MPG_DEV __amdgpu_buffer_rsrc_t img_rsrc(const void* p, int nfrag) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, nfrag * 1024, 0x00020000);
}
template <typename V>
MPG_DEV V img_frag(__amdgpu_buffer_rsrc_t r, int lane16, int frag) {
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, frag * 1024, 0));
}

// global -> LDS copy by LDS-DMA: 1 KiB per wave-instruction straight into LDS (lane-linear: the image in LDS is the image in
// memory), no registers, every piece of a wave in flight at once -- behind it the kernel's other prologue loads land while
// the image streams in (tools/ubench/fill_rate.hip: 120 KiB on four waves in ~4,000 clk at the texture path's 28 B/clk/CU;
// the register copy below took two dependent passes of 16 loads + 16 ds_writes each)
MPG_DEV void fill_lds_dma(void* dst, const void* src, int bytes, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int c = wave; c < bytes / 1024; c += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(src) + c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(static_cast<char*>(dst) + c * 1024), 16, 0, 0);
}

// global -> LDS copy with 16 x 16 B loads in flight per thread (a plain copy loop runs one L2 round
// trip per iteration: ~10 us for the 150 KiB of weight images)
template <typename V>
MPG_DEV void copy_to_lds(V* dst, const V* __restrict__ src, int n16, int tid) {
    for (int base = 0; base < n16; base += 256 * 16) {
        V tmp[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = base + u * 256 + tid;
            if (i < n16) tmp[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = base + u * 256 + tid;
            if (i < n16) dst[i] = tmp[u];
        }
    }
}

// One 32-row output tile of a chained layer: KS = 2*QT k-steps of 3 MFMAs.  Fragments of k-step
// k+1 are requested before the MFMAs of k-step k are issued, and after every k-step a slice of
// OTHER work (the epilogue of the previous tile, passed as `side(k)`) is placed, so that a single
// wave keeps the matrix pipe and the VALU busy together; sched_barrier pins that order.
template <int KS, typename V, typename LH, typename LL, typename Side>
MPG_DEV void tile_chain(f32x16& acc, const V (*bhi)[2], const V (*blo)[2], LH load_hi, LL load_lo, Side side) {
    constexpr int PF = MPG_CHAIN_PF;  // k-steps of fragment prefetch (ring of PF + 1 slots)
    V ah[PF + 1], al[PF + 1];
#pragma unroll
    for (int k = 0; k < PF && k < KS; ++k) {
        ah[k] = load_hi(k);
        al[k] = load_lo(k);
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        if (k + PF < KS) {
            ah[(k + PF) % (PF + 1)] = load_hi(k + PF);
            al[(k + PF) % (PF + 1)] = load_lo(k + PF);
        }
        acc = mfma3(ah[k % (PF + 1)], al[k % (PF + 1)], bhi[k >> 1][k & 1], blo[k >> 1][k & 1], acc);
        side(k);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Power-of-two operand scales of the FORWARD products (and their recomputation in the backward).  An fp16 hi/lo pair
// carries 22 significand bits only while lo = x - hi stays a normal fp16 number, i.e. for |x| >= 2^-3; below that lo
// is subnormal and the pair's absolute error is stuck at 2^-25 -- for weights of ~0.05 that is 10x the error of
// fp32, enough to flip the sign of near-zero pre-activations (LeakyReLU kinks) measurably more often than fp32
// does.  Scaling by a power of two is exact, so: layer-1 terms a, c carry SC_A, the W2 image SC_W2, the W3 image
// SC_W3; LeakyReLU is positively homogeneous, so e1 carries SC_A, e2 SC_A*SC_W2 (= SC_E2) and the layer-3 output
// SC_E2*SC_W3 (= SC_E3), which the aggregation folds into m_j * dscale.  (fp16 range: e1 < 16k, e2 < 1k, weights < 1k.)
constexpr float SC_A = 4.f, SC_W2 = 16.f, SC_W3 = 64.f, SC_E2 = SC_A * SC_W2, SC_E3 = SC_E2 * SC_W3;

// if_set / if_clear chosen by bit BIT of `word`: v_bfe_i32 + v_bfi_b32 (left to itself the compiler builds a compare,
// two wait states for VCC and a v_cndmask per element).  Behind the inline assembly the compiler pads dependent instructions
// with s_nop 0 (340 of the 3,760 instructions of a pair in edge_bwd_kernel); the same select from __builtin_amdgcn_sbfe with
// an opaque width and the bit-select pattern (-> v_bfe_i32 + v_bitop3_b32, 39 nops) measured the same to the microsecond
// (60.1 / 140.9 against 60.7 / 140.6 us): an s_nop does not take an issue turn.
template <int BIT>
MPG_DEV float sel_by_bit(uint32_t word, float if_set, float if_clear) {
    int m;
    float r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "n"(BIT));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(if_set), "v"(if_clear));
    return r;
}

// Dither factor of a block (jet, receiver block, sender) of the edge backward: c in [1, 2), a hash of the block index.
// mpg_edge_bwd works on the block's gradients times c, mpg_edge_dw divides the parked dZ2 by it (edge_bwd2_impl.h).
MPG_DEV float dither_of(uint32_t blk) {
    uint32_t x = blk * 0x9E3779B1u + 0x7F4A7C15u;
    x ^= x >> 15; x *= 0x85EBCA77u;
    x ^= x >> 13;
    return __builtin_bit_cast(float, 0x3F800000u | (x >> 9));
}

constexpr int NF3T = T2 * T3 * 2;  // W3^T image: 5 row tiles x 6 k-tiles x 2 = 60 fragments
constexpr int NF2T = T1 * T2 * 2;  // W2^T image: 3 x 5 x 2 = 30
constexpr int NFR2 = T2 * 2;       // B-operand fragments of a 160-feature tensor

// LDS plan of the forward kernel: W3 hi | W3 lo | W2 hi  (W2 lo streams from L2)
constexpr int NF2 = T2 * T1 * 2;  // 30 fragments of 1 KiB
constexpr int NF3 = T3 * T2 * 2;  // 60
constexpr int FWD_W_BYTES = (2 * NF3 + NF2) * 1024;    // 153,600 weight images
constexpr int FWD_BIAS_BYTES = (H2 + H3) * 4;          //   1,408 b2 | b3
constexpr int FWD_C_SLOTS = 22;                        // sender rows of c staged in LDS (22 * 384 B)
constexpr int FWD_LDS_BYTES = FWD_W_BYTES + FWD_BIAS_BYTES + FWD_C_SLOTS * H1 * 4;  // 163,456 <= 163,840
constexpr int RED_BYTES = 4 * T3 * 16 * 64 * 4;        //  98,304
constexpr int NOLDS_BYTES = RED_BYTES + FWD_BIAS_BYTES + FWD_C_SLOTS * H1 * 4;

}  // namespace
