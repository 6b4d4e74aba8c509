// Common device helpers for the MPGAN/GAPT hot-path kernels (gfx950 / CDNA4 only).
//
// Arithmetic: every matrix product runs on the 16-bit matrix cores (v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate) on
// operands split as x = hi + lo, hi = T(x), lo = T(x - hi).  How many of the partial products are issued:
//   3 terms  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  -- every FORWARD product (T = _Float16 with power-of-two operand scales,
//            product error ~2^-21: they decide on which side of the LeakyReLU kink a pre-activation falls; operands must stay
//            below 65504 after scaling) and the plain gradient GEMMs of the node network and of GAPT (T = __bf16: 8+8 bits,
//            ~2^-17, fp32 exponent range -- gradients have any magnitude);
//   2 terms  weight hi + lo times the gradient rounded to ONE fp16 value in a per-sender power-of-two unit -- the fused edge
//            backward's dE2 = W3^T dZ3, dE1 = W2^T dZ2 (edge_bwd2_impl.h);
//   1 term   fp16 x fp16 -- the fused edge weight gradients, sums over ~1e5 edges of independent roundings (edge_dw.hip).
// All sit inside the 1e-3 parity bar of BASELINE.json (plain bf16 would be ~2e-3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// 16-byte MFMA operand fragment of either 16-bit type
template <bool F16> struct FragT { typedef bf16x8 type; typedef __bf16 elem; };
template <> struct FragT<true> { typedef f16x8 type; typedef _Float16 elem; };
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MPG_DEV __device__ __forceinline__

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1.  The fused kernels index register
// arrays with these constants; `#pragma unroll` over run-time ints gives up on their deeply nested bodies
// (and then indexes registers dynamically through scratch).
template <int B, int E, typename F>
MPG_DEV void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
#define MPG_CI(name, c) constexpr int name = decltype(c)::value

// ---------------------------------------------------------------------------------------
// MFMA 32x32x16 bf16 fragment maps (cdna_hip_programming.md section 3):
//   A operand: lane l (r = l&31, h = l>>5) holds A[row r][k = 8h + j], j = 0..7
//   B operand: lane l holds B[k = 8h + j][col r]
//   C/D      : lane l, reg g*4+t (g,t = 0..3) holds D[row = 8g + 4h + t][col r]
// "Chain" use of a D tile X as the next product's B operand (Y = A.X, contraction over X's
// rows): registers 8s..8s+7 form the fragment of k-step s (s = 0,1); element j of lane half
// h then IS row  rho(s,h,j) = 16s + 8(j>>2) + 4h + (j&3)  of X, so the A operand of that
// k-step must hold column rho(s,h,j) of the weight in its element j.  pack_weights() builds
// weight images in exactly that order, for every layer, so all products share one map.
MPG_DEV int chain_rho(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

MPG_DEV f32x16 mfma3(const bf16x8 ahi, const bf16x8 alo, const bf16x8 bhi, const bf16x8 blo, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc, 0, 0, 0);
    return acc;
}
MPG_DEV f32x16 mfma3(const f16x8 ahi, const f16x8 alo, const f16x8 bhi, const f16x8 blo, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
    return acc;
}

// hi/lo split of 8 floats into two 16-bit x8 fragments
template <typename V> struct ElemOf;
template <> struct ElemOf<bf16x8> { typedef __bf16 type; };
template <> struct ElemOf<f16x8> { typedef _Float16 type; };

template <typename V>
MPG_DEV void split8(const float* v, V& hi, V& lo) {
    typedef typename ElemOf<V>::type E;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const E hh = (E)v[j];
        hi[j] = hh;
        lo[j] = (E)(v[j] - (float)hh);
    }
}
// bf16 form: v_cvt_pk_bf16_f32 per PAIR (left to itself the compiler converts one element per instruction);
// the residual is exact: f32(hi) is the packed half moved to the upper 16 bits.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
MPG_DEV uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    // The conversion as the compiler's own instruction (a pair converts to ONE v_cvt_pk_bf16_f32), not inline assembly: behind
    // assembly the hazard recognizer sees neither the write that an MFMA reads two instructions later nor the read of an MFMA's
    // result.  csrc/mab.hip's large-set backward lost the lo halves of some heads' fragments that way -- 1e-3 of dQ / dK on those
    // heads, gone with `s_nop 3` behind the instruction -- until the instruction was visible to the compiler.  Same values; the
    // headline and GAPT benches are unchanged (128.9k / 129.0k, 986k / 981k jets/s, old and new library alternated on one box).
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t ab = {a, b};
    r = __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, bf16x2));
    return r;
}
template <>
MPG_DEV void split8<bf16x8>(const float* v, bf16x8& hi, bf16x8& lo) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t h4, l4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t hp = cvt_pk_bf16(v[2 * j], v[2 * j + 1]);
        const float r0 = v[2 * j] - __builtin_bit_cast(float, hp << 16);
        const float r1 = v[2 * j + 1] - __builtin_bit_cast(float, hp & 0xffff0000u);
        h4[j] = hp;
        l4[j] = cvt_pk_bf16(r0, r1);
    }
    hi = __builtin_bit_cast(bf16x8, h4);
    lo = __builtin_bit_cast(bf16x8, l4);
}
// fp16 form: per pair one v_cvt_pk_f16_f32 (RTNE) for hi, two v_fma_mix_f32 (x - f32(hi) straight from the
// packed halves, exact) and one v_cvt_pk_f16_f32 for lo -- 2 VALU ops per element instead of 3.5.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <>
MPG_DEV void split8<f16x8>(const float* v, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f16x2 hp = {(_Float16)v[j], (_Float16)v[j + 1]};
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(v[j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(v[j + 1]));
        hi[j] = hp[0]; hi[j + 1] = hp[1];
        lo[j] = (_Float16)r0; lo[j + 1] = (_Float16)r1;
    }
}
// The same split for one PAIR of values, cut in two halves (2 VALU ops each in the fp16 form) so that the
// fused kernels can put them into different issue slots between MFMAs.
template <typename V> struct PairSplit {
    typedef typename ElemOf<V>::type E;
    E h0, h1;
    float r0;
    MPG_DEV void first(float v0, float v1) { h0 = (E)v0; h1 = (E)v1; r0 = v0 - (float)h0; }
    MPG_DEV void second(float v1, V& hi, V& lo, int jj) {
        const float r1 = v1 - (float)h1;
        hi[jj] = h0; hi[jj + 1] = h1;
        lo[jj] = (E)r0; lo[jj + 1] = (E)r1;
    }
};
template <> struct PairSplit<f16x8> {
    f16x2 hp;
    float r0;
    MPG_DEV void first(float v0, float v1) {
        hp = f16x2{(_Float16)v0, (_Float16)v1};
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(v0));
    }
    MPG_DEV void second(float v1, f16x8& hi, f16x8& lo, int jj) {
        float r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(v1));
        hi[jj] = hp[0]; hi[jj + 1] = hp[1];
        lo[jj] = (_Float16)r0; lo[jj + 1] = (_Float16)r1;
    }
};
template <typename E>
MPG_DEV void split1(float x, E& hh, E& ll) { hh = (E)x; ll = (E)(x - (float)hh); }

// LeakyReLU as max(v, alpha v): two VALU ops instead of compare + multiply + select.  Valid for
// 0 <= alpha <= 1 (the launchers reject anything else).
// (inline asm: fmaxf() would add a canonicalising v_max v,v,v per MFMA result)
MPG_DEV float lrelu(float v, float alpha) {
    float r;
    const float va = v * alpha;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(va));
    return r;
}
// derivative as torch's leaky_relu_backward takes it: slope at v <= 0
MPG_DEV float lrelu_grad(float v, float alpha) { return v > 0.f ? 1.f : alpha; }

// ---------------------------------------------------------------------------------------
// Counter-based dropout.  One 32-bit hash word per (row, group of 4 consecutive features);
// byte t of the word decides feature 4*grp + t:  keep  <=>  byte >= thr,  thr = round(256 p).
// (p = 0.5 -> thr = 128, keep probability exactly 1/2.)  The scale applied to kept values
// is 256 / (256 - thr).  Forward and backward regenerate the same word from
// (seed, tag, row, grp); nothing is stored.  `tag` names the dropout site.
struct DropCfg {
    uint32_t seed_lo, seed_hi;
    uint32_t thr;    // 0 => dropout off
    float scale;     // 256/(256-thr)
};

MPG_DEV uint32_t drop_word(uint32_t seed_lo, uint32_t seed_hi, uint32_t tag, uint32_t row, uint32_t grp) {
    uint32_t x = (row + seed_lo) * 0x9E3779B1u;
    x ^= (grp + tag * 0x10001u) * 0x85EBCA77u + seed_hi;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
MPG_DEV bool drop_keep(uint32_t word, int t, uint32_t thr) { return ((word >> (8 * t)) & 0xffu) >= thr; }

// p = 1/2 (thr == 128, the reference's default disc_dropout) uses one BIT per element instead: the word of
// (row, 32-feature tile f>>5) is drop_word(.., row, DROP_BIT_GRP + (f>>5)) and feature f keeps iff bit f&31
// is set -- 8x fewer hashes and a bfe+and per element in the fused kernels.  Every kernel and the test
// helper go through these two functions, so forward, backward and the mask dump agree by construction.
constexpr uint32_t DROP_BIT_GRP = 0x8000u;
MPG_DEV bool drop_keep_f(uint32_t seed_lo, uint32_t seed_hi, uint32_t tag, uint32_t row, int f, uint32_t thr) {
    if (thr == 128u) return (drop_word(seed_lo, seed_hi, tag, row, DROP_BIT_GRP + (uint32_t)(f >> 5)) >> (f & 31)) & 1u;
    return drop_keep(drop_word(seed_lo, seed_hi, tag, row, (uint32_t)(f >> 2)), f & 3, thr);
}
// fused-kernel form: DM = 1 byte mode, DM = 2 bit mode.  `w` = the word of this lane's tile already shifted
// right by 4*h (bit mode) or the word of the 4-feature group (byte mode); `k` = compile-time bit index
// within the lane's view (8g+t in accumulator order, 16s+8u+t in operand order) / byte index t.
template <int DM>
MPG_DEV float drop_apply(float x, uint32_t w, int k_bit, int t_byte, uint32_t thr) {
    if constexpr (DM == 2) {
        int m = __builtin_amdgcn_sbfe((int)w, k_bit, 1);  // 0 or -1
        // (opaque to the optimiser: it would turn "x & mask" into v_and + v_cmp + v_cndmask through an SGPR pair per
        // element -- three instructions instead of two, and enough live SGPR pairs to spill them into VGPR lanes)
        asm("" : "+v"(m));
        return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
    } else if constexpr (DM == 1) {
        return drop_keep(w, t_byte, thr) ? x : 0.f;
    } else {
        return x;
    }
}
// word for accumulator-order group g of tile m (bit mode: one word per tile, CSE'd across g)
template <int DM>
MPG_DEV uint32_t drop_tile_word(uint32_t seed_lo, uint32_t seed_hi, uint32_t tag, uint32_t row, int m, int grp_in_tile, int h) {
    if constexpr (DM == 2) return drop_word(seed_lo, seed_hi, tag, row, DROP_BIT_GRP + (uint32_t)m) >> (4 * h);
    else if constexpr (DM == 1) return drop_word(seed_lo, seed_hi, tag, row, (uint32_t)(8 * m + grp_in_tile));
    else return 0u;
}

// x + (x of another lane) with the other lane picked by a DPP control (quad_perm / row_ror): no LDS traffic,
// unlike __shfl_xor (ds_bpermute).  CTRL: 0xB1 = lane^1, 0x4E = lane^2, 0x124 / 0x128 = rotate the 16-lane row by 4 / 8.
template <int CTRL>
MPG_DEV float dpp_add(float keep, float send) {
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTRL, 0xf, 0xf, true));
}
// One halving step of a many-value lane reduction: the two lanes of a pair both hold partial sums of values
// A and B; the lane with `bit` clear ends up with A's sum over the pair, the other with B's.
template <int CTRL>
MPG_DEV float halve_add(bool bit, float a, float b) {
    return dpp_add<CTRL>(bit ? b : a, bit ? a : b);
}

// dropout sites (tag values); the layer id of the call is mixed in by the host as tag_base
enum { TAG_E0 = 1, TAG_E1 = 2, TAG_E2 = 3, TAG_N0 = 4, TAG_N1 = 5, TAG_N2 = 6, TAG_GENERIC = 7 };

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: remember per (call site =
// kernel instantiation, device) the size already granted.  A process-wide flag would leave every device but the
// first at the 64 KiB default (the reference's nn.DataParallel drives several devices from one process), and the
// launchers may be entered from several autograd worker threads at once -- hence the atomics.
#define MPG_MAX_DEVICES 64
#define MPG_ENSURE_LDS(kernel, bytes)                                                                              \
    do {                                                                                                           \
        static std::atomic<int> _granted[MPG_MAX_DEVICES];                                                         \
        int _dev = 0;                                                                                              \
        HIP_CHECK_RET(hipGetDevice(&_dev));                                                                        \
        _dev &= MPG_MAX_DEVICES - 1;                                                                               \
        if ((int)(bytes) > 64 * 1024 && _granted[_dev].load(std::memory_order_relaxed) < (int)(bytes)) {           \
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                              (int)(bytes)));                                                      \
            _granted[_dev].store((int)(bytes), std::memory_order_relaxed);                                         \
        }                                                                                                          \
    } while (0)

#define HIP_CHECK_RET(expr)                                        \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return (int)_e;                      \
    } while (0)
