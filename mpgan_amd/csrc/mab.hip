// One launch per multihead attention block of GAPT (MAB.forward, gapt/model.py:124-139), for sets of up to 32 tokens (and, further
// down, mab_fwdN_kernel / mab_bwdS_kernel for 33 ... 160):
//
//   q = x Wq' + bq ; k = y Wk' + bk ; v = y Wv' + bv            (nn.MultiheadAttention in-projection, packed [3E, E])
//   P_h = softmax(q_h k_h' / sqrt(d) + key mask) ; o_h = P_h v_h (per head, d = 16)
//   za = x + o Wo' + bo ; z = dropout(za)
//   u = z Wf' + bf ; out = dropout(z + dropout_ff(LeakyReLU(u)))  (MAB.ff = one Linear(E, E) [+ LeakyReLU])
//
// ONE WAVE PER JET; nothing but the weights is read twice and no intermediate leaves the registers.  Everything is a
// chain of 32x32x16 MFMAs in the chain layout of common.h: a tile is [32 features (registers) x 32 tokens (lanes)], and
// registers 8s..8s+7 of a tile ARE the B fragment of k-step s of the next product.  The attention itself stays in that
// layout because A and B fragments of this instruction have the same shape (lane = row or column, 8 k-values):
//   * S' = K Q'    : A = registers of the K tile, B = registers of the Q tile (one k-step = the 16 features of a head);
//                    the result has keys in registers and queries on lanes, so the softmax is a register reduction
//                    plus one exchange between the two lane halves
//   * O' = V' P'   : needs V with keys in registers and features on lanes -- that is the V projection with its two MFMA
//                    operands SWAPPED (activations as A, weights as B), so no transpose is ever made; P's registers are
//                    its B fragments.  The 16 valid rows of the two heads of a 32-feature tile are merged into one
//                    tile, which is again the B operand of the out-projection.
// Products run as fp16 hi/lo 3-term splits (common.h) with power-of-two operand scales; the backward kernel recomputes
// the same values and carries gradients as bf16 hi/lo.
#include "common.h"
#include "../../include/mpgan_amd.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#ifdef MPG_MABSTAMP  // diagnostic build (tools/mab_stamps.py): s_memtime at the phase boundaries, the waves of workgroup 0
__device__ unsigned long long g_mab_stamps[2 * 4 * 8];   // [forward | backward][wave][stamp]
#define MAB_STAMP(i) do { mab_st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define MAB_STAMPP(i) do { if (mab_st != nullptr) mab_st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MAB_STAMP(i) do {} while (0)
#define MAB_STAMPP(i) do {} while (0)
#endif

namespace {

constexpr float MAB_SP = 256.f;   // attention probabilities are split as 256 P (an fp16 lo half stays normal down to P ~ 1e-3)

MPG_DEV float4 mld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Weight images live in LDS for the lifetime of a workgroup (a fragment fetched from L2 is ~1 us away and every product
// of the chain waits for its own): one cooperative copy, then ds_read_b128 per fragment.
typedef const char* WImg;
template <typename V>
MPG_DEV V mab_wfrag(WImg w, int frag, int lane16) {
    return *reinterpret_cast<const V*>(w + frag * 1024 + lane16);
}
// (LDS-DMA: 1 KiB per wave-instruction straight into LDS, no register staging, all of a wave's pieces in flight at once.
// Every workgroup of a launch wants the same image at the same moment; walking it from the same end they would all stand
// at the same few L2 channels, so each starts at a piece of its own; -DMPG_MAB_NOROT builds the A/B.)
MPG_DEV void mab_fill(char* dst, const void* src, int bytes) {
    const int wave = threadIdx.x >> 6, nwav = blockDim.x >> 6, lane = threadIdx.x & 63;
    const int n = bytes / 1024;
#ifdef MPG_MAB_NOROT
    const int rot = 0;
#else
    const int rot = ((int)(blockIdx.x >> 3) * 5) % n;      // (workgroups 8 apart share an XCD and its L2)
#endif
    for (int c = wave; c < n; c += nwav) {
        int cc = c + rot;
        cc = cc >= n ? cc - n : cc;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(src) + cc * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + cc * 1024), 16, 0, 0);
    }
}

// rows of a [*, E] matrix as B (or A) fragments: k-step ks, element j of lane half h = feature 16 ks + 8 (j >> 2) + 4 h + (j & 3)
template <int KS, typename V>
MPG_DEV void rows_to_frags(const float* base, int ld, long row, float scale, int h, V* hi, V* lo) {
    static_for<0, KS>([&](auto kc) {
        MPG_CI(ks, kc);
        const float4 a = mld4(base + row * ld + 16 * ks + 4 * h), b = mld4(base + row * ld + 16 * ks + 8 + 4 * h);
        const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, b.x * scale, b.y * scale, b.z * scale, b.w * scale};
        split8(v, hi[ks], lo[ks]);
    });
}
// rows of a [*, E] matrix as an accumulator-layout tile: register 4g+e = feature 32 tile + 8g + 4h + e of the lane's row
MPG_DEV f32x16 rows_to_tile(const float* base, int ld, long row, int tile, int h) {
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 a = mld4(base + row * ld + 32 * tile + 8 * g + 4 * h);
        t[4 * g] = a.x; t[4 * g + 1] = a.y; t[4 * g + 2] = a.z; t[4 * g + 3] = a.w;
    }
    return t;
}
// ---- Row loads the compiler does not count.  While an LDS-DMA fill is in flight hipcc answers the first use of ANY loaded
// register with s_waitcnt vmcnt(0): a jet's rows, requested ahead of the fill, could only be touched once the whole image had
// landed, and their hi/lo conversion (~1,500 clk of VALU) ran behind the fill instead of under it.  Loaded by inline assembly
// the rows are invisible to that pass; rows_wait<N> is the counted wait -- N = the vector-memory instructions this wave has
// issued SINCE the rows (loads return in issue order) -- and ties the registers to itself so nothing reads them earlier.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
MPG_DEV f32x4 ld4_hidden(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(v) : "v"(p), "n"(OFF) : "memory");
    return v;
}
// the rows of NT tiles: piece 4t + g = features 32t + 8g + 4h .. +3 of the lane's row
template <int NT>
MPG_DEV void rows_request(const float* base, int ld, long row, int h, f32x4* q) {
    const float* const p = base + row * ld + 4 * h;
    static_for<0, 4 * NT>([&](auto ic) {
        MPG_CI(i, ic);
        q[i] = ld4_hidden<(32 * (i / 4) + 8 * (i % 4)) * 4>(p);
    });
}
template <int N>
MPG_DEV void rows_wait(f32x4* q) {     // eight pieces
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "n"(N) : "memory");
}
// ... and eight more pieces that were requested BEFORE a set already waited for (in-order return: they are here too); the empty
// statement only ties the registers to this point
MPG_DEV void rows_pin(f32x4* q) {
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : : "memory");
}
MPG_DEV void rows_pin4(f32x4* q) {
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : : "memory");
}
MPG_DEV f32x16 pieces_tile(const f32x4* q, int t) {
    f32x16 o;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * g + e] = q[4 * t + g][e];
    return o;
}
MPG_DEV void tile_to_rows(float* base, int ld, long row, int tile, int h, const f32x16& t, float scale) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(base + row * ld + 32 * tile + 8 * g + 4 * h) =
            make_float4(t[4 * g] * scale, t[4 * g + 1] * scale, t[4 * g + 2] * scale, t[4 * g + 3] * scale);
}
// registers 8s..8s+7 of a tile, scaled, as one hi/lo fragment pair
template <typename V>
MPG_DEV void tile_frag(const f32x16& t, int s, float scale, V& hi, V& lo) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = t[8 * s + j] * scale;
    split8(v, hi, lo);
}
// all 2 NT fragments of NT row tiles (register 8s+j of tile t = element j of k-step 2t+s: the two layouts hold the same values)
template <int NT, typename V>
MPG_DEV void tiles_to_frags(const f32x16* t, float scale, V* hi, V* lo) {
    static_for<0, NT>([&](auto tc) {
        MPG_CI(tt, tc);
        tile_frag(t[tt], 0, scale, hi[2 * tt], lo[2 * tt]);
        tile_frag(t[tt], 1, scale, hi[2 * tt + 1], lo[2 * tt + 1]);
    });
}
// bias of the features a lane's accumulator registers hold (features in registers); the staged copies are pre-scaled
MPG_DEV f32x16 bias_regs(const float* bias, int tile, int h) {
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 a = mld4(bias + 32 * tile + 8 * g + 4 * h);
        t[4 * g] = a.x; t[4 * g + 1] = a.y; t[4 * g + 2] = a.z; t[4 * g + 3] = a.w;
    }
    return t;
}
// ... and with the features on the lanes (swapped-operand products)
MPG_DEV f32x16 bias_lanes(const float* bias, int tile, int r) {
    const float b = bias[32 * tile + r];
    f32x16 t;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = b;
    return t;
}
// acc += W_tile x  (features of W's row tile `m` in registers, tokens on lanes): W image as A, activation fragments as B
template <int KS, typename V>
MPG_DEV f32x16 proj_n(WImg rw, int nfrag, int m, const V* xh, const V* xl, f32x16 acc, int lane16) {
    static_for<0, KS>([&](auto kc) {
        MPG_CI(ks, kc);
        const V wh = mab_wfrag<V>(rw, m * KS + ks, lane16), wl = mab_wfrag<V>(rw, nfrag + m * KS + ks, lane16);
        acc = mfma3(wh, wl, xh[ks], xl[ks], acc);
    });
    return acc;
}
// the same product with the operands swapped: tokens in registers, W's row tile on the lanes
template <int KS, typename V>
MPG_DEV f32x16 proj_t(WImg rw, int nfrag, int m, const V* xh, const V* xl, f32x16 acc, int lane16) {
    static_for<0, KS>([&](auto kc) {
        MPG_CI(ks, kc);
        const V wh = mab_wfrag<V>(rw, m * KS + ks, lane16), wl = mab_wfrag<V>(rw, nfrag + m * KS + ks, lane16);
        acc = mfma3(xh[ks], xl[ks], wh, wl, acc);
    });
    return acc;
}
MPG_DEV f32x16 zero16() {
    f32x16 t;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.f;
    return t;
}
MPG_DEV float other_half(float v) { return __shfl_xor(v, 32); }

// dropout keep mask times scale on an accumulator-layout tile (the hash of common.h: drop_keep_f(row, feature))
MPG_DEV void drop_tile(f32x16& t, uint32_t seed_lo, uint32_t seed_hi, uint32_t tag, uint32_t row, int tile, int h, uint32_t thr, float scale) {
    if (thr == 0u) return;
    if (thr == 128u) {
        const uint32_t w = drop_word(seed_lo, seed_hi, tag, row, DROP_BIT_GRP + (uint32_t)tile) >> (4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 * g + e] = ((w >> (8 * g + e)) & 1u) ? t[4 * g + e] * scale : 0.f;
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t w = drop_word(seed_lo, seed_hi, tag, row, (uint32_t)(8 * tile + 2 * g + h));
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 * g + e] = drop_keep(w, e, thr) ? t[4 * g + e] * scale : 0.f;
        }
    }
}

// additive key mask for the registers of a score tile with KEYS in registers: register 4g+e = key 8g + 4h + e.  In two halves,
// so that the loads can be issued ahead of a weight fill and the selects run behind it, and WITHOUT a branch on the pointer:
// tested per element, each load sat in a block of its own behind an s_waitcnt vmcnt(0) -- sixteen round trips to memory in
// a row at the head of every kernel; tested once, the join still waited for all of them before the fill was issued.  With no
// mask the sixteen loads read `safe` (any 32 readable floats: the caller passes its x rows) and the values are not looked at.
struct KeyIgn { float v[16]; bool on; };
MPG_DEV KeyIgn key_mask_load(const float* ignore, const float* safe, long jet, int S, int h) {
    KeyIgn k;
    k.on = ignore != nullptr;
    const float* const row = k.on ? ignore + jet * S : safe;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) k.v[4 * g + e] = row[min(8 * g + 4 * h + e, S - 1)];
    return k;
}
MPG_DEV f32x16 key_mask_from(const KeyIgn& k, int S, int h) {
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[4 * g + e] = (8 * g + 4 * h + e >= S || (k.on && k.v[4 * g + e] != 0.f)) ? -INFINITY : 0.f;
    return t;
}
MPG_DEV f32x16 key_mask_regs(const float* ignore, const float* safe, long jet, int S, int h) {
    return key_mask_from(key_mask_load(ignore, safe, jet, S, h), S, h);
}

struct MabScales { float sa, zs, inv_zs; };

// ---- LayerNorm over the E = 32 NT features of a token (gapt/model.py:118-120: nn.LayerNorm(embed_dim), biased variance).  A
// token's features sit in the NT tiles' registers of lanes r and r + 32, so the two sums are in-lane adds and one exchange
// with the other half.  ln_stats: mean and 1 / sqrt(var + eps) (two passes: the variance from the centred values);
// ln_apply: t -> (t - mean) rstd w + b in place.
template <int NT>
MPG_DEV void ln_stats(const f32x16 (&t)[NT], float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += t[tt][i];
    s += other_half(s);
    mean = s * (1.f / (32 * NT));
    float q = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = t[tt][i] - mean; q = fmaf(d, d, q); }
    q += other_half(q);
    rstd = 1.f / sqrtf(q * (1.f / (32 * NT)) + eps);
}
template <int NT>
MPG_DEV void ln_apply(f32x16 (&t)[NT], const float* w, const float* b, float eps, int h) {
    float mean, rstd;
    ln_stats<NT>(t, eps, mean, rstd);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const f32x16 wv = bias_regs(w, tt, h), bv = bias_regs(b, tt, h);
#pragma unroll
        for (int i = 0; i < 16; ++i) t[tt][i] = fmaf((t[tt][i] - mean) * rstd, wv[i], bv[i]);
    }
}

// The half of a block behind the attention, on the 32 rows a wave holds: za = x + o Wo' + bo, [norm1], dropout, the feed-forward
// layer with its residual, [norm2], dropout.  oh / ol: the attention output as B fragments; xt: the rows' input tiles on entry, the
// block's OUTPUT rows on exit.  Shared by the one-wave kernels and the large-set kernel (a wave per tile of 32 queries).
template <int NT, bool LN>
MPG_DEV void mab_post(const MpgMab& p, f32x16 (&xt)[NT], const f16x8* oh, const f16x8* ol, WImg rO, WImg rF, const float* sBo, const float* sBf,
                      const long xrow, const bool xvalid, const uint32_t seed_lo, const uint32_t seed_hi, const float sa, const float inv_zs,
                      const int h, const int lane16, unsigned long long* mab_st) {
    typedef f16x8 V;
    constexpr int KS = 2 * NT;
    constexpr int nfE = NT * KS;
    MAB_STAMPP(3);
    // za = x + o Wo' + bo ; z = dropout(za)
    f32x16 z[NT];
    V zh[KS], zl[KS];
    if constexpr (LN) {
        // ... with norm1 between the residual and the dropout: every tile of za first (kept for the backward), then the norm
        static_for<0, NT>([&](auto tc) {
            MPG_CI(t, tc);
            const f32x16 acc = proj_n<KS>(rO, nfE, t, oh, ol, bias_regs(sBo, t, h), lane16);
#pragma unroll
            for (int i = 0; i < 16; ++i) z[t][i] = acc[i] * inv_zs + xt[t][i];
            if (p.save_za != nullptr && xvalid) tile_to_rows(p.save_za, p.E, xrow, t, h, z[t], 1.f);
        });
        ln_apply<NT>(z, p.ln1_w, p.ln1_b, p.ln_eps, h);
    }
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        if constexpr (!LN) {
            const f32x16 acc = proj_n<KS>(rO, nfE, t, oh, ol, bias_regs(sBo, t, h), lane16);
#pragma unroll
            for (int i = 0; i < 16; ++i) z[t][i] = acc[i] * inv_zs + xt[t][i];
        }
        drop_tile(z[t], seed_lo, seed_hi, p.tag + 0, (uint32_t)xrow, t, h, p.thr_mab, p.sc_mab);
        if (p.save_z != nullptr && xvalid) tile_to_rows(p.save_z, p.E, xrow, t, h, z[t], 1.f);
        tile_frag(z[t], 0, sa, zh[2 * t], zl[2 * t]);
        tile_frag(z[t], 1, sa, zh[2 * t + 1], zl[2 * t + 1]);
    });
    MAB_STAMPP(4);
    // out = dropout([norm2](z + dropout_ff(LeakyReLU(z Wf' + bf))))
    f32x16 op[LN ? NT : 1];
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        f32x16 u = proj_n<KS>(rF, nfE, t, zh, zl, bias_regs(sBf, t, h), lane16);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = u[i] * inv_zs;
            u[i] = p.ff_act ? lrelu(v, p.alpha) : v;
        }
        drop_tile(u, seed_lo, seed_hi, p.tag + 1, (uint32_t)xrow, t, h, p.thr_ff, p.sc_ff);
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] += z[t][i];
        if constexpr (LN) {
            op[t] = u;
        } else {
            drop_tile(u, seed_lo, seed_hi, p.tag + 2, (uint32_t)xrow, t, h, p.thr_mab, p.sc_mab);
            if (xvalid) tile_to_rows(p.out, p.ldo, xrow, t, h, u, 1.f);
            xt[t] = u;     // (the block's output rows, as the next block of a chain takes them; z[t] carried the residual)
        }
    });
    if constexpr (LN) {
        ln_apply<NT>(op, p.ln2_w, p.ln2_b, p.ln_eps, h);
        static_for<0, NT>([&](auto tc) {
            MPG_CI(t, tc);
            drop_tile(op[t], seed_lo, seed_hi, p.tag + 2, (uint32_t)xrow, t, h, p.thr_mab, p.sc_mab);
            if (xvalid) tile_to_rows(p.out, p.ldo, xrow, t, h, op[t], 1.f);
            xt[t] = op[t];
        });
    }
}

// One block on one jet (one wave).  xt: the block's query rows as accumulator-layout tiles on entry, its OUTPUT rows on exit (what
// the next block of a chain of self-attention blocks takes as its input: mab_chain_fwd_kernel); yt: the key / value rows
// (CROSS), kneg the additive key mask; the weight images and the pre-scaled biases are in LDS.
template <int NT, bool CROSS, bool LN = false>
MPG_DEV void mab_fwd_jet(const MpgMab& p, f32x16 (&xt)[NT], const f32x16* yt, const f32x16& kneg, WImg rIn, WImg rO, WImg rF,
                         const float* sBin, const float* sBo, const float* sBf, const long xrow, const bool xvalid,
                         const uint32_t seed_lo, const uint32_t seed_hi, const float sa, const float inv_zs, const int r, const int h,
                         const int lane16, unsigned long long* mab_st) {
    typedef f16x8 V;
    constexpr int KS = 2 * NT;
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS;
    V xh[KS], xl[KS], yh_[CROSS ? KS : 1], yl_[CROSS ? KS : 1];
    tiles_to_frags<NT>(xt, sa, xh, xl);
    if constexpr (CROSS) tiles_to_frags<NT>(yt, sa, yh_, yl_);
    const V* yh = CROSS ? yh_ : xh;
    const V* yl = CROSS ? yl_ : xl;

    MAB_STAMP(2);
    V oh[KS], ol[KS];                     // attention output as B fragments of the out-projection
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        const f32x16 Qn = proj_n<KS>(rIn, nfIn, t, xh, xl, bias_regs(sBin, t, h), lane16);
        const f32x16 Kn = proj_n<KS>(rIn, nfIn, NT + t, yh, yl, bias_regs(sBin, NT + t, h), lane16);
        const f32x16 Vt = proj_t<KS>(rIn, nfIn, 2 * NT + t, yh, yl, bias_lanes(sBin, 2 * NT + t, r), lane16);
        V vh[2], vl[2];
        tile_frag(Vt, 0, inv_zs * sa, vh[0], vl[0]);
        tile_frag(Vt, 1, inv_zs * sa, vh[1], vl[1]);
        f32x16 Ot;
        static_for<0, 2>([&](auto ac) {
            MPG_CI(a, ac);                // head 2t + a = features 16a .. 16a+15 of the tile = registers 8a .. 8a+7
            V qh, ql, kh, kl;
            tile_frag(Qn, a, inv_zs * sa * 0.25f, qh, ql);   // 1/sqrt(d), d = 16
            tile_frag(Kn, a, inv_zs * sa, kh, kl);
            f32x16 s = mfma3(kh, kl, qh, ql, zero16());      // keys in registers, queries on lanes
            const float sc2 = 1.44269504088896341f / (sa * sa);   // scores in the base-2 domain: v_exp_f32 is 2^x
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = s[i] * sc2 + kneg[i]; mx = fmaxf(mx, s[i]); }
            mx = fmaxf(mx, other_half(mx));
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mx); den += s[i]; }
            den += other_half(den);
            const float pn = MAB_SP / den;
            V ph[2], pl[2];
            tile_frag(s, 0, pn, ph[0], pl[0]);
            tile_frag(s, 1, pn, ph[1], pl[1]);
            f32x16 oa = mfma3(vh[0], vl[0], ph[0], pl[0], zero16());
            oa = mfma3(vh[1], vl[1], ph[1], pl[1], oa);
#pragma unroll
            for (int j = 0; j < 8; ++j) Ot[8 * a + j] = oa[8 * a + j];
        });
        // Ot = 256 sa o  (features 32t .. 32t+31 x queries)
        if (p.save_o != nullptr && xvalid) tile_to_rows(p.save_o, p.E, xrow, t, h, Ot, 1.f / (MAB_SP * sa));
        tile_frag(Ot, 0, 1.f / MAB_SP, oh[2 * t], ol[2 * t]);
        tile_frag(Ot, 1, 1.f / MAB_SP, oh[2 * t + 1], ol[2 * t + 1]);
    });

    mab_post<NT, LN>(p, xt, oh, ol, rO, rF, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, h, lane16, mab_st);
    MAB_STAMP(5);
}

template <int NT, bool CROSS, bool LN = false>
__global__ __launch_bounds__(256) void mab_fwd_kernel(const MpgMab p) {
    typedef f16x8 V;
    constexpr int KS = 2 * NT;            // k-steps of 16 over E = 32 NT features
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5, lane16 = lane * 16;
#ifdef MPG_MABSTAMP
    unsigned long long mab_st[8] = {};
    unsigned long long* const mab_stp = mab_st;
#else
    unsigned long long* const mab_stp = nullptr;
#endif
    MAB_STAMP(0);
    // (no branch on the pointer: behind one, the wave waits for the seed before it issues its first load)
    const uint64_t sd = *(p.seed != nullptr ? p.seed : reinterpret_cast<const uint64_t*>(p.x));
    const uint32_t seed_lo = p.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p.ascale > 0.f ? p.ascale : 1.f, ws = p.wscale > 0.f ? p.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs;
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sO = sIn + 2 * nfIn * 1024;
    char* const sF = sO + 2 * nfE * 1024;
    float* const sBin = reinterpret_cast<float*>(sF + 2 * nfE * 1024);   // biases: in_proj [3E] | out_proj [E] | ff [E]
    float* const sBo = sBin + 96 * NT;
    float* const sBf = sBo + 32 * NT;
    // the wave's first jet: its rows are requested BEFORE the weight fill (loads return in issue order; behind 80 KiB of
    // weights per workgroup they would arrive ~3,000 clk later, and their conversion can run while the fill lands)
    const int nw = blockDim.x >> 6;
    const long jet0 = (long)blockIdx.x * nw + w;
    f32x16 xt[NT], yt[CROSS ? NT : 1], kneg;
    auto load_rows = [&](long jet) {
        const long xr = jet * p.L + min(r, p.L - 1), yr = jet * p.S + min(r, p.S - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) xt[t] = rows_to_tile(p.x, p.ldx, xr, t, h);
        if constexpr (CROSS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) yt[t] = rows_to_tile(p.y, p.ldy, yr, t, h);
        }
        kneg = key_mask_regs(p.ignore, p.x, jet, p.S, h);
    };
    load_rows(min(jet0, (long)p.B - 1));
    mab_fill(sIn, p.Win, 2 * nfIn * 1024);
    mab_fill(sO, p.Wo, 2 * nfE * 1024);
    mab_fill(sF, p.Wf, 2 * nfE * 1024);
    for (int i = threadIdx.x; i < 160 * NT; i += blockDim.x)
        sBin[i] = (i < 96 * NT ? p.bin[i] : (i < 128 * NT ? p.bo[i - 96 * NT] : p.bf[i - 128 * NT])) * zs;   // (as the accumulators carry them)
    __syncthreads();
    const WImg rIn = sIn, rO = sO, rF = sF;
    MAB_STAMP(1);
    for (long jet = jet0; jet < p.B; jet += (long)gridDim.x * nw) {   // (no barrier inside)

    // rows past the end of a set are read from its last row and never stored; as keys they are masked
    const long xrow = jet * p.L + min(r, p.L - 1);
    const bool xvalid = r < p.L;
    // every global load of a jet is issued together: x (kept as tiles for the residual), y, the key mask
    if (jet != jet0) load_rows(jet);
    mab_fwd_jet<NT, CROSS, LN>(p, xt, yt, kneg, rIn, rO, rF, sBin, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, r, h, lane16, mab_stp);
    }  // jets of this wave
#ifdef MPG_MABSTAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) g_mab_stamps[w * 8 + i] = mab_st[i];
#endif
}

// ---- LARGE SETS (33 ... 160 tokens: --num-hits 150, gapt/model.py has no size limit).  One workgroup per jet, one wave per TILE of
// 32 queries.  A wave walks the key tiles with a running maximum / sum per (query, head) -- scores of one key tile at a time:
// keys in registers, queries on lanes, exactly the one-wave kernel's products --, projecting each key tile's K and V from its
// rows itself (the rows come out of L2; no exchange between the waves, no barrier behind the weight fill), rescales its output
// accumulators when the maximum moves, and runs the half behind the attention (mab_post) on its own 32 rows.  A query whose keys
// are all masked gets a zero attention output (torch's _safe_softmax).
MPG_DEV f32x16 key_mask_tile(const float* ignore, const float* safe, long jet, int S, int h, int kt) {
    const bool on = ignore != nullptr;
    const float* const row = on ? ignore + jet * S : safe;
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 32 * kt + 8 * g + 4 * h + e;
            const float v = row[on ? min(key, S - 1) : 0];
            t[4 * g + e] = (key >= S || (on && v != 0.f)) ? -INFINITY : 0.f;
        }
    return t;
}

template <int NT, bool LN>
__global__ __launch_bounds__(320) void mab_fwdN_kernel(const MpgMab p) {
    typedef f16x8 V;
    constexpr int KS = 2 * NT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5, lane16 = lane * 16;
    const uint64_t sd = *(p.seed != nullptr ? p.seed : reinterpret_cast<const uint64_t*>(p.x));
    const uint32_t seed_lo = p.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p.ascale > 0.f ? p.ascale : 1.f, ws = p.wscale > 0.f ? p.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs;
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sO = sIn + 2 * nfIn * 1024;
    char* const sF = sO + 2 * nfE * 1024;
    float* const sBin = reinterpret_cast<float*>(sF + 2 * nfE * 1024);
    float* const sBo = sBin + 96 * NT;
    float* const sBf = sBo + 32 * NT;
    const long jet = blockIdx.x;
    const int nqt = (p.L + 31) >> 5, nkt = (p.S + 31) >> 5;
    const int tok = 32 * w + r;
    const long xrow = jet * p.L + min(tok, p.L - 1);
    const bool xvalid = tok < p.L;
    f32x16 xt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xt[t] = rows_to_tile(p.x, p.ldx, xrow, t, h);
    mab_fill(sIn, p.Win, 2 * nfIn * 1024);
    mab_fill(sO, p.Wo, 2 * nfE * 1024);
    mab_fill(sF, p.Wf, 2 * nfE * 1024);
    for (int i = threadIdx.x; i < 160 * NT; i += blockDim.x)
        sBin[i] = (i < 96 * NT ? p.bin[i] : (i < 128 * NT ? p.bo[i - 96 * NT] : p.bf[i - 128 * NT])) * zs;
    __syncthreads();
    const WImg rIn = sIn, rO = sO, rF = sF;
    const bool isq = w < nqt, isk = w < nkt;
    V qh[2 * NT], ql[2 * NT];             // the queries' heads as B fragments: head 2t + a
    if (isq) {
        V xh[KS], xl[KS];
        tiles_to_frags<NT>(xt, sa, xh, xl);
        static_for<0, NT>([&](auto tc) {
            MPG_CI(t, tc);
            const f32x16 Qn = proj_n<KS>(rIn, nfIn, t, xh, xl, bias_regs(sBin, t, h), lane16);
            tile_frag(Qn, 0, inv_zs * sa * 0.25f, qh[2 * t], ql[2 * t]);      // 1/sqrt(d), d = 16
            tile_frag(Qn, 1, inv_zs * sa * 0.25f, qh[2 * t + 1], ql[2 * t + 1]);
        });
    }
    // ---- the key side ONCE per key tile: wave w projects K and V of key tile w, and behind a barrier -- every wave is done with
    // Win's image then -- lays the fragments the attention takes down where that image was (3 tiles' worth) and in 32 KiB of
    // their own behind the biases: [key tile][feature tile][kh0 kl0 kh1 kl1 | vh0 vl0 vh1 vl1][lane] 16 B.  (Projected by every
    // query wave for itself they were 48 of the 84 MFMAs per key tile.)
    constexpr int KVT = 8 * NT * 1024;    // bytes of one key tile's fragments
    auto kv_slot = [&](int kt) -> char* {
        constexpr int WIN_TILES = 3 * NT / 2;   // key tiles whose fragments fit Win's image (12 NT^2 KiB / 8 NT KiB)
        return kt < WIN_TILES ? sIn + kt * KVT : reinterpret_cast<char*>(sBf + 32 * NT) + (kt - WIN_TILES) * KVT;
    };
    {
        V kvf[8 * NT];
        if (isk) {
            const long yrow = jet * p.S + min(32 * w + r, p.S - 1);
            V yh[KS], yl[KS];
            rows_to_frags<KS>(p.y, p.ldy, yrow, sa, h, yh, yl);
            static_for<0, NT>([&](auto tc) {
                MPG_CI(t, tc);
                const f32x16 Kn = proj_n<KS>(rIn, nfIn, NT + t, yh, yl, bias_regs(sBin, NT + t, h), lane16);
                const f32x16 Vt = proj_t<KS>(rIn, nfIn, 2 * NT + t, yh, yl, bias_lanes(sBin, 2 * NT + t, r), lane16);
                tile_frag(Kn, 0, inv_zs * sa, kvf[8 * t + 0], kvf[8 * t + 1]);
                tile_frag(Kn, 1, inv_zs * sa, kvf[8 * t + 2], kvf[8 * t + 3]);
                tile_frag(Vt, 0, inv_zs * sa, kvf[8 * t + 4], kvf[8 * t + 5]);
                tile_frag(Vt, 1, inv_zs * sa, kvf[8 * t + 6], kvf[8 * t + 7]);
            });
        }
        __syncthreads();                  // every projection that reads Win's image is done
        if (isk) {
            V* const dst = reinterpret_cast<V*>(kv_slot(w));
#pragma unroll
            for (int i = 0; i < 8 * NT; ++i) dst[i * 64 + lane] = kvf[i];
        }
        __syncthreads();
    }
    if (!isq) return;                     // (a wave without a query tile: it has helped with the fill and the key side)
    const float sc2 = 1.44269504088896341f / (sa * sa);   // scores in the base-2 domain
    float mrun[2 * NT], den[2 * NT];
    f32x16 oacc[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) { mrun[i] = -INFINITY; den[i] = 0.f; oacc[i] = zero16(); }
    for (int kt = 0; kt < nkt; ++kt) {
        const f32x16 kneg = key_mask_tile(p.ignore, p.x, jet, p.S, h, kt);
        const V* const src = reinterpret_cast<const V*>(kv_slot(kt));
        static_for<0, NT>([&](auto tc) {
            MPG_CI(t, tc);
            const V vh[2] = {src[(8 * t + 4) * 64 + lane], src[(8 * t + 6) * 64 + lane]};
            const V vl[2] = {src[(8 * t + 5) * 64 + lane], src[(8 * t + 7) * 64 + lane]};
            static_for<0, 2>([&](auto ac) {
                MPG_CI(a, ac);
                constexpr int hd = 2 * t + a;
                const V kh = src[(8 * t + 2 * a) * 64 + lane], kl = src[(8 * t + 2 * a + 1) * 64 + lane];
                f32x16 sx = mfma3(kh, kl, qh[hd], ql[hd], zero16());      // keys in registers, queries on lanes
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 16; ++i) { sx[i] = sx[i] * sc2 + kneg[i]; mx = fmaxf(mx, sx[i]); }
                mx = fmaxf(mx, other_half(mx));
                const float mnew = fmaxf(mrun[hd], mx);
                const float msafe = mnew == -INFINITY ? 0.f : mnew;       // (nothing but masked keys so far: every term below is 0)
                const float resc = __builtin_amdgcn_exp2f(mrun[hd] - msafe);
                mrun[hd] = mnew;
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) { sx[i] = __builtin_amdgcn_exp2f(sx[i] - msafe); part += sx[i]; }
                part += other_half(part);
                den[hd] = den[hd] * resc + part;
                V ph[2], pl[2];
                tile_frag(sx, 0, MAB_SP, ph[0], pl[0]);
                tile_frag(sx, 1, MAB_SP, ph[1], pl[1]);
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[hd][i] *= resc;
                oacc[hd] = mfma3(vh[0], vl[0], ph[0], pl[0], oacc[hd]);
                oacc[hd] = mfma3(vh[1], vl[1], ph[1], pl[1], oacc[hd]);
            });
        });
    }
    V oh[KS], ol[KS];
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        f32x16 Ot;
        static_for<0, 2>([&](auto ac) {
            MPG_CI(a, ac);
            constexpr int hd = 2 * t + a;
            const float inv = den[hd] > 0.f ? 1.f / den[hd] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) Ot[8 * a + j] = oacc[hd][8 * a + j] * inv;
        });
        // Ot = 256 sa o  (features 32t .. 32t+31 x queries)
        if (p.save_o != nullptr && xvalid) tile_to_rows(p.save_o, p.E, xrow, t, h, Ot, 1.f / (MAB_SP * sa));
        tile_frag(Ot, 0, 1.f / MAB_SP, oh[2 * t], ol[2 * t]);
        tile_frag(Ot, 1, 1.f / MAB_SP, oh[2 * t + 1], ol[2 * t + 1]);
    });
    mab_post<NT, LN>(p, xt, oh, ol, rO, rF, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, h, lane16, nullptr);
}

// A CHAIN of self-attention blocks (the SABs of a GAPT network, gapt/model.py:261-262 / :341-342: x = sab(x, mask) in a loop) in
// one launch: a wave keeps its jet's rows in registers from block to block -- only the weights change (one cooperative refill
// of the LDS images per block, between two barriers) -- so a block's launch, its prologue and the round trip of its input rows
// through memory are paid once per chain.  Every block still writes what its backward needs (out = the next block's x,
// save_o, save_z) as the single launches do, with its own dropout sites: bit-identical to them.  One jet per wave.
template <int NT>
__global__ __launch_bounds__(256) void mab_chain_fwd_kernel(const MpgMabChain c) {
    constexpr int KS = 2 * NT;
    const MpgMab& p0 = c.blk[0];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5, lane16 = lane * 16;
    // (no branch on the pointer: behind one, the wave waits for the seed before it issues its first load)
    const uint64_t sd = *(p0.seed != nullptr ? p0.seed : reinterpret_cast<const uint64_t*>(p0.x));
    const uint32_t seed_lo = p0.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p0.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p0.ascale > 0.f ? p0.ascale : 1.f, ws = p0.wscale > 0.f ? p0.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs;
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sO = sIn + 2 * nfIn * 1024;
    char* const sF = sO + 2 * nfE * 1024;
    float* const sBin = reinterpret_cast<float*>(sF + 2 * nfE * 1024);
    float* const sBo = sBin + 96 * NT;
    float* const sBf = sBo + 32 * NT;
    const int nw = blockDim.x >> 6;
    const long jet = (long)blockIdx.x * nw + w;
    const bool live = jet < p0.B;                      // (a wave without a jet still takes part in the fills and barriers)
    const long jc = live ? jet : (long)p0.B - 1;
    const long xrow = jc * p0.L + min(r, p0.L - 1);
    const bool xvalid = live && r < p0.L;
    f32x16 xt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xt[t] = rows_to_tile(p0.x, p0.ldx, xrow, t, h);
    const f32x16 kneg = key_mask_regs(p0.ignore, p0.x, jc, p0.S, h);
    static_for<0, MPG_MAB_CHAIN_MAX>([&](auto bc) {     // (compile-time index: a run-time one would move the argument block to scratch)
        MPG_CI(b, bc);
        if (b < c.n) {
            const MpgMab& p = c.blk[b];
            if (b > 0) __syncthreads();                 // every wave is done with the images of the block before
            mab_fill(sIn, p.Win, 2 * nfIn * 1024);
            mab_fill(sO, p.Wo, 2 * nfE * 1024);
            mab_fill(sF, p.Wf, 2 * nfE * 1024);
            for (int i = threadIdx.x; i < 160 * NT; i += blockDim.x)
                sBin[i] = (i < 96 * NT ? p.bin[i] : (i < 128 * NT ? p.bo[i - 96 * NT] : p.bf[i - 128 * NT])) * zs;
            __syncthreads();
            mab_fwd_jet<NT, false>(p, xt, xt, kneg, sIn, sO, sF, sBin, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, r, h, lane16, nullptr);
        }
    });
}

// ---- TWO WAVES PER JET (E = 64, self-attention).  A block is a dependent chain of ~150 MFMAs and the hi/lo splits between
// them on ONE wave, and a launch of 512 jets leaves half of the chip's SIMDs without a wave.  Here wave `T` of a pair owns
// feature tile T of every tensor of the block -- heads 2T and 2T + 1 of the attention (which never look at the other heads),
// tile T of the out-projection, of the feed-forward layer and of the output -- and the pair meets twice per block in LDS:
// the attention output and z are needed by both as B fragments of the next product (each writes its two k-steps, reads the
// partner's two).  In a chain the output rows go across a third time, as the next block's x fragments.  Half the MFMAs, half
// the splits and half the softmax per wave; same arithmetic per element, hence the same bits as the one-wave form.
constexpr int MAB_XCH = 4 * 2 * 1024;   // one exchange buffer of a pair: [k-step][hi | lo][lane] 16 B

template <int T>
MPG_DEV void mab_fwd_half(const MpgMab& p, f16x8* xh, f16x8* xl, const f16x8* yh, const f16x8* yl, f32x16& xtile, const f32x16& kneg, WImg rIn, WImg rO, WImg rF,
                          const float* sBin, const float* sBo, const float* sBf, const long xrow, const bool xvalid,
                          const uint32_t seed_lo, const uint32_t seed_hi, const float sa, const float inv_zs, const int r, const int h,
                          const int lane, char* xchA, char* xchB, const bool next_x, unsigned long long* mab_st = nullptr) {
    typedef f16x8 V;
    constexpr int NT = 2, KS = 4, nfIn = 3 * NT * KS, nfE = NT * KS, O = 1 - T;
    const int lane16 = lane * 16;
    V* const fa = reinterpret_cast<V*>(xchA);
    V* const fb = reinterpret_cast<V*>(xchB);
    auto put = [&](V* f, int ks, const V& hi, const V& lo) { f[(ks * 2 + 0) * 64 + lane] = hi; f[(ks * 2 + 1) * 64 + lane] = lo; };
    auto get = [&](const V* f, int ks, V& hi, V& lo) { hi = f[(ks * 2 + 0) * 64 + lane]; lo = f[(ks * 2 + 1) * 64 + lane]; };
    V oh[KS], ol[KS];
    {
        const f32x16 Qn = proj_n<KS>(rIn, nfIn, T, xh, xl, bias_regs(sBin, T, h), lane16);
        const f32x16 Kn = proj_n<KS>(rIn, nfIn, NT + T, yh, yl, bias_regs(sBin, NT + T, h), lane16);
        const f32x16 Vt = proj_t<KS>(rIn, nfIn, 2 * NT + T, yh, yl, bias_lanes(sBin, 2 * NT + T, r), lane16);
        V vh[2], vl[2];
        tile_frag(Vt, 0, inv_zs * sa, vh[0], vl[0]);
        tile_frag(Vt, 1, inv_zs * sa, vh[1], vl[1]);
        MAB_STAMPP(2);
        f32x16 Ot;
        static_for<0, 2>([&](auto ac) {
            MPG_CI(a, ac);
            V qh, ql, kh, kl;
            tile_frag(Qn, a, inv_zs * sa * 0.25f, qh, ql);
            tile_frag(Kn, a, inv_zs * sa, kh, kl);
            f32x16 sc = mfma3(kh, kl, qh, ql, zero16());
            const float sc2 = 1.44269504088896341f / (sa * sa);
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sc[i] = sc[i] * sc2 + kneg[i]; mx = fmaxf(mx, sc[i]); }
            mx = fmaxf(mx, other_half(mx));
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sc[i] = __builtin_amdgcn_exp2f(sc[i] - mx); den += sc[i]; }
            den += other_half(den);
            const float pn = MAB_SP / den;
            V ph[2], pl[2];
            tile_frag(sc, 0, pn, ph[0], pl[0]);
            tile_frag(sc, 1, pn, ph[1], pl[1]);
            f32x16 oa = mfma3(vh[0], vl[0], ph[0], pl[0], zero16());
            oa = mfma3(vh[1], vl[1], ph[1], pl[1], oa);
#pragma unroll
            for (int j = 0; j < 8; ++j) Ot[8 * a + j] = oa[8 * a + j];
        });
        if (p.save_o != nullptr && xvalid) tile_to_rows(p.save_o, p.E, xrow, T, h, Ot, 1.f / (MAB_SP * sa));
        tile_frag(Ot, 0, 1.f / MAB_SP, oh[2 * T], ol[2 * T]);
        tile_frag(Ot, 1, 1.f / MAB_SP, oh[2 * T + 1], ol[2 * T + 1]);
    }
    MAB_STAMPP(3);
    put(fa, 2 * T, oh[2 * T], ol[2 * T]);
    put(fa, 2 * T + 1, oh[2 * T + 1], ol[2 * T + 1]);
    __syncthreads();
    get(fa, 2 * O, oh[2 * O], ol[2 * O]);
    get(fa, 2 * O + 1, oh[2 * O + 1], ol[2 * O + 1]);
    MAB_STAMPP(4);
    // za = x + o Wo' + bo ; z = dropout(za): tile T
    f32x16 z;
    V zh[KS], zl[KS];
    {
        const f32x16 acc = proj_n<KS>(rO, nfE, T, oh, ol, bias_regs(sBo, T, h), lane16);
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = acc[i] * inv_zs + xtile[i];
        drop_tile(z, seed_lo, seed_hi, p.tag + 0, (uint32_t)xrow, T, h, p.thr_mab, p.sc_mab);
        if (p.save_z != nullptr && xvalid) tile_to_rows(p.save_z, p.E, xrow, T, h, z, 1.f);
        tile_frag(z, 0, sa, zh[2 * T], zl[2 * T]);
        tile_frag(z, 1, sa, zh[2 * T + 1], zl[2 * T + 1]);
    }
    MAB_STAMPP(5);
    put(fb, 2 * T, zh[2 * T], zl[2 * T]);
    put(fb, 2 * T + 1, zh[2 * T + 1], zl[2 * T + 1]);
    __syncthreads();
    get(fb, 2 * O, zh[2 * O], zl[2 * O]);
    get(fb, 2 * O + 1, zh[2 * O + 1], zl[2 * O + 1]);
    MAB_STAMPP(6);
    // out = dropout(z + dropout_ff(LeakyReLU(z Wf' + bf))): tile T
    f32x16 u = proj_n<KS>(rF, nfE, T, zh, zl, bias_regs(sBf, T, h), lane16);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = u[i] * inv_zs;
        u[i] = p.ff_act ? lrelu(v, p.alpha) : v;
    }
    drop_tile(u, seed_lo, seed_hi, p.tag + 1, (uint32_t)xrow, T, h, p.thr_ff, p.sc_ff);
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] += z[i];
    drop_tile(u, seed_lo, seed_hi, p.tag + 2, (uint32_t)xrow, T, h, p.thr_mab, p.sc_mab);
    if (xvalid) tile_to_rows(p.out, p.ldo, xrow, T, h, u, 1.f);
    xtile = u;
    MAB_STAMPP(7);
    if (next_x) {   // the next block's x fragments: own k-steps from the registers, the partner's through the first buffer
        tile_frag(u, 0, sa, xh[2 * T], xl[2 * T]);
        tile_frag(u, 1, sa, xh[2 * T + 1], xl[2 * T + 1]);
        put(fa, 2 * T, xh[2 * T], xl[2 * T]);          // (the partner read its o fragments before the second barrier)
        put(fa, 2 * T + 1, xh[2 * T + 1], xl[2 * T + 1]);
        __syncthreads();
        get(fa, 2 * O, xh[2 * O], xl[2 * O]);
        get(fa, 2 * O + 1, xh[2 * O + 1], xl[2 * O + 1]);
    }
}

// the chain of self-attention blocks with two waves per jet; a workgroup of four waves carries two jets
template <int NW>   // waves of a workgroup: 4 (two jets) or 8 (four jets, two waves to a SIMD)
__global__ __launch_bounds__(64 * NW) void mab_chain_fwd2_kernel(const MpgMabChain c) {
    typedef f16x8 V;
    constexpr int NT = 2, KS = 4;
    const MpgMab& p0 = c.blk[0];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pair = w >> 1, role = w & 1;
    // (no branch on the pointer: behind one, the wave waits for the seed before it issues its first load)
    const uint64_t sd = *(p0.seed != nullptr ? p0.seed : reinterpret_cast<const uint64_t*>(p0.x));
    const uint32_t seed_lo = p0.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p0.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p0.ascale > 0.f ? p0.ascale : 1.f, ws = p0.wscale > 0.f ? p0.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs;
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sO = sIn + 2 * nfIn * 1024;
    char* const sF = sO + 2 * nfE * 1024;
    float* const sBin = reinterpret_cast<float*>(sF + 2 * nfE * 1024);
    float* const sBo = sBin + 96 * NT;
    float* const sBf = sBo + 32 * NT;
    char* const xchA = reinterpret_cast<char*>(sBf + 32 * NT) + pair * 2 * MAB_XCH;
    char* const xchB = xchA + MAB_XCH;
    const int npair = blockDim.x >> 7;
    const long jet = (long)blockIdx.x * npair + pair;
    const bool live = jet < p0.B;                      // (a pair without a jet still takes part in the fills and barriers)
    const long jc = live ? jet : (long)p0.B - 1;
    const long xrow = jc * p0.L + min(r, p0.L - 1);
    const bool xvalid = live && r < p0.L;
    // order of issue: a block's biases and (first block) the key mask -- plain loads, used last --, the jet's rows (loads
    // the compiler does not count: rows_request), the fill; the rows are converted UNDER the fill, which takes the CU ~2,900 clk
    // to issue alone (tools/ubench/fill_rate.hip)
    V xh[KS], xl[KS];
    f32x16 xtile, kneg;
    const KeyIgn kig = key_mask_load(p0.ignore, p0.x, jc, p0.S, h);
    f32x4 xq[4 * NT];
    const int bi0 = min((int)threadIdx.x, 160 * NT - 1), bi1 = min((int)threadIdx.x + 256, 160 * NT - 1);
    static_assert(NT == 2 && (NW == 4 || NW == 8), "rows_wait counts the fill of an E = 64 block on NW waves");
    static_for<0, MPG_MAB_CHAIN_MAX>([&](auto bc) {
        MPG_CI(b, bc);
        if (b < c.n) {
            const MpgMab& p = c.blk[b];
            auto bias_at = [&](int i) { return i < 96 * NT ? p.bin[i] : (i < 128 * NT ? p.bo[i - 96 * NT] : p.bf[i - 128 * NT]); };
            const float bv0 = bias_at(bi0), bv1 = bias_at(bi1);
            if (b == 0) rows_request<NT>(p0.x, p0.ldx, xrow, h, xq);
            if (b > 0) __syncthreads();                 // every wave is done with the images of the block before
            mab_fill(sIn, p.Win, 2 * nfIn * 1024);
            mab_fill(sO, p.Wo, 2 * nfE * 1024);
            mab_fill(sF, p.Wf, 2 * nfE * 1024);
            if (b == 0) {
                rows_wait<80 / NW>(xq);
                f32x16 xt[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) xt[t] = pieces_tile(xq, t);
                tiles_to_frags<NT>(xt, sa, xh, xl);
                xtile = role == 0 ? xt[0] : xt[1];
            }
            if (threadIdx.x < 160 * NT) sBin[bi0] = bv0 * zs;
            if (NW == 4 && threadIdx.x + 256 < 160 * NT) sBin[bi1] = bv1 * zs;
            if (b == 0) kneg = key_mask_from(kig, p0.S, h);
            __syncthreads();
            const bool nx = b + 1 < c.n;
            if (role == 0) mab_fwd_half<0>(p, xh, xl, xh, xl, xtile, kneg, sIn, sO, sF, sBin, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, r, h, lane, xchA, xchB, nx);
            else mab_fwd_half<1>(p, xh, xl, xh, xl, xtile, kneg, sIn, sO, sF, sBin, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, r, h, lane, xchA, xchB, nx);
        }
    });
}

// one block with two waves per jet, self- or cross-attention (the key / value rows y as a second set of fragments)
template <bool CROSS, int NW>
__global__ __launch_bounds__(64 * NW) void mab_fwd2_kernel(const MpgMab p) {
    typedef f16x8 V;
    constexpr int NT = 2, KS = 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pair = w >> 1, role = w & 1;
#ifdef MPG_MABSTAMP
    unsigned long long mab_stv[8] = {};
    unsigned long long* const mab_st = mab_stv;
#else
    unsigned long long* const mab_st = nullptr;
#endif
    MAB_STAMPP(0);
    // (no branch on the pointer: behind one, the wave waits for the seed before it issues its first load)
    const uint64_t sd = *(p.seed != nullptr ? p.seed : reinterpret_cast<const uint64_t*>(p.x));
    const uint32_t seed_lo = p.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p.ascale > 0.f ? p.ascale : 1.f, ws = p.wscale > 0.f ? p.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs;
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sO = sIn + 2 * nfIn * 1024;
    char* const sF = sO + 2 * nfE * 1024;
    float* const sBin = reinterpret_cast<float*>(sF + 2 * nfE * 1024);
    float* const sBo = sBin + 96 * NT;
    float* const sBf = sBo + 32 * NT;
    char* const xchA = reinterpret_cast<char*>(sBf + 32 * NT) + pair * 2 * MAB_XCH;
    char* const xchB = xchA + MAB_XCH;
    const int npair = blockDim.x >> 7;
    const long jet = (long)blockIdx.x * npair + pair;
    const bool live = jet < p.B;
    const long jc = live ? jet : (long)p.B - 1;
    const long xrow = jc * p.L + min(r, p.L - 1), yrow = jc * p.S + min(r, p.S - 1);
    const bool xvalid = live && r < p.L;
    // rows and key mask requested ahead of the weight fill, converted behind it (see mab_chain_fwd2_kernel)
    V xh[KS], xl[KS], yh_[CROSS ? KS : 1], yl_[CROSS ? KS : 1];
    f32x16 xtile;
    // order of issue: biases and key mask (plain loads, used last), the rows (uncounted loads), the fill; then the rows are
    // converted UNDER the fill (20 LDS-DMA instructions per wave behind them: rows_wait<20>)
    const int bi0 = min((int)threadIdx.x, 160 * NT - 1), bi1 = min((int)threadIdx.x + 256, 160 * NT - 1);
    auto bias_at = [&](int i) { return i < 96 * NT ? p.bin[i] : (i < 128 * NT ? p.bo[i - 96 * NT] : p.bf[i - 128 * NT]); };
    const float bv0 = bias_at(bi0), bv1 = bias_at(bi1);
    const KeyIgn kig = key_mask_load(p.ignore, p.x, jc, p.S, h);
    f32x4 xq[4 * NT], yq[CROSS ? 4 * NT : 1];
    if constexpr (CROSS) rows_request<NT>(p.y, p.ldy, yrow, h, yq);
    rows_request<NT>(p.x, p.ldx, xrow, h, xq);
    static_assert(NT == 2 && (NW == 4 || NW == 8), "rows_wait counts the fill of an E = 64 block on NW waves");
    mab_fill(sIn, p.Win, 2 * nfIn * 1024);
    mab_fill(sO, p.Wo, 2 * nfE * 1024);
    mab_fill(sF, p.Wf, 2 * nfE * 1024);
    f32x16 xt[NT], yt[CROSS ? NT : 1];
    if constexpr (CROSS) {
        rows_wait<80 / NW + 4 * NT>(yq);
#pragma unroll
        for (int t = 0; t < NT; ++t) yt[t] = pieces_tile(yq, t);
        tiles_to_frags<NT>(yt, sa, yh_, yl_);
    }
    rows_wait<80 / NW>(xq);
#pragma unroll
    for (int t = 0; t < NT; ++t) xt[t] = pieces_tile(xq, t);
    tiles_to_frags<NT>(xt, sa, xh, xl);
    xtile = role == 0 ? xt[0] : xt[1];
    const V* yh = CROSS ? yh_ : xh;
    const V* yl = CROSS ? yl_ : xl;
    if (threadIdx.x < 160 * NT) sBin[bi0] = bv0 * zs;
    if (NW == 4 && threadIdx.x + 256 < 160 * NT) sBin[bi1] = bv1 * zs;
    const f32x16 kneg = key_mask_from(kig, p.S, h);
    __syncthreads();
    MAB_STAMPP(1);
    if (role == 0) mab_fwd_half<0>(p, xh, xl, yh, yl, xtile, kneg, sIn, sO, sF, sBin, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, r, h, lane, xchA, xchB, false, mab_st);
    else mab_fwd_half<1>(p, xh, xl, yh, yl, xtile, kneg, sIn, sO, sF, sBin, sBo, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, r, h, lane, xchA, xchB, false, mab_st);
#ifdef MPG_MABSTAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) g_mab_stamps[w * 8 + i] = mab_stv[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the block, one wave per jet again.  q, k, v, P and u are recomputed from x, y and the saved z with the
// forward's own arithmetic; gradients run as bf16 hi/lo products.  With S' = K Q' (keys in registers) the softmax
// statistics are per lane, and everything that contracts over KEYS follows from it directly:
//     dP = V dO'   (A = V tile registers, B = dO tile registers)      dS = P (dP - sum_keys dP P) / 4
//     dQ' = K' dS  (A = the K projection with swapped operands: features on lanes, keys in registers)
// What contracts over QUERIES needs the transposed tiles, and those are the SAME MFMAs with A and B exchanged:
//     S^T = Q K'   -> P^T = exp(S^T - c_query), c = max + log(sum) fetched per register from the lane that owns the query
//     dP^T = dO V' -> dS^T ;  dK' = Q' dS^T  (A = swapped-operand Q projection) ;  dV' = dO' P^T  (A = swapped-operand dO)
// so no tile is ever transposed through memory.  Rows of x past L carry a zero output gradient and keys past S or
// masked have P = 0, so padding contributes nothing.
template <int NT, typename V>
MPG_DEV void acc_wt(WImg rT, int nfragT, int KST, int ks0, const V* fh, const V* fl, f32x16* acc, int lane16) {
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        static_for<0, 2>([&](auto sc) {
            MPG_CI(s, sc);
            const V wh = mab_wfrag<V>(rT, t * KST + ks0 + s, lane16), wl = mab_wfrag<V>(rT, nfragT + t * KST + ks0 + s, lane16);
            acc[t] = mfma3(wh, wl, fh[s], fl[s], acc[t]);
        });
    });
}

// LayerNorm backward for the tokens of a wave, in place on the gradient tiles: on entry g = dL/d(norm output), xin = the
// norm's INPUT (its statistics are recomputed); the rows of g and of g * xhat go to dn / gn (their column sums are the norm's
// bias and weight gradients: the grouped weight-gradient launch adds them up); on exit g = dL/d(norm input):
//   rstd (g w - mean_f(g w) - xhat mean_f(g w xhat)).
template <int NT>
MPG_DEV void ln_backward(f32x16 (&g)[NT], const f32x16 (&xin)[NT], const float* w, float eps, float* dn, float* gn, int E,
                         long row, bool valid, int h) {
    float mean, rstd;
    ln_stats<NT>(xin, eps, mean, rstd);
    float m1 = 0.f, m2 = 0.f;
    f32x16 xh[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x16 wv = bias_regs(w, t, h);
        f32x16 gx;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            xh[t][i] = (xin[t][i] - mean) * rstd;
            gx[i] = g[t][i] * xh[t][i];
        }
        if (dn != nullptr && valid) { tile_to_rows(dn, E, row, t, h, g[t], 1.f); tile_to_rows(gn, E, row, t, h, gx, 1.f); }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            g[t][i] *= wv[i];
            m1 += g[t][i];
            m2 = fmaf(g[t][i], xh[t][i], m2);
        }
    }
    m1 += other_half(m1); m2 += other_half(m2);
    m1 *= 1.f / (32 * NT); m2 *= 1.f / (32 * NT);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) g[t][i] = rstd * (g[t][i] - m1 - xh[t][i] * m2);
}

// The feed-forward half of a block's backward on the 32 rows a wave holds: dzf = dropout'(dout) [through norm2] ;
// du = dzf drop_ff' act'(u) ; dz = dzf + du Wf ; dza = dropout'(dz) [through norm1].  In: dzf = the rows of dout, zt = the saved z,
// zat = the saved za (LN).  Out: du / dza rows (p.du, p.dza), dza as bf16 fragments, dxa = dza (the residual path of dx).
template <int NT, bool LN>
MPG_DEV void mab_bwd_ff(const MpgMab& p, f32x16 (&dzf)[NT], const f32x16 (&zt)[NT], const f32x16* zat, const float xlive, WImg rF, WImg rFT,
                        const float* sBf, const long xrow, const bool xvalid, const uint32_t seed_lo, const uint32_t seed_hi, const float sa,
                        const float inv_zs, const int h, const int lane16, bf16x8* dzah, bf16x8* dzal, f32x16 (&dxa)[NT]) {
    typedef f16x8 VF;
    typedef bf16x8 VB;
    constexpr int KS = 2 * NT;
    constexpr int nfE = NT * KS;
    {
        VF zh[KS], zl[KS];
        tiles_to_frags<NT>(zt, sa, zh, zl);
        VB duh[KS], dul[KS];
        f32x16 ut[LN ? NT : 1];
        if constexpr (LN) {
            // norm2 sits between the second residual and the last dropout: its input z + dropout_ff(act(u)) is rebuilt as the
            // forward built it (u is needed below anyway), the gradient goes through the norm, and dzf is then what it is
            // without a norm -- the gradient with respect to that sum
            f32x16 op[NT];
            static_for<0, NT>([&](auto tc) {
                MPG_CI(t, tc);
#pragma unroll
                for (int i = 0; i < 16; ++i) dzf[t][i] *= xlive;
                drop_tile(dzf[t], seed_lo, seed_hi, p.tag + 2, (uint32_t)xrow, t, h, p.thr_mab, p.sc_mab);
                ut[t] = proj_n<KS>(rF, nfE, t, zh, zl, bias_regs(sBf, t, h), lane16);
                f32x16 a;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = ut[t][i] * inv_zs;
                    a[i] = p.ff_act ? lrelu(v, p.alpha) : v;
                }
                drop_tile(a, seed_lo, seed_hi, p.tag + 1, (uint32_t)xrow, t, h, p.thr_ff, p.sc_ff);
#pragma unroll
                for (int i = 0; i < 16; ++i) op[t][i] = zt[t][i] + a[i];
            });
            ln_backward<NT>(dzf, op, p.ln2_w, p.ln_eps, p.dn2, p.gn2, p.E, xrow, xvalid, h);
        }
        static_for<0, NT>([&](auto tc) {
            MPG_CI(t, tc);
            f32x16 u;
            if constexpr (LN) {
                u = ut[t];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) dzf[t][i] *= xlive;
                drop_tile(dzf[t], seed_lo, seed_hi, p.tag + 2, (uint32_t)xrow, t, h, p.thr_mab, p.sc_mab);
                u = proj_n<KS>(rF, nfE, t, zh, zl, bias_regs(sBf, t, h), lane16);
            }
            f32x16 du;
#pragma unroll
            for (int i = 0; i < 16; ++i) du[i] = dzf[t][i] * ((p.ff_act && !(u[i] > 0.f)) ? p.alpha : 1.f);
            drop_tile(du, seed_lo, seed_hi, p.tag + 1, (uint32_t)xrow, t, h, p.thr_ff, p.sc_ff);
            if (p.du != nullptr && xvalid) tile_to_rows(p.du, p.E, xrow, t, h, du, 1.f);
            tile_frag(du, 0, 1.f, duh[2 * t], dul[2 * t]);
            tile_frag(du, 1, 1.f, duh[2 * t + 1], dul[2 * t + 1]);
        });
        if constexpr (LN) {
            // dz of every tile, through the first dropout: the gradient with respect to norm1's output; through the norm (its
            // input za was kept by the forward); what comes out is dza
            f32x16 dn[NT];
            static_for<0, NT>([&](auto tc) {
                MPG_CI(t, tc);
                dn[t] = proj_n<KS>(rFT, nfE, t, duh, dul, dzf[t], lane16);
                drop_tile(dn[t], seed_lo, seed_hi, p.tag + 0, (uint32_t)xrow, t, h, p.thr_mab, p.sc_mab);
            });
            ln_backward<NT>(dn, *reinterpret_cast<const f32x16 (*)[NT]>(zat), p.ln1_w, p.ln_eps, p.dn1, p.gn1, p.E, xrow, xvalid, h);
            static_for<0, NT>([&](auto tc) {
                MPG_CI(t, tc);
                if (p.dza != nullptr && xvalid) tile_to_rows(p.dza, p.E, xrow, t, h, dn[t], 1.f);
                tile_frag(dn[t], 0, 1.f, dzah[2 * t], dzal[2 * t]);
                tile_frag(dn[t], 1, 1.f, dzah[2 * t + 1], dzal[2 * t + 1]);
                dxa[t] = dn[t];
            });
        } else {
            static_for<0, NT>([&](auto tc) {
                MPG_CI(t, tc);
                f32x16 dz = proj_n<KS>(rFT, nfE, t, duh, dul, dzf[t], lane16);
                drop_tile(dz, seed_lo, seed_hi, p.tag + 0, (uint32_t)xrow, t, h, p.thr_mab, p.sc_mab);
                if (p.dza != nullptr && xvalid) tile_to_rows(p.dza, p.E, xrow, t, h, dz, 1.f);
                tile_frag(dz, 0, 1.f, dzah[2 * t], dzal[2 * t]);
                tile_frag(dz, 1, 1.f, dzah[2 * t + 1], dzal[2 * t + 1]);
                dxa[t] = dz;
            });
        }
    }
}

template <int NT, bool CROSS, bool LN = false>
__global__ __launch_bounds__(256) void mab_bwd_kernel(const MpgMab p) {
    typedef f16x8 VF;
    typedef bf16x8 VB;
    constexpr int KS = 2 * NT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5, lane16 = lane * 16;
#ifdef MPG_MABSTAMP
    unsigned long long mab_st[8] = {};
#endif
    MAB_STAMP(0);
    // (no branch on the pointer: behind one, the wave waits for the seed before it issues its first load)
    const uint64_t sd = *(p.seed != nullptr ? p.seed : reinterpret_cast<const uint64_t*>(p.x));
    const uint32_t seed_lo = p.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p.ascale > 0.f ? p.ascale : 1.f, ws = p.wscale > 0.f ? p.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs, sc2 = 1.44269504088896341f / (sa * sa);   // (scores in the base-2 domain)
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS, nfInT = NT * 3 * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sF = sIn + 2 * nfIn * 1024;
    char* const sInT = sF + 2 * nfE * 1024;
    char* const sOT = sInT + 2 * nfInT * 1024;
    char* const sFT = sOT + 2 * nfE * 1024;
    // ONE jet per wave (the launcher sizes the grid for it).  Its dout and z rows -- all the feed-forward half needs --
    // are requested BEFORE the 144 KiB weight fill (loads return in issue order: behind the fill they arrived ~8,000 clk
    // late); x and y follow the fill and land while that half runs.
    const int nw = blockDim.x >> 6;
    const long jet_raw = (long)blockIdx.x * nw + w;
    const long jet = min(jet_raw, (long)p.B - 1);
    const long xrow = jet * p.L + min(r, p.L - 1), yrow = jet * p.S + min(r, p.S - 1);
    f32x16 dzf[NT], zt[NT], zat[LN ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        dzf[t] = rows_to_tile(p.dout, p.lddout, xrow, t, h);
        zt[t] = rows_to_tile(p.save_z, p.E, xrow, t, h);
        if constexpr (LN) zat[t] = rows_to_tile(p.save_za, p.E, xrow, t, h);
    }
    mab_fill(sIn, p.Win, 2 * nfIn * 1024);
    mab_fill(sF, p.Wf, 2 * nfE * 1024);
    mab_fill(sInT, p.WinT, 2 * nfInT * 1024);
    mab_fill(sOT, p.WoT, 2 * nfE * 1024);
    mab_fill(sFT, p.WfT, 2 * nfE * 1024);
    float* const sBin = reinterpret_cast<float*>(sFT + 2 * nfE * 1024);  // biases: in_proj [3E] | ff [E]
    float* const sBf = sBin + 96 * NT;
    for (int i = threadIdx.x; i < 128 * NT; i += blockDim.x) sBin[i] = (i < 96 * NT ? p.bin[i] : p.bf[i - 96 * NT]) * zs;
    __syncthreads();
    const WImg rIn = sIn, rF = sF, rInT = sInT, rOT = sOT, rFT = sFT;
    MAB_STAMP(1);
    if (jet_raw >= p.B) return;           // (a wave without a jet: it has helped with the fill)
    {
    const bool xvalid = r < p.L, yvalid = r < p.S;
    const float xlive = xvalid ? 1.f : 0.f;
    const f32x16 kneg = key_mask_regs(p.ignore, p.x, jet, p.S, h);
    // this lane as a KEY (transposed tiles); no branch on the pointer, as in key_mask_load
    const float ign_r = (p.ignore != nullptr ? p.ignore + jet * p.S : p.x)[min(r, p.S - 1)];
    const bool key_off = !yvalid || (p.ignore != nullptr && ign_r != 0.f);
    f32x16 xt[NT], yt[CROSS ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        xt[t] = rows_to_tile(p.x, p.ldx, xrow, t, h);
        if constexpr (CROSS) yt[t] = rows_to_tile(p.y, p.ldy, yrow, t, h);
    }
    VF xh[KS], xl[KS], yh_[CROSS ? KS : 1], yl_[CROSS ? KS : 1];
    tiles_to_frags<NT>(xt, sa, xh, xl);
    if constexpr (CROSS) tiles_to_frags<NT>(yt, sa, yh_, yl_);
    const VF* yh = CROSS ? yh_ : xh;
    const VF* yl = CROSS ? yl_ : xl;

    MAB_STAMP(2);
    // ---- feed-forward half: dzf = dropout'(dout) ; du = dzf drop_ff' act'(u) ; dz = dzf + du Wf ; dza = dropout'(dz)
    VB dzah[KS], dzal[KS];
    f32x16 dxa[NT];                       // gradient with respect to x: starts as the residual path
    mab_bwd_ff<NT, LN>(p, dzf, zt, zat, xlive, rF, rFT, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, h, lane16, dzah, dzal, dxa);
    MAB_STAMP(3);
    f32x16 dya[CROSS ? NT : 1];
    if constexpr (CROSS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) dya[t] = zero16();
    }
    f32x16* dkv_acc = CROSS ? dya : dxa;

    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        // gradient of the attention output of this tile's two heads, both orientations
        const f32x16 dOn = proj_n<KS>(rOT, nfE, t, dzah, dzal, zero16(), lane16);
        const f32x16 dOp = proj_t<KS>(rOT, nfE, t, dzah, dzal, zero16(), lane16);
        const f32x16 Qn = proj_n<KS>(rIn, nfIn, t, xh, xl, bias_regs(sBin, t, h), lane16);
        const f32x16 Kn = proj_n<KS>(rIn, nfIn, NT + t, yh, yl, bias_regs(sBin, NT + t, h), lane16);
        const f32x16 Vn = proj_n<KS>(rIn, nfIn, 2 * NT + t, yh, yl, bias_regs(sBin, 2 * NT + t, h), lane16);
        const f32x16 Qp = proj_t<KS>(rIn, nfIn, t, xh, xl, bias_lanes(sBin, t, r), lane16);
        const f32x16 Kp = proj_t<KS>(rIn, nfIn, NT + t, yh, yl, bias_lanes(sBin, NT + t, r), lane16);
        VB kph[2], kpl[2], qph[2], qpl[2], doph[2], dopl[2];
        static_for<0, 2>([&](auto sc) {
            MPG_CI(s, sc);
            tile_frag(Kp, s, inv_zs, kph[s], kpl[s]);
            tile_frag(Qp, s, inv_zs, qph[s], qpl[s]);
            tile_frag(dOp, s, 1.f, doph[s], dopl[s]);
        });
        f32x16 dQt, dKt, dVt;
        static_for<0, 2>([&](auto ac) {
            MPG_CI(a, ac);
            VF qh, ql, kh, kl;
            tile_frag(Qn, a, inv_zs * sa * 0.25f, qh, ql);
            tile_frag(Kn, a, inv_zs * sa, kh, kl);
            VB vbh, vbl, dobh, dobl;
            tile_frag(Vn, a, inv_zs, vbh, vbl);
            tile_frag(dOn, a, 1.f, dobh, dobl);
            // ---- keys in registers, queries on lanes
            f32x16 s = mfma3(kh, kl, qh, ql, zero16());
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = s[i] * sc2 + kneg[i]; mx = fmaxf(mx, s[i]); }
            mx = fmaxf(mx, other_half(mx));
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mx); den += s[i]; }
            den += other_half(den);
            const float inv_den = 1.f / den, cq = mx + __builtin_amdgcn_logf(den);   // log2
            const f32x16 dP = mfma3(vbh, vbl, dobh, dobl, zero16());
            float D = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] *= inv_den; D += s[i] * dP[i]; }
            D += other_half(D);
            f32x16 dS;
#pragma unroll
            for (int i = 0; i < 16; ++i) dS[i] = s[i] * (dP[i] - D) * 0.25f;
            VB dsh[2], dsl[2];
            tile_frag(dS, 0, 1.f, dsh[0], dsl[0]);
            tile_frag(dS, 1, 1.f, dsh[1], dsl[1]);
            f32x16 dq = mfma3(kph[0], kpl[0], dsh[0], dsl[0], zero16());
            dq = mfma3(kph[1], kpl[1], dsh[1], dsl[1], dq);
            // ---- queries in registers, keys on lanes
            f32x16 sT = mfma3(qh, ql, kh, kl, zero16());
            const f32x16 dPT = mfma3(dobh, dobl, vbh, vbl, zero16());
            f32x16 pT, dST;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g + e, qi = 8 * g + 4 * h + e;   // the query this register belongs to
                    const float c_q = __shfl(cq, qi), D_q = __shfl(D, qi);
                    const float pr = key_off ? 0.f : __builtin_amdgcn_exp2f(sT[i] * sc2 - c_q);
                    pT[i] = pr;
                    dST[i] = pr * (dPT[i] - D_q) * 0.25f;
                }
            VB pth[2], ptl[2], dsth[2], dstl[2];
            static_for<0, 2>([&](auto sc) {
                MPG_CI(s2, sc);
                tile_frag(pT, s2, 1.f, pth[s2], ptl[s2]);
                tile_frag(dST, s2, 1.f, dsth[s2], dstl[s2]);
            });
            f32x16 dk = mfma3(qph[0], qpl[0], dsth[0], dstl[0], zero16());
            dk = mfma3(qph[1], qpl[1], dsth[1], dstl[1], dk);
            f32x16 dv = mfma3(doph[0], dopl[0], pth[0], ptl[0], zero16());
            dv = mfma3(doph[1], dopl[1], pth[1], ptl[1], dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dQt[8 * a + j] = dq[8 * a + j]; dKt[8 * a + j] = dk[8 * a + j]; dVt[8 * a + j] = dv[8 * a + j]; }
        });
        if (p.dq != nullptr && xvalid) tile_to_rows(p.dq, p.lddq, xrow, t, h, dQt, 1.f);
        if (p.dk != nullptr && yvalid) {
            tile_to_rows(p.dk, p.lddkv, yrow, t, h, dKt, 1.f);
            tile_to_rows(p.dv, p.lddkv, yrow, t, h, dVt, 1.f);
        }
        // input gradients: dx += dq Wq ; (dy or dx) += dk Wk + dv Wv
        VB fh[2], fl[2];
        tile_frag(dQt, 0, 1.f, fh[0], fl[0]); tile_frag(dQt, 1, 1.f, fh[1], fl[1]);
        acc_wt<NT>(rInT, nfInT, 3 * KS, 2 * t, fh, fl, dxa, lane16);
        tile_frag(dKt, 0, 1.f, fh[0], fl[0]); tile_frag(dKt, 1, 1.f, fh[1], fl[1]);
        acc_wt<NT>(rInT, nfInT, 3 * KS, 2 * (NT + t), fh, fl, dkv_acc, lane16);
        tile_frag(dVt, 0, 1.f, fh[0], fl[0]); tile_frag(dVt, 1, 1.f, fh[1], fl[1]);
        acc_wt<NT>(rInT, nfInT, 3 * KS, 2 * (2 * NT + t), fh, fl, dkv_acc, lane16);
    });
    MAB_STAMP(4);
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        if (p.dx != nullptr && xvalid) tile_to_rows(p.dx, p.lddx, xrow, t, h, dxa[t], 1.f);
        if constexpr (CROSS) {
            if (p.dy != nullptr && yvalid) tile_to_rows(p.dy, p.lddy, yrow, t, h, dya[t], 1.f);
        }
    });
    MAB_STAMP(5);
    }  // (the wave's jet)
#ifdef MPG_MABSTAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) g_mab_stamps[32 + w * 8 + i] = mab_st[i];
#endif
}

// ---- LARGE SETS, backward, the form that SHARES the projections (the recomputing form of round 5 -- every wave projecting the other
// tiles' rows itself -- is tools/ubench/mab_bwdN_recompute.inc).  Same ownership -- wave w: query tile w and key tile w --, but every projection is made ONCE, by its
// tile's owner, and handed round through LDS as the fragments the products take.  The room comes from the weight images: Wf / WfT
// are dead behind the feed-forward half and WinT is not loaded until its turn, which leaves 80 KiB beside Win and WoT -- the key
// side of ONE feature tile for all five tiles (12 KiB each) or the query side (16 KiB each).  Per feature tile t:
//   K-phase: owners write kh kl (per head) | vbh vbl (per head) | kph kpl (per k-step)      -> barrier
//   P2:      own queries against every key tile: statistics pass, dS pass -> dQ' of t         -> barrier
//   Q-phase: owners write qh ql | dobh dobl (per head) | qph qpl | doph dopl (per k-step)    -> barrier
//   P3:      own keys against every query tile -> dK', dV' of t                              -> barrier
//   WinT's image is filled over the exchange area                                             -> barrier
//   tail:    dxa += dQ' Wq, (dya or dxa) += dK' Wk + dV' Wv                                   -> barrier
// Every sum runs in tile order inside one wave (deterministic); nothing goes through memory any more (dza is written for Wo's weight
// gradient only).  ~1,050 MFMAs per wave instead of ~1,950.
template <int NT, bool CROSS, bool LN>
__global__ __launch_bounds__(320) void mab_bwdS_kernel(const MpgMab p) {
    typedef f16x8 VF;
    typedef bf16x8 VB;
    constexpr int KS = 2 * NT, NH = 2 * NT, TOKMAX = 160;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5, lane16 = lane * 16;
    const uint64_t sd = *(p.seed != nullptr ? p.seed : reinterpret_cast<const uint64_t*>(p.x));
    const uint32_t seed_lo = p.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p.ascale > 0.f ? p.ascale : 1.f, ws = p.wscale > 0.f ? p.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs, sc2 = 1.44269504088896341f / (sa * sa);   // (scores in the base-2 domain)
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS, nfInT = NT * 3 * KS;
    constexpr int XCH = 80 * 1024;        // the exchange area: [Wf | WfT | room] -- later WinT's image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sOT = sIn + 2 * nfIn * 1024;
    char* const sX = sOT + 2 * nfE * 1024;
    char* const sF = sX;
    char* const sFT = sF + 2 * nfE * 1024;
    float* const sBin = reinterpret_cast<float*>(sX + XCH);   // biases: in_proj [3E] | ff [E]
    float* const sBf = sBin + 96 * NT;
    float* const sCQ = sBin + 128 * NT;   // [head][token]: the maximum of a query's scores (+inf: the query attends to nothing)
    float* const sID = sCQ + NH * TOKMAX; // [head][token]: 1 / sum_keys 2^(score - maximum)
    float* const sDD = sID + NH * TOKMAX; // [head][token]: sum_keys P dP
    static_assert(2 * 2 * nfE * 1024 <= XCH && 2 * nfInT * 1024 <= XCH, "Wf, WfT and later WinT live in the exchange area");
    const long jet = blockIdx.x;
    const int nqt = (p.L + 31) >> 5, nkt = (p.S + 31) >> 5;
    const bool isq = w < nqt, isk = w < nkt;
    const int tok = 32 * w + r;
    const long xrow = jet * p.L + min(tok, p.L - 1), yrow = jet * p.S + min(tok, p.S - 1);
    const bool xvalid = isq && tok < p.L, yvalid = isk && tok < p.S;
    f32x16 dzf[NT], zt[NT], zat[LN ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        dzf[t] = rows_to_tile(p.dout, p.lddout, xrow, t, h);
        zt[t] = rows_to_tile(p.save_z, p.E, xrow, t, h);
        if constexpr (LN) zat[t] = rows_to_tile(p.save_za, p.E, xrow, t, h);
    }
    mab_fill(sIn, p.Win, 2 * nfIn * 1024);
    mab_fill(sOT, p.WoT, 2 * nfE * 1024);
    mab_fill(sF, p.Wf, 2 * nfE * 1024);
    mab_fill(sFT, p.WfT, 2 * nfE * 1024);
    for (int i = threadIdx.x; i < 128 * NT; i += blockDim.x) sBin[i] = (i < 96 * NT ? p.bin[i] : p.bf[i - 96 * NT]) * zs;
    __syncthreads();
    const WImg rIn = sIn, rF = sF, rOT = sOT, rFT = sFT, rInT = sX;
    f32x16 dxa[NT], dya[NT];              // (dya: the key side -- dy of a cross block, the second half of dx of a self-attention block)
#pragma unroll
    for (int t = 0; t < NT; ++t) { dxa[t] = zero16(); dya[t] = zero16(); }
    VB dzah[KS], dzal[KS];
    if (isq) mab_bwd_ff<NT, LN>(p, dzf, zt, zat, xvalid ? 1.f : 0.f, rF, rFT, sBf, xrow, xvalid, seed_lo, seed_hi, sa, inv_zs, h, lane16, dzah, dzal, dxa);
    __syncthreads();                      // every wave is through the feed-forward half: Wf and WfT are dead, the exchange area is free
    const float ign_r = (p.ignore != nullptr ? p.ignore + jet * p.S : p.x)[p.ignore != nullptr ? min(tok, p.S - 1) : 0];
    const bool key_off = !yvalid || (p.ignore != nullptr && ign_r != 0.f);
    auto frag_at = [&](int tile, int per_tile, int i) -> VB* { return reinterpret_cast<VB*>(sX + ((tile * per_tile + i) * 64 + lane) * 16); };

    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        // ---- K-phase: the key side of (own tile, t): [0..3] kh kl of heads 0, 1 (fp16) | [4..7] vbh vbl | [8..11] kph kpl of k-steps 0, 1
        if (isk) {
            VF yh[KS], yl[KS];
            rows_to_frags<KS>(p.y, p.ldy, yrow, sa, h, yh, yl);
            const f32x16 Kn = proj_n<KS>(rIn, nfIn, NT + t, yh, yl, bias_regs(sBin, NT + t, h), lane16);
            const f32x16 Vn = proj_n<KS>(rIn, nfIn, 2 * NT + t, yh, yl, bias_regs(sBin, 2 * NT + t, h), lane16);
            const f32x16 Kp = proj_t<KS>(rIn, nfIn, NT + t, yh, yl, bias_lanes(sBin, NT + t, r), lane16);
            static_for<0, 2>([&](auto ac) {
                MPG_CI(a, ac);
                VF kh, kl;
                VB vh, vl, ph, pl;
                tile_frag(Kn, a, inv_zs * sa, kh, kl);
                tile_frag(Vn, a, inv_zs, vh, vl);
                tile_frag(Kp, a, inv_zs, ph, pl);
                *reinterpret_cast<VF*>(frag_at(w, 12, 2 * a)) = kh;
                *reinterpret_cast<VF*>(frag_at(w, 12, 2 * a + 1)) = kl;
                *frag_at(w, 12, 4 + 2 * a) = vh;
                *frag_at(w, 12, 5 + 2 * a) = vl;
                *frag_at(w, 12, 8 + 2 * a) = ph;
                *frag_at(w, 12, 9 + 2 * a) = pl;
            });
        }
        __syncthreads();
        // ---- P2: the wave's own queries against every key tile
        VF qh[2], ql[2];
        VB dobh[2], dobl[2];
        f32x16 dQt = zero16();
        if (isq) {
            {
                VF xh[KS], xl[KS];
                rows_to_frags<KS>(p.x, p.ldx, xrow, sa, h, xh, xl);
                const f32x16 Qn = proj_n<KS>(rIn, nfIn, t, xh, xl, bias_regs(sBin, t, h), lane16);
                const f32x16 dOn = proj_n<KS>(rOT, nfE, t, dzah, dzal, zero16(), lane16);
                static_for<0, 2>([&](auto ac) {
                    MPG_CI(a, ac);
                    tile_frag(Qn, a, inv_zs * sa * 0.25f, qh[a], ql[a]);
                    tile_frag(dOn, a, 1.f, dobh[a], dobl[a]);
                });
            }
            // pass 1: the running maximum, sum and sum of P dP of every (query, head) -- D from the very dP values pass 2 works with
            float cq[2], iden[2], Dq[2];
            {
                float mrun[2] = {-INFINITY, -INFINITY}, den[2] = {0.f, 0.f};
                Dq[0] = Dq[1] = 0.f;
                for (int kt = 0; kt < nkt; ++kt) {
                    const f32x16 kneg = key_mask_tile(p.ignore, p.x, jet, p.S, h, kt);
                    static_for<0, 2>([&](auto ac) {
                        MPG_CI(a, ac);
                        const VF kh = *reinterpret_cast<const VF*>(frag_at(kt, 12, 2 * a)), kl = *reinterpret_cast<const VF*>(frag_at(kt, 12, 2 * a + 1));
                        const VB vbh = *frag_at(kt, 12, 4 + 2 * a), vbl = *frag_at(kt, 12, 5 + 2 * a);
                        f32x16 sx = mfma3(kh, kl, qh[a], ql[a], zero16());
                        const f32x16 dP = mfma3(vbh, vbl, dobh[a], dobl[a], zero16());
                        float mx = -INFINITY;
#pragma unroll
                        for (int i = 0; i < 16; ++i) { sx[i] = sx[i] * sc2 + kneg[i]; mx = fmaxf(mx, sx[i]); }
                        mx = fmaxf(mx, other_half(mx));
                        const float mnew = fmaxf(mrun[a], mx);
                        const float msafe = mnew == -INFINITY ? 0.f : mnew;
                        float part = 0.f, dpart = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float e = __builtin_amdgcn_exp2f(sx[i] - msafe);
                            part += e;
                            dpart = fmaf(e, dP[i], dpart);
                        }
                        part += other_half(part);
                        dpart += other_half(dpart);
                        const float resc = __builtin_amdgcn_exp2f(mrun[a] - msafe);
                        den[a] = den[a] * resc + part;
                        Dq[a] = Dq[a] * resc + dpart;
                        mrun[a] = mnew;
                    });
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const bool live = den[a] > 0.f && xvalid;
                    cq[a] = live ? mrun[a] : INFINITY;
                    iden[a] = live ? 1.f / den[a] : 0.f;
                    Dq[a] *= iden[a];
                    const int hd = 2 * t + a;
                    if (h == 0) { sCQ[hd * TOKMAX + tok] = cq[a]; sID[hd * TOKMAX + tok] = iden[a]; sDD[hd * TOKMAX + tok] = Dq[a]; }
                }
            }
            // pass 2: dS tile by tile, dQ' accumulated over the key tiles
            f32x16 dqa[2] = {zero16(), zero16()};
            for (int kt = 0; kt < nkt; ++kt) {
                const f32x16 kneg = key_mask_tile(p.ignore, p.x, jet, p.S, h, kt);
                const VB kph[2] = {*frag_at(kt, 12, 8), *frag_at(kt, 12, 10)}, kpl[2] = {*frag_at(kt, 12, 9), *frag_at(kt, 12, 11)};
                static_for<0, 2>([&](auto ac) {
                    MPG_CI(a, ac);
                    const VF kh = *reinterpret_cast<const VF*>(frag_at(kt, 12, 2 * a)), kl = *reinterpret_cast<const VF*>(frag_at(kt, 12, 2 * a + 1));
                    const VB vbh = *frag_at(kt, 12, 4 + 2 * a), vbl = *frag_at(kt, 12, 5 + 2 * a);
                    const f32x16 sx = mfma3(kh, kl, qh[a], ql[a], zero16());
                    const f32x16 dP = mfma3(vbh, vbl, dobh[a], dobl[a], zero16());
                    f32x16 dS;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float pr = __builtin_amdgcn_exp2f(sx[i] * sc2 + kneg[i] - cq[a]) * iden[a];
                        dS[i] = pr * (dP[i] - Dq[a]) * 0.25f;
                    }
                    VB dsh[2], dsl[2];
                    tile_frag(dS, 0, 1.f, dsh[0], dsl[0]);
                    tile_frag(dS, 1, 1.f, dsh[1], dsl[1]);
                    dqa[a] = mfma3(kph[0], kpl[0], dsh[0], dsl[0], dqa[a]);
                    dqa[a] = mfma3(kph[1], kpl[1], dsh[1], dsl[1], dqa[a]);
                });
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { dQt[j] = dqa[0][j]; dQt[8 + j] = dqa[1][8 + j]; }
            if (p.dq != nullptr && xvalid) tile_to_rows(p.dq, p.lddq, xrow, t, h, dQt, 1.f);
        }
        __syncthreads();                  // the key side of t is read; every query's statistics of t's heads are in LDS
        // ---- Q-phase: the query side of (own tile, t): [0..3] qh ql of heads 0, 1 (fp16) | [4..7] dobh dobl | [8..11] qph qpl of k-steps 0, 1 |
        //      [12..15] doph dopl
        if (isq) {
            VF xh[KS], xl[KS];
            rows_to_frags<KS>(p.x, p.ldx, xrow, sa, h, xh, xl);
            const f32x16 Qp = proj_t<KS>(rIn, nfIn, t, xh, xl, bias_lanes(sBin, t, r), lane16);
            const f32x16 dOp = proj_t<KS>(rOT, nfE, t, dzah, dzal, zero16(), lane16);
            static_for<0, 2>([&](auto ac) {
                MPG_CI(a, ac);
                VB ph, pl, oh, ol;
                tile_frag(Qp, a, inv_zs, ph, pl);
                tile_frag(dOp, a, 1.f, oh, ol);
                *reinterpret_cast<VF*>(frag_at(w, 16, 2 * a)) = qh[a];
                *reinterpret_cast<VF*>(frag_at(w, 16, 2 * a + 1)) = ql[a];
                *frag_at(w, 16, 4 + 2 * a) = dobh[a];
                *frag_at(w, 16, 5 + 2 * a) = dobl[a];
                *frag_at(w, 16, 8 + 2 * a) = ph;
                *frag_at(w, 16, 9 + 2 * a) = pl;
                *frag_at(w, 16, 12 + 2 * a) = oh;
                *frag_at(w, 16, 13 + 2 * a) = ol;
            });
        }
        __syncthreads();
        // ---- P3: the wave's own keys against every query tile
        f32x16 dKt = zero16(), dVt = zero16();
        if (isk) {
            VF kh[2], kl[2];
            VB vbh[2], vbl[2];
            {
                VF yh[KS], yl[KS];
                rows_to_frags<KS>(p.y, p.ldy, yrow, sa, h, yh, yl);
                const f32x16 Kn = proj_n<KS>(rIn, nfIn, NT + t, yh, yl, bias_regs(sBin, NT + t, h), lane16);
                const f32x16 Vn = proj_n<KS>(rIn, nfIn, 2 * NT + t, yh, yl, bias_regs(sBin, 2 * NT + t, h), lane16);
                static_for<0, 2>([&](auto ac) {
                    MPG_CI(a, ac);
                    tile_frag(Kn, a, inv_zs * sa, kh[a], kl[a]);
                    tile_frag(Vn, a, inv_zs, vbh[a], vbl[a]);
                });
            }
            f32x16 dka[2] = {zero16(), zero16()}, dva[2] = {zero16(), zero16()};
            for (int qt = 0; qt < nqt; ++qt) {
                const VB qph[2] = {*frag_at(qt, 16, 8), *frag_at(qt, 16, 10)}, qpl[2] = {*frag_at(qt, 16, 9), *frag_at(qt, 16, 11)};
                const VB doph[2] = {*frag_at(qt, 16, 12), *frag_at(qt, 16, 14)}, dopl[2] = {*frag_at(qt, 16, 13), *frag_at(qt, 16, 15)};
                static_for<0, 2>([&](auto ac) {
                    MPG_CI(a, ac);
                    constexpr int hd = 2 * t + a;
                    const VF qh2 = *reinterpret_cast<const VF*>(frag_at(qt, 16, 2 * a)), ql2 = *reinterpret_cast<const VF*>(frag_at(qt, 16, 2 * a + 1));
                    const VB dbh = *frag_at(qt, 16, 4 + 2 * a), dbl = *frag_at(qt, 16, 5 + 2 * a);
                    const f32x16 sT = mfma3(qh2, ql2, kh[a], kl[a], zero16());        // queries in registers, keys on lanes
                    const f32x16 dPT = mfma3(dbh, dbl, vbh[a], vbl[a], zero16());
                    f32x16 pT, dST;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 4 * g + e, qi = 32 * qt + 8 * g + 4 * h + e;   // the query this register belongs to
                            const float c_q = sCQ[hd * TOKMAX + qi], id_q = sID[hd * TOKMAX + qi], D_q = sDD[hd * TOKMAX + qi];
                            const float pr = key_off ? 0.f : __builtin_amdgcn_exp2f(sT[i] * sc2 - c_q) * id_q;
                            pT[i] = pr;
                            dST[i] = pr * (dPT[i] - D_q) * 0.25f;
                        }
                    VB pth[2], ptl[2], dsth[2], dstl[2];
                    static_for<0, 2>([&](auto sc) {
                        MPG_CI(s2, sc);
                        tile_frag(pT, s2, 1.f, pth[s2], ptl[s2]);
                        tile_frag(dST, s2, 1.f, dsth[s2], dstl[s2]);
                    });
                    dka[a] = mfma3(qph[0], qpl[0], dsth[0], dstl[0], dka[a]);
                    dka[a] = mfma3(qph[1], qpl[1], dsth[1], dstl[1], dka[a]);
                    dva[a] = mfma3(doph[0], dopl[0], pth[0], ptl[0], dva[a]);
                    dva[a] = mfma3(doph[1], dopl[1], pth[1], ptl[1], dva[a]);
                });
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { dKt[j] = dka[0][j]; dKt[8 + j] = dka[1][8 + j]; dVt[j] = dva[0][j]; dVt[8 + j] = dva[1][8 + j]; }
            if (p.dk != nullptr && yvalid) {
                tile_to_rows(p.dk, p.lddkv, yrow, t, h, dKt, 1.f);
                tile_to_rows(p.dv, p.lddkv, yrow, t, h, dVt, 1.f);
            }
        }
        __syncthreads();                  // the query side of t is read: the exchange area takes WinT's image
        mab_fill(sX, p.WinT, 2 * nfInT * 1024);
        __syncthreads();
        // ---- input gradients of this feature tile's dq, dk, dv
        {
            VB fh[2], fl[2];
            if (isq) {
                tile_frag(dQt, 0, 1.f, fh[0], fl[0]); tile_frag(dQt, 1, 1.f, fh[1], fl[1]);
                acc_wt<NT>(rInT, nfInT, 3 * KS, 2 * t, fh, fl, dxa, lane16);
            }
            if (isk) {
                tile_frag(dKt, 0, 1.f, fh[0], fl[0]); tile_frag(dKt, 1, 1.f, fh[1], fl[1]);
                acc_wt<NT>(rInT, nfInT, 3 * KS, 2 * (NT + t), fh, fl, dya, lane16);
                tile_frag(dVt, 0, 1.f, fh[0], fl[0]); tile_frag(dVt, 1, 1.f, fh[1], fl[1]);
                acc_wt<NT>(rInT, nfInT, 3 * KS, 2 * (2 * NT + t), fh, fl, dya, lane16);
            }
        }
        if constexpr (t + 1 < NT) __syncthreads();   // WinT's image is read: the next feature tile's key side may land there
    });
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        if constexpr (CROSS) {
            if (p.dx != nullptr && xvalid) tile_to_rows(p.dx, p.lddx, xrow, t, h, dxa[t], 1.f);
            if (p.dy != nullptr && yvalid) tile_to_rows(p.dy, p.lddy, yrow, t, h, dya[t], 1.f);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) dxa[t][i] += dya[t][i];
            if (p.dx != nullptr && xvalid) tile_to_rows(p.dx, p.lddx, xrow, t, h, dxa[t], 1.f);
        }
    });
}

// ---- The backward with TWO WAVES PER JET (E = 64), the split of mab_fwd_half: wave T owns tile T of du, dz, dza, the two heads
// of tile T in the attention (q, k, v, P recomputed for those heads only) and tile T of dx (dy).  Three meetings in LDS: du and
// dza are B fragments of products over ALL features, and the input gradient contracts over all of dq | dk | dv.  Each wave
// adds the terms of its dx tile in the order the one-wave kernel does, so the two give the same bits.  The weight images fill
// the LDS (144 of 160 KiB), so the exchange buffers are the images nobody needs any more, each behind one more barrier:
//     du  -> Wf   (free once every wave has recomputed u)         dza -> WfT (free once every wave has dz)
//     dq | dk | dv -> Win (free once every wave is through the attention)
template <int T, typename V>
MPG_DEV void acc_wt1(WImg rT, int nfragT, int KST, int ks0, const V* fh, const V* fl, f32x16& acc, int lane16) {
    static_for<0, 2>([&](auto sc) {
        MPG_CI(s, sc);
        const V wh = mab_wfrag<V>(rT, T * KST + ks0 + s, lane16), wl = mab_wfrag<V>(rT, nfragT + T * KST + ks0 + s, lane16);
        acc = mfma3(wh, wl, fh[s], fl[s], acc);
    });
}

template <int T, bool CROSS>
MPG_DEV void mab_bwd2_body(const MpgMab& p) {
    typedef f16x8 VF;
    typedef bf16x8 VB;
    constexpr int NT = 2, KS = 4, O = 1 - T;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5, lane16 = lane * 16, pair = w >> 1;
#ifdef MPG_MABSTAMP
    unsigned long long mab_st[8] = {};
#endif
    MAB_STAMP(0);
    // (no branch on the pointer: behind one, the wave waits for the seed before it issues its first load)
    const uint64_t sd = *(p.seed != nullptr ? p.seed : reinterpret_cast<const uint64_t*>(p.x));
    const uint32_t seed_lo = p.seed != nullptr ? (uint32_t)sd : 0u, seed_hi = p.seed != nullptr ? (uint32_t)(sd >> 32) : 0u;
    const float sa = p.ascale > 0.f ? p.ascale : 1.f, ws = p.wscale > 0.f ? p.wscale : 1.f;
    const float zs = sa * ws, inv_zs = 1.f / zs, sc2 = 1.44269504088896341f / (sa * sa);
    constexpr int nfIn = 3 * NT * KS, nfE = NT * KS, nfInT = NT * 3 * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sIn = smem;
    char* const sF = sIn + 2 * nfIn * 1024;
    char* const sInT = sF + 2 * nfE * 1024;
    char* const sOT = sInT + 2 * nfInT * 1024;
    char* const sFT = sOT + 2 * nfE * 1024;
    const int npair = blockDim.x >> 7;
    const long jet_raw = (long)blockIdx.x * npair + pair;
    const bool live = jet_raw < p.B;                   // (a pair without a jet goes through the motions: fills and barriers)
    const long jet = live ? jet_raw : (long)p.B - 1;
    const long xrow = jet * p.L + min(r, p.L - 1), yrow = jet * p.S + min(r, p.S - 1);
    // order of issue: biases and key mask (plain loads, used last), every row of the jet (loads the compiler does not count:
    // rows_request), the 144 KiB fill (36 LDS-DMA instructions per wave); the rows are converted UNDER the fill
    float* const sBin = reinterpret_cast<float*>(sFT + 2 * nfE * 1024);
    float* const sBf = sBin + 96 * NT;
    static_assert(NT == 2, "rows_wait counts the fill of an E = 64 block on four waves");
    const int bi0 = threadIdx.x;                       // 128 NT = 256 biases: one per thread
    const float bv0 = bi0 < 96 * NT ? p.bin[bi0] : p.bf[bi0 - 96 * NT];
    const KeyIgn kig = key_mask_load(p.ignore, p.x, jet, p.S, h);
    const float ign_r = (kig.on ? p.ignore + jet * p.S : p.x)[min(r, p.S - 1)];   // this lane as a KEY (transposed tiles)
    f32x4 dq_[4], zq[4 * NT], xq[4 * NT], yq[CROSS ? 4 * NT : 1];
    {
        const float* const pd = p.dout + xrow * p.lddout + 4 * h + 32 * T;
        static_for<0, 4>([&](auto ic) { MPG_CI(i, ic); dq_[i] = ld4_hidden<8 * i * 4>(pd); });
    }
    rows_request<NT>(p.save_z, p.E, xrow, h, zq);
    if constexpr (CROSS) rows_request<NT>(p.y, p.ldy, yrow, h, yq);
    rows_request<NT>(p.x, p.ldx, xrow, h, xq);
    mab_fill(sIn, p.Win, 2 * nfIn * 1024);
    mab_fill(sF, p.Wf, 2 * nfE * 1024);
    mab_fill(sInT, p.WinT, 2 * nfInT * 1024);
    mab_fill(sOT, p.WoT, 2 * nfE * 1024);
    mab_fill(sFT, p.WfT, 2 * nfE * 1024);
    rows_wait<36>(xq);                                 // (the youngest rows: everything requested before them is here too)
    rows_pin(zq);
    rows_pin4(dq_);
    if constexpr (CROSS) rows_pin(yq);
    f32x16 dzf, zt[NT], xt[NT], yt[CROSS ? NT : 1];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) dzf[4 * g + e] = dq_[g][e];
#pragma unroll
    for (int t = 0; t < NT; ++t) { zt[t] = pieces_tile(zq, t); xt[t] = pieces_tile(xq, t); }
    if constexpr (CROSS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) yt[t] = pieces_tile(yq, t);
    }
    VF xh[KS], xl[KS], yh_[CROSS ? KS : 1], yl_[CROSS ? KS : 1], zh[KS], zl[KS];
    tiles_to_frags<NT>(zt, sa, zh, zl);
    tiles_to_frags<NT>(xt, sa, xh, xl);
    if constexpr (CROSS) tiles_to_frags<NT>(yt, sa, yh_, yl_);
    sBin[bi0] = bv0 * zs;
    const f32x16 kneg = key_mask_from(kig, p.S, h);
    const bool key_off = !(r < p.S) || (kig.on && ign_r != 0.f);
    __syncthreads();
    const WImg rIn = sIn, rF = sF, rInT = sInT, rOT = sOT, rFT = sFT;
    MAB_STAMP(1);
    // exchange slots of 1 KiB: [k-step or fragment index][hi | lo][lane]
    auto put = [&](char* base, int slot, const VB& hi, const VB& lo) {
        reinterpret_cast<VB*>(base)[(slot * 2 + 0) * 64 + lane] = hi;
        reinterpret_cast<VB*>(base)[(slot * 2 + 1) * 64 + lane] = lo;
    };
    auto get = [&](const char* base, int slot, VB& hi, VB& lo) {
        hi = reinterpret_cast<const VB*>(base)[(slot * 2 + 0) * 64 + lane];
        lo = reinterpret_cast<const VB*>(base)[(slot * 2 + 1) * 64 + lane];
    };
    const bool xvalid = live && r < p.L, yvalid = live && r < p.S;
    const float xlive = r < p.L ? 1.f : 0.f;
    const VF* yh = CROSS ? yh_ : xh;
    const VF* yl = CROSS ? yl_ : xl;

    // ---- feed-forward half, tile T
    VB dzah[KS], dzal[KS];
    f32x16 dxa;
    {
        VB duh[KS], dul[KS];
#pragma unroll
        for (int i = 0; i < 16; ++i) dzf[i] *= xlive;
        drop_tile(dzf, seed_lo, seed_hi, p.tag + 2, (uint32_t)xrow, T, h, p.thr_mab, p.sc_mab);
        const f32x16 u = proj_n<KS>(rF, nfE, T, zh, zl, bias_regs(sBf, T, h), lane16);
        f32x16 du;
#pragma unroll
        for (int i = 0; i < 16; ++i) du[i] = dzf[i] * ((p.ff_act && !(u[i] > 0.f)) ? p.alpha : 1.f);
        drop_tile(du, seed_lo, seed_hi, p.tag + 1, (uint32_t)xrow, T, h, p.thr_ff, p.sc_ff);
        if (p.du != nullptr && xvalid) tile_to_rows(p.du, p.E, xrow, T, h, du, 1.f);
        tile_frag(du, 0, 1.f, duh[2 * T], dul[2 * T]);
        tile_frag(du, 1, 1.f, duh[2 * T + 1], dul[2 * T + 1]);
        MAB_STAMP(2);
        __syncthreads();                               // every wave has its u: Wf's image is dead
        char* const x1 = sF + pair * (KS * 2048);
        put(x1, 2 * T, duh[2 * T], dul[2 * T]);
        put(x1, 2 * T + 1, duh[2 * T + 1], dul[2 * T + 1]);
        __syncthreads();
        get(x1, 2 * O, duh[2 * O], dul[2 * O]);
        get(x1, 2 * O + 1, duh[2 * O + 1], dul[2 * O + 1]);
        MAB_STAMP(3);
        f32x16 dz = proj_n<KS>(rFT, nfE, T, duh, dul, dzf, lane16);
        drop_tile(dz, seed_lo, seed_hi, p.tag + 0, (uint32_t)xrow, T, h, p.thr_mab, p.sc_mab);
        if (p.dza != nullptr && xvalid) tile_to_rows(p.dza, p.E, xrow, T, h, dz, 1.f);
        tile_frag(dz, 0, 1.f, dzah[2 * T], dzal[2 * T]);
        tile_frag(dz, 1, 1.f, dzah[2 * T + 1], dzal[2 * T + 1]);
        dxa = dz;
        __syncthreads();                               // every wave has its dz: WfT's image is dead
        char* const x2 = sFT + pair * (KS * 2048);
        put(x2, 2 * T, dzah[2 * T], dzal[2 * T]);
        put(x2, 2 * T + 1, dzah[2 * T + 1], dzal[2 * T + 1]);
        __syncthreads();
        get(x2, 2 * O, dzah[2 * O], dzal[2 * O]);
        get(x2, 2 * O + 1, dzah[2 * O + 1], dzal[2 * O + 1]);
    }
    f32x16 dya = zero16();
    MAB_STAMP(4);

    // ---- the attention of heads 2T, 2T + 1 (mab_bwd_kernel's tile loop body with t = T)
    VB gq_h[NT][2], gq_l[NT][2], gk_h[NT][2], gk_l[NT][2], gv_h[NT][2], gv_l[NT][2];
    {
        constexpr int t = T;
        const f32x16 dOn = proj_n<KS>(rOT, nfE, t, dzah, dzal, zero16(), lane16);
        const f32x16 dOp = proj_t<KS>(rOT, nfE, t, dzah, dzal, zero16(), lane16);
        const f32x16 Qn = proj_n<KS>(rIn, nfIn, t, xh, xl, bias_regs(sBin, t, h), lane16);
        const f32x16 Kn = proj_n<KS>(rIn, nfIn, NT + t, yh, yl, bias_regs(sBin, NT + t, h), lane16);
        const f32x16 Vn = proj_n<KS>(rIn, nfIn, 2 * NT + t, yh, yl, bias_regs(sBin, 2 * NT + t, h), lane16);
        const f32x16 Qp = proj_t<KS>(rIn, nfIn, t, xh, xl, bias_lanes(sBin, t, r), lane16);
        const f32x16 Kp = proj_t<KS>(rIn, nfIn, NT + t, yh, yl, bias_lanes(sBin, NT + t, r), lane16);
        VB kph[2], kpl[2], qph[2], qpl[2], doph[2], dopl[2];
        static_for<0, 2>([&](auto sc) {
            MPG_CI(s, sc);
            tile_frag(Kp, s, inv_zs, kph[s], kpl[s]);
            tile_frag(Qp, s, inv_zs, qph[s], qpl[s]);
            tile_frag(dOp, s, 1.f, doph[s], dopl[s]);
        });
        f32x16 dQt, dKt, dVt;
        static_for<0, 2>([&](auto ac) {
            MPG_CI(a, ac);
            VF qh, ql, kh, kl;
            tile_frag(Qn, a, inv_zs * sa * 0.25f, qh, ql);
            tile_frag(Kn, a, inv_zs * sa, kh, kl);
            VB vbh, vbl, dobh, dobl;
            tile_frag(Vn, a, inv_zs, vbh, vbl);
            tile_frag(dOn, a, 1.f, dobh, dobl);
            f32x16 s = mfma3(kh, kl, qh, ql, zero16());
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = s[i] * sc2 + kneg[i]; mx = fmaxf(mx, s[i]); }
            mx = fmaxf(mx, other_half(mx));
            float den = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mx); den += s[i]; }
            den += other_half(den);
            const float inv_den = 1.f / den, cq = mx + __builtin_amdgcn_logf(den);
            const f32x16 dP = mfma3(vbh, vbl, dobh, dobl, zero16());
            float D = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] *= inv_den; D += s[i] * dP[i]; }
            D += other_half(D);
            f32x16 dS;
#pragma unroll
            for (int i = 0; i < 16; ++i) dS[i] = s[i] * (dP[i] - D) * 0.25f;
            VB dsh[2], dsl[2];
            tile_frag(dS, 0, 1.f, dsh[0], dsl[0]);
            tile_frag(dS, 1, 1.f, dsh[1], dsl[1]);
            f32x16 dq = mfma3(kph[0], kpl[0], dsh[0], dsl[0], zero16());
            dq = mfma3(kph[1], kpl[1], dsh[1], dsl[1], dq);
            f32x16 sT = mfma3(qh, ql, kh, kl, zero16());
            const f32x16 dPT = mfma3(dobh, dobl, vbh, vbl, zero16());
            f32x16 pT, dST;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g + e, qi = 8 * g + 4 * h + e;
                    const float c_q = __shfl(cq, qi), D_q = __shfl(D, qi);
                    const float pr = key_off ? 0.f : __builtin_amdgcn_exp2f(sT[i] * sc2 - c_q);
                    pT[i] = pr;
                    dST[i] = pr * (dPT[i] - D_q) * 0.25f;
                }
            VB pth[2], ptl[2], dsth[2], dstl[2];
            static_for<0, 2>([&](auto sc) {
                MPG_CI(s2, sc);
                tile_frag(pT, s2, 1.f, pth[s2], ptl[s2]);
                tile_frag(dST, s2, 1.f, dsth[s2], dstl[s2]);
            });
            f32x16 dk = mfma3(qph[0], qpl[0], dsth[0], dstl[0], zero16());
            dk = mfma3(qph[1], qpl[1], dsth[1], dstl[1], dk);
            f32x16 dv = mfma3(doph[0], dopl[0], pth[0], ptl[0], zero16());
            dv = mfma3(doph[1], dopl[1], pth[1], ptl[1], dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dQt[8 * a + j] = dq[8 * a + j]; dKt[8 * a + j] = dk[8 * a + j]; dVt[8 * a + j] = dv[8 * a + j]; }
        });
        if (p.dq != nullptr && xvalid) tile_to_rows(p.dq, p.lddq, xrow, t, h, dQt, 1.f);
        if (p.dk != nullptr && yvalid) {
            tile_to_rows(p.dk, p.lddkv, yrow, t, h, dKt, 1.f);
            tile_to_rows(p.dv, p.lddkv, yrow, t, h, dVt, 1.f);
        }
        static_for<0, 2>([&](auto sc) {
            MPG_CI(s, sc);
            tile_frag(dQt, s, 1.f, gq_h[T][s], gq_l[T][s]);
            tile_frag(dKt, s, 1.f, gk_h[T][s], gk_l[T][s]);
            tile_frag(dVt, s, 1.f, gv_h[T][s], gv_l[T][s]);
        });
    }
    MAB_STAMP(5);
    __syncthreads();                                   // every wave is through the attention: Win's image is dead
    char* const x3 = sIn + pair * (12 * 2048);         // slots: [tile][q | k | v][s]
    static_for<0, 2>([&](auto sc) {
        MPG_CI(s, sc);
        put(x3, (T * 3 + 0) * 2 + s, gq_h[T][s], gq_l[T][s]);
        put(x3, (T * 3 + 1) * 2 + s, gk_h[T][s], gk_l[T][s]);
        put(x3, (T * 3 + 2) * 2 + s, gv_h[T][s], gv_l[T][s]);
    });
    __syncthreads();
    static_for<0, 2>([&](auto sc) {
        MPG_CI(s, sc);
        get(x3, (O * 3 + 0) * 2 + s, gq_h[O][s], gq_l[O][s]);
        get(x3, (O * 3 + 1) * 2 + s, gk_h[O][s], gk_l[O][s]);
        get(x3, (O * 3 + 2) * 2 + s, gv_h[O][s], gv_l[O][s]);
    });
    MAB_STAMP(6);
    // input gradients of tile T: dx += dq Wq ; (dy or dx) += dk Wk + dv Wv, the heads' tiles in the one-wave kernel's order
    f32x16& dkv = CROSS ? dya : dxa;
    static_for<0, NT>([&](auto tc) {
        MPG_CI(t, tc);
        acc_wt1<T>(rInT, nfInT, 3 * KS, 2 * t, gq_h[t], gq_l[t], dxa, lane16);
        acc_wt1<T>(rInT, nfInT, 3 * KS, 2 * (NT + t), gk_h[t], gk_l[t], dkv, lane16);
        acc_wt1<T>(rInT, nfInT, 3 * KS, 2 * (2 * NT + t), gv_h[t], gv_l[t], dkv, lane16);
    });
    if (p.dx != nullptr && xvalid) tile_to_rows(p.dx, p.lddx, xrow, T, h, dxa, 1.f);
    if constexpr (CROSS) {
        if (p.dy != nullptr && yvalid) tile_to_rows(p.dy, p.lddy, yrow, T, h, dya, 1.f);
    }
    MAB_STAMP(7);
#ifdef MPG_MABSTAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) g_mab_stamps[32 + w * 8 + i] = mab_st[i];
#endif
}

template <bool CROSS>
__global__ __launch_bounds__(256) void mab_bwd2_kernel(const MpgMab p) {
    if (((threadIdx.x >> 6) & 1) == 0) mab_bwd2_body<0, CROSS>(p);
    else mab_bwd2_body<1, CROSS>(p);
}

int mab_check(const MpgMab* p) {
    if (p->B < 1 || p->L < 1 || p->S < 1 || p->L > 160 || p->S > 160) return -1;   // (sets of up to 5 tiles of 32 tokens)
    if ((p->E != 32 && p->E != 64) || p->H * 16 != p->E) return -2;
    if (!(p->alpha >= 0.f && p->alpha <= 1.f)) return -4;
    if (p->ldx % 4 || p->ldy % 4 || p->ldo % 4) return -3;
    return 0;
}


#ifdef MPG_MABSTAMP
}  // namespace
extern "C" int mpg_debug_mab_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mab_stamps), sizeof(unsigned long long) * 64);
}
namespace {
#endif

// waves (= jets in flight) per workgroup: as many workgroups as CUs first, then up to four waves each
int mab_waves(int B) {
    static const int forced = getenv("MPG_MAB_WAVES") ? atoi(getenv("MPG_MAB_WAVES")) : 0;   // (experiments)
    if (forced >= 1 && forced <= 4) return forced;
    return B <= 256 ? 1 : (B <= 512 ? 2 : 4);
}

// E = 64: two waves per jet (mab_fwd_half, mab_bwd2_body) while the launch has fewer jets than the chip has SIMDs to give
// each two (256 CUs x 4 / 2 = 512): a shorter chain per wave on SIMDs that would idle.  Past that every SIMD has a jet of its
// own.  The FORWARD (under 256 registers) then doubles up to 1,024 jets: four jets = eight waves to a workgroup, two waves
// to a SIMD, one issuing its hi/lo splits and softmax while the other's MFMAs run; the backward (340-400 registers, one wave
// to a SIMD) keeps one wave per jet.  MPG_MAB_SPLIT=0 keeps one wave per jet everywhere, =2 splits at any size, =3 splits
// up to 512 jets only (the A/B of the doubled forward); read at every launch, so a test can hold the forms against each
// other in one process (they give the same bits).
int mab_split_mode() {
    const char* e = getenv("MPG_MAB_SPLIT");
    return e == nullptr ? 1 : atoi(e);
}
bool mab_split(int B) {
    const int mode = mab_split_mode();
    return mode == 2 || ((mode == 1 || mode == 3) && B <= 512);
}
// waves per workgroup of the split forward: 0 = not split
int mab_split_fwd_waves(int B) {
    const int mode = mab_split_mode();
    if (mode == 0) return 0;
    if (B <= 512) return 4;
    if (mode == 2 || (mode == 1 && B <= 1024)) return 8;
    return 0;
}

template <typename K>
int mab_launch(K kernel, const MpgMab* p, int lds_bytes, hipStream_t st, bool one_jet_per_wave = false) {
    const int nw = mab_waves(p->B);
    const int grid = ((p->B + nw - 1) / nw < 1024 || one_jet_per_wave) ? (p->B + nw - 1) / nw : 1024;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * nw), lds_bytes, st, *p);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mpg_mab_bwd(const MpgMab* p, void* stream) {
    if (const int rc = mab_check(p)) return rc;
    if (p->dout == nullptr || p->save_z == nullptr || (p->dk == nullptr) != (p->dv == nullptr)) return -5;
    if (p->lddout % 4 || (p->dq != nullptr && p->lddq % 4) || (p->dk != nullptr && p->lddkv % 4) ||
        (p->dx != nullptr && p->lddx % 4) || (p->dy != nullptr && p->lddy % 4)) return -3;
    hipStream_t st = (hipStream_t)stream;
    const bool cross = p->y != p->x;
    const int NT = p->E / 32;
    const int lds = 2 * 1024 * (2 * 3 * NT * 2 * NT + 3 * NT * 2 * NT) + 4 * 128 * NT;   // Win, WinT (3E x E) + Wf, WoT, WfT (E x E) + biases
    if (p->L > 32 || p->S > 32 || getenv("MPG_MAB_BIG") != nullptr) {   // large sets: a workgroup per jet, a wave per tile of 32 tokens (mab_bwdS_kernel)
        const bool ln = p->ln1_w != nullptr;
        if (ln && (p->ln2_w == nullptr || p->save_za == nullptr || !(p->ln_eps > 0.f) || (p->dn1 == nullptr) != (p->gn1 == nullptr) ||
                   (p->dn1 == nullptr) != (p->dn2 == nullptr) || (p->dn1 == nullptr) != (p->gn2 == nullptr))) return -6;
        const int nw = (std::max(p->L, p->S) + 31) / 32;
        // Win + WoT + the 80 KiB exchange area + biases + statistics
        const int ldsS = 2 * 1024 * (3 * NT * 2 * NT + NT * 2 * NT) + 80 * 1024 + 4 * 128 * NT + 4 * 3 * (2 * NT) * 160;
        const dim3 grid(p->B), block(64 * nw);
#define MPG_BWDN(NTv, CR, LNv) do { MPG_ENSURE_LDS((mab_bwdS_kernel<NTv, CR, LNv>), ldsS); \
        hipLaunchKernelGGL((mab_bwdS_kernel<NTv, CR, LNv>), grid, block, ldsS, st, *p); } while (0)
        if (NT == 2) { if (cross) { if (ln) MPG_BWDN(2, true, true); else MPG_BWDN(2, true, false); }
                       else { if (ln) MPG_BWDN(2, false, true); else MPG_BWDN(2, false, false); } }
        else { if (cross) { if (ln) MPG_BWDN(1, true, true); else MPG_BWDN(1, true, false); }
               else { if (ln) MPG_BWDN(1, false, true); else MPG_BWDN(1, false, false); } }
#undef MPG_BWDN
        return (int)hipGetLastError();
    }
    if (p->ln1_w != nullptr) {   // layer_norm=True: one wave per jet
        if (p->ln2_w == nullptr || p->save_za == nullptr || !(p->ln_eps > 0.f) || (p->dn1 == nullptr) != (p->gn1 == nullptr) ||
            (p->dn1 == nullptr) != (p->dn2 == nullptr) || (p->dn1 == nullptr) != (p->gn2 == nullptr)) return -6;
        if (p->E == 64) {
            if (cross) { MPG_ENSURE_LDS((mab_bwd_kernel<2, true, true>), lds); return mab_launch(mab_bwd_kernel<2, true, true>, p, lds, st, true); }
            MPG_ENSURE_LDS((mab_bwd_kernel<2, false, true>), lds);
            return mab_launch(mab_bwd_kernel<2, false, true>, p, lds, st, true);
        }
        if (cross) return mab_launch(mab_bwd_kernel<1, true, true>, p, lds, st, true);
        return mab_launch(mab_bwd_kernel<1, false, true>, p, lds, st, true);
    }
    if (p->E == 64 && mab_split(p->B)) {
        // two waves per jet, two jets per workgroup
        if (cross) {
            MPG_ENSURE_LDS((mab_bwd2_kernel<true>), lds);
            hipLaunchKernelGGL((mab_bwd2_kernel<true>), dim3((p->B + 1) / 2), dim3(256), lds, st, *p);
        } else {
            MPG_ENSURE_LDS((mab_bwd2_kernel<false>), lds);
            hipLaunchKernelGGL((mab_bwd2_kernel<false>), dim3((p->B + 1) / 2), dim3(256), lds, st, *p);
        }
        return (int)hipGetLastError();
    }
    if (p->E == 64) {
        if (cross) { MPG_ENSURE_LDS((mab_bwd_kernel<2, true>), lds); return mab_launch(mab_bwd_kernel<2, true>, p, lds, st, true); }
        MPG_ENSURE_LDS((mab_bwd_kernel<2, false>), lds);
        return mab_launch(mab_bwd_kernel<2, false>, p, lds, st, true);
    }
    if (cross) return mab_launch(mab_bwd_kernel<1, true>, p, lds, st, true);
    return mab_launch(mab_bwd_kernel<1, false>, p, lds, st, true);
}

extern "C" int mpg_mab_fwd(const MpgMab* p, void* stream) {
    if (const int rc = mab_check(p)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool cross = p->y != p->x;
    const int NT = p->E / 32;
    const int lds = 2 * 1024 * (3 * NT * 2 * NT + 2 * NT * 2 * NT) + 4 * 160 * NT;   // Win + Wo, Wf + biases
    const bool ln = p->ln1_w != nullptr;
    if (p->L > 32 || p->S > 32 || getenv("MPG_MAB_BIG") != nullptr) {   // large sets: a workgroup per jet, a wave per tile of 32 queries (mab_fwdN_kernel)
        if (ln && (p->ln1_b == nullptr || p->ln2_w == nullptr || p->ln2_b == nullptr || !(p->ln_eps > 0.f))) return -6;
        const int nw = (std::max(p->L, p->S) + 31) / 32;
        const dim3 grid(p->B), block(64 * nw);
        // (Win's image holds 3 key tiles' fragments at E = 64, 1 at E = 32; the rest of up to 5 behind the biases)
        const int ldsF = lds + (5 - 3 * NT / 2) * 8 * NT * 1024;
#define MPG_FWDN(NTv, LNv) do { MPG_ENSURE_LDS((mab_fwdN_kernel<NTv, LNv>), ldsF); \
        hipLaunchKernelGGL((mab_fwdN_kernel<NTv, LNv>), grid, block, ldsF, st, *p); } while (0)
        if (NT == 2) { if (ln) MPG_FWDN(2, true); else MPG_FWDN(2, false); }
        else { if (ln) MPG_FWDN(1, true); else MPG_FWDN(1, false); }
#undef MPG_FWDN
        return (int)hipGetLastError();
    }
    if (ln) {   // layer_norm=True: one wave per jet (a token's statistics run over both feature tiles)
        if (p->ln1_b == nullptr || p->ln2_w == nullptr || p->ln2_b == nullptr || !(p->ln_eps > 0.f)) return -6;
        if (p->E == 64) {
            if (cross) { MPG_ENSURE_LDS((mab_fwd_kernel<2, true, true>), lds); return mab_launch(mab_fwd_kernel<2, true, true>, p, lds, st); }
            MPG_ENSURE_LDS((mab_fwd_kernel<2, false, true>), lds);
            return mab_launch(mab_fwd_kernel<2, false, true>, p, lds, st);
        }
        if (cross) return mab_launch(mab_fwd_kernel<1, true, true>, p, lds, st);
        return mab_launch(mab_fwd_kernel<1, false, true>, p, lds, st);
    }
    if (const int nw2 = p->E == 64 ? mab_split_fwd_waves(p->B) : 0) {
        const int npair = nw2 / 2, lds2 = lds + npair * 2 * MAB_XCH, grid = (p->B + npair - 1) / npair;
        if (cross && nw2 == 4) { MPG_ENSURE_LDS((mab_fwd2_kernel<true, 4>), lds2); hipLaunchKernelGGL((mab_fwd2_kernel<true, 4>), dim3(grid), dim3(256), lds2, st, *p); }
        else if (cross) { MPG_ENSURE_LDS((mab_fwd2_kernel<true, 8>), lds2); hipLaunchKernelGGL((mab_fwd2_kernel<true, 8>), dim3(grid), dim3(512), lds2, st, *p); }
        else if (nw2 == 4) { MPG_ENSURE_LDS((mab_fwd2_kernel<false, 4>), lds2); hipLaunchKernelGGL((mab_fwd2_kernel<false, 4>), dim3(grid), dim3(256), lds2, st, *p); }
        else { MPG_ENSURE_LDS((mab_fwd2_kernel<false, 8>), lds2); hipLaunchKernelGGL((mab_fwd2_kernel<false, 8>), dim3(grid), dim3(512), lds2, st, *p); }
        return (int)hipGetLastError();
    }
    if (p->E == 64) {
        if (cross) { MPG_ENSURE_LDS((mab_fwd_kernel<2, true>), lds); return mab_launch(mab_fwd_kernel<2, true>, p, lds, st); }
        MPG_ENSURE_LDS((mab_fwd_kernel<2, false>), lds);
        return mab_launch(mab_fwd_kernel<2, false>, p, lds, st);
    }
    if (cross) return mab_launch(mab_fwd_kernel<1, true>, p, lds, st);
    return mab_launch(mab_fwd_kernel<1, false>, p, lds, st);
}

extern "C" int mpg_mab_chain_fwd(const MpgMabChain* c, void* stream) {
    if (c->n < 1 || c->n > MPG_MAB_CHAIN_MAX) return -1;
    const MpgMab& p0 = c->blk[0];
    if (const int rc = mab_check(&p0)) return rc;
    if (p0.L > 32) return -1;             // (the chain keeps a jet's rows in ONE wave's registers)
    for (int b = 0; b < c->n; ++b) {
        const MpgMab& p = c->blk[b];
        // self-attention blocks of one shape on one set of jets, each taking the rows the one before it writes
        if (p.y != p.x || p.B != p0.B || p.L != p0.L || p.S != p0.L || p.E != p0.E || p.H != p0.H || p.ignore != p0.ignore ||
            p.seed != p0.seed || p.wscale != p0.wscale || p.ascale != p0.ascale || p.out == nullptr || p.ldo % 4 || p.ln1_w != nullptr) return -2;
        if (b > 0 && (p.x != c->blk[b - 1].out || p.ldx != c->blk[b - 1].ldo)) return -2;
        if (!(p.alpha >= 0.f && p.alpha <= 1.f)) return -4;
    }
    const int nw = mab_waves(p0.B);
    const int grid = (p0.B + nw - 1) / nw;              // one jet per wave: its rows stay in registers across the blocks
    const int NT = p0.E / 32;
    const int lds = 2 * 1024 * (3 * NT * 2 * NT + 2 * NT * 2 * NT) + 4 * 160 * NT;
    hipStream_t st = (hipStream_t)stream;
    if (const int nw2 = p0.E == 64 ? mab_split_fwd_waves(p0.B) : 0) {
        // two waves per jet, two or four jets per workgroup
        const int npair = nw2 / 2, lds2 = lds + npair * 2 * MAB_XCH, grid2 = (p0.B + npair - 1) / npair;
        if (nw2 == 4) { MPG_ENSURE_LDS(mab_chain_fwd2_kernel<4>, lds2); hipLaunchKernelGGL(mab_chain_fwd2_kernel<4>, dim3(grid2), dim3(256), lds2, st, *c); }
        else { MPG_ENSURE_LDS(mab_chain_fwd2_kernel<8>, lds2); hipLaunchKernelGGL(mab_chain_fwd2_kernel<8>, dim3(grid2), dim3(512), lds2, st, *c); }
    } else if (p0.E == 64) {
        MPG_ENSURE_LDS((mab_chain_fwd_kernel<2>), lds);
        hipLaunchKernelGGL((mab_chain_fwd_kernel<2>), dim3(grid), dim3(64 * nw), lds, st, *c);
    } else {
        hipLaunchKernelGGL((mab_chain_fwd_kernel<1>), dim3(grid), dim3(64 * nw), lds, st, *c);
    }
    return (int)hipGetLastError();
}
