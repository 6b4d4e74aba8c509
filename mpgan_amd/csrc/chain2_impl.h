// The schedule of chain2.hip as a device function, shared by the stand-alone kernel (chain2_kernel: rows staged from
// memory) and by the fused edge forward (edge_fwd2_impl.h, FN: the node network fn runs as the epilogue of the workgroup
// that has just aggregated its 32 receivers -- their rows are staged straight from the LDS reduction).
// See chain2.hip for the description of the schedule itself.
#pragma once
#include "common.h"
#include "../../include/mpgan_amd.h"

#ifdef MPG_CHSTAMP  // diagnostic build (tools/chain_stamps.py): s_memtime at the phase boundaries
#define C2_STAMP(i) do { if (c2_st != nullptr) c2_st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C2_STAMP(i) do {} while (0)
#endif

namespace {

typedef unsigned int c2_u32x4 __attribute__((ext_vector_type(4)));
constexpr int C2_FB = 16 * 2 * 1024;               // one fragment buffer: [k-step][hi|lo][lane] 16 B
constexpr int C2_BIAS = 3 * 256;
constexpr int C2_LDS = 2 * C2_FB + C2_BIAS * 4;    // 68,608

template <int NU, int NS, int SL, typename F>
MPG_DEV void c2_slot(F&& unit) {                   // units u of [0, NU) that fall into slot SL of NS
    if constexpr (NU > 0) {
        constexpr int u0 = (SL * NU) / NS, u1 = ((SL + 1) * NU) / NS;
        static_for<u0, u1>(unit);
    }
}
template <bool F16, typename V>
MPG_DEV f32x16 c2_mma(const V a, const V b, const f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
MPG_DEV float4 c2_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// state of a tile between its k loop and the end of its epilogue
struct C2Tile {
    f32x16 acc;
    float hv[16], rv[16], v[16];
    float b4[4];
    uint32_t wdrop, wgate;
    int tile;
};

// keep decision of element (g, t) from the word(s) of its tile / group; DM 1 = byte mode, 2 = bit mode (common.h)
template <int DM>
MPG_DEV bool c2_keep(uint32_t word, int g, int t, uint32_t thr) {
    if constexpr (DM == 2) return (word >> (8 * g + t)) & 1u;   // word already shifted right by 4h
    else return drop_keep(word, t, thr);
}

// GATES / RESID: bit l set = layer l multiplies by a gate operand / adds a residual (known per shape: no dead arithmetic)
// SL: the LAST layer's rows are not whole 16-byte groups (N or a row stride not a multiple of 4): its output and residual
// go element by element
// Rows m0 .. m0 + 31 of the chain ``p`` (the first ``nrows`` of them are this workgroup's to write; ``p.M`` bounds them as
// well).  fb0 / fb1: the two fragment buffers (C2_FB bytes each), sbias: C2_BIAS floats, all in LDS.  ``stage`` fills fb0
// with the input rows as B fragments ([k-step][hi|lo][lane], ascale * x) and, around its own loads, calls
// first_tile(I0), bias_request(), first_tile(I1), bias_store() in that order (the first weight tiles are requested as
// early as the caller's own loads allow: vector memory completes in issue order).
// NW: waves of the workgroup.  4 (one per SIMD, 512 registers each): a wave owns up to TWO output tiles of a layer (w, w + 4), whole
// weight tiles in two register slots, the first tile's epilogue in the MFMA slots of the second.  8 (two per SIMD, 256 registers
// each: the epilogue of the eight-wave edge forward, edge_fwd1_impl.h): a wave owns ONE tile (w) in one slot; its epilogue runs
// beside its SIMD partner's MFMAs instead.  Same arithmetic per element either way.
template <bool F16, int KS0, int KS1, int KS2, int DROP, int GATES, int RESID, bool SL, int NW = 4, typename Stage>
MPG_DEV void c2_body(const MpgChain& p, const int m0, const int nrows, char* fb0, char* fb1, float* sbias, Stage&& stage,
                     unsigned long long* c2_st = nullptr) {
    typedef typename FragT<F16>::type V;
    constexpr int NL = 1 + (KS1 > 0) + (KS2 > 0);
    constexpr int NSL = 8 / NW;   // register slots (tiles per wave and layer)
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, lane16 = lane * 16;
    const int m = m0 + r;
    const bool mvalid = r < nrows && m < p.M;
    const int mc = min(m, p.M - 1);
    uint32_t seed_lo = 0, seed_hi = 0;
    if (p.seed != nullptr) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }
    const float ascale = p.ascale > 0.f ? p.ascale : 1.f;
    C2_STAMP(0);

    // biases into LDS: requested behind the input rows and the first weight tile (which the first k loop waits for anyway),
    // laid down after the rows are staged -- at the top of the kernel their latency was 1.3k clk in front of everything
    float bv[NL];
    auto bias_request = [&]() {
        static_for<0, NL>([&](auto lc) {   // (a run-time index into p.L would move the whole argument block to scratch)
            MPG_CI(l, lc);
            const int nb = p.L[l].bias != nullptr ? (p.L[l].nbias ? p.L[l].nbias : p.L[l].N) : 0;
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.L[l].bias), 0, nb * 4, 0x00020000);
            bv[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, tid * 4, 0, 0));   // (past nb: zero)
        });
    };
    auto bias_store = [&]() {
        static_for<0, NL>([&](auto lc) {
            MPG_CI(l, lc);
            const float so = p.L[l].drop_thr ? p.L[l].drop_scale : 1.f;   // the layer's dropout scale rides on bias and product
            if (NW == 4 || tid < 256) sbias[256 * l + tid] = bv[l] * so;
        });
    };

    C2_STAMP(18);
    // NSL == 2: [slot A | slot B][k-step][hi | lo]: whole tiles.  NSL == 1 (256 registers): a RING of eight k-steps over the chain's
    // whole sequence of k-steps -- item g = (layer, k-step) sits in slot g % 8; the ring starts with items 0..7 and item g + 8 is
    // requested into the slot item g has just left (a whole 16-k-step tile, 128 registers, did not fit beside the epilogue state:
    // 72 registers spilled, and every scratch access drained the weight prefetch)
#ifndef MPG_C2_RING
#define MPG_C2_RING 8   // (experiments: -DMPG_C2_RING=n; beyond 8 the eight-wave edge kernels that carry the chain as their epilogue spill)
#endif
    constexpr int RING = NSL == 1 ? MPG_C2_RING : 16;
    V wb[NSL][RING][2];
    constexpr int KSA[3] = {KS0, KS1, KS2};
    constexpr int OFF1 = KS0, OFF2 = KS0 + KS1, NITEM = KS0 + KS1 + KS2;
    auto load_item = [&](auto gc) {   // NSL == 1: item g of the sequence -> slot g % 8 (a wave without a tile in that layer reads zeros)
        MPG_CI(g, gc);
        if constexpr (g < NITEM) {
            constexpr int l = (KS2 > 0 && g >= OFF2) ? 2 : ((KS1 > 0 && g >= OFF1) ? 1 : 0);
            constexpr int ks = g - (l == 2 ? OFF2 : (l == 1 ? OFF1 : 0));
            const int mt = (p.L[l].N + 31) / 32, nfrag = mt * KSA[l];
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.L[l].Wimg), 0, w < mt ? 2 * nfrag * 1024 : 0, 0x00020000);
            wb[0][g % RING][0] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (w * KSA[l] + ks) * 1024, 0));
            wb[0][g % RING][1] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (nfrag + w * KSA[l] + ks) * 1024, 0));
        }
    };
    auto load_tile = [&](auto lc, auto bc, int tile) {   // (NSL == 2)
        MPG_CI(l, lc);
        MPG_CI(b, bc);
        constexpr int KSC = (l == 0 ? KS0 : (l == 1 ? KS1 : KS2)) <= RING ? (l == 0 ? KS0 : (l == 1 ? KS1 : KS2)) : RING;
        constexpr int KST = l == 0 ? KS0 : (l == 1 ? KS1 : KS2);   // (KSC == KST for NSL == 2: the only caller)
        const int nfrag = ((p.L[l].N + 31) / 32) * KST;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.L[l].Wimg), 0, 2 * nfrag * 1024, 0x00020000);
        static_for<0, KSC>([&](auto kc) {
            MPG_CI(ks, kc);
            wb[b][ks][0] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (tile * KST + ks) * 1024, 0));
            wb[b][ks][1] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, (nfrag + tile * KST + ks) * 1024, 0));
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // layer 0's tiles are requested AFTER the input rows: loads return in issue order, and the staging must not queue
    // behind 64 KiB of weights per wave
    auto first_tile = [&](auto bc) {
        MPG_CI(b, bc);
        if constexpr (NSL == 1) {
            if constexpr (b == 0) static_for<0, RING>([&](auto gc) { load_item(gc); });
        } else {
            if (w + NW * b < (p.L[0].N + 31) / 32) load_tile(I0{}, bc, w + NW * b);
        }
    };

    stage(first_tile, bias_request, bias_store, seed_lo, seed_hi, ascale);
    C2_STAMP(1);
    __syncthreads();
    C2_STAMP(2);

    static_for<0, NL>([&](auto lc) {
        MPG_CI(l, lc);
        constexpr int KSC = l == 0 ? KS0 : (l == 1 ? KS1 : KS2);
        constexpr bool last = l + 1 == NL;
        constexpr int NU = last ? 16 : 18;      // epilogue units of a tile: 16 elements (+ 2 fragment splits)
        const MpgChainLayer& L = p.L[l];
        const V* fin = reinterpret_cast<const V*>((l & 1) ? fb1 : fb0);
        V* fout = reinterpret_cast<V*>((l & 1) ? fb0 : fb1);
        const int MT = (L.N + 31) / 32;
        const bool actA = w < MT, actB = NSL == 2 && w + 4 < MT;
        const float zscale = (L.wscale > 0.f ? L.wscale : 1.f) * ascale, inv_z = 1.f / zscale;
        const float alpha_eff = L.act ? p.alpha : 1.f;
        const bool has_gate = L.gateH != nullptr;
        // options as uniform values for selects (a `thr ? .. : ..` per element becomes a branch per element)
        const uint32_t drop_thr = L.drop_thr, gate_thr = L.gate_thr;
        const bool drop_on = drop_thr != 0u, gdrop_on = has_gate && gate_thr != 0u;
        const float drop_s = drop_on ? L.drop_scale : 1.f;
        const float g_neg = L.gate_act ? p.alpha : 1.f, g_scale = gdrop_on ? L.gate_scale : 1.f;
        constexpr bool GATE = (GATES >> l) & 1, RES = (RESID >> l) & 1;
        const float zf = inv_z * drop_s;                       // (LeakyReLU commutes with a positive scale)
        const float g_pos = g_scale, g_ng = g_neg * g_scale;   // gate factor at h > 0 / h <= 0
        const int rowoff = m * L.ldo * 4;                      // (the launcher checks M * ldo * 4 < 2^31)
        const uint32_t drop_tag = L.drop_tag, gate_tag = L.gate_tag;
        const int LN = L.N, ldo = L.ldo;
        const __amdgpu_buffer_rsrc_t rgate = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(L.gateH), 0, L.gateH != nullptr ? (int)((size_t)p.M * L.ldh * 4) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(L.resid), 0, L.resid != nullptr ? (int)((size_t)p.M * L.ldr * 4) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            L.out, 0, L.out != nullptr ? (int)((size_t)p.M * L.ldo * 4) : 0, 0x00020000);

        // operands of a tile's epilogue that live in memory: requested before the tile's k loop
        auto request = [&](C2Tile& T, int tile) {
            T.tile = tile;
#pragma unroll
            for (int k = 0; k < 16; ++k) T.acc[k] = 0.f;
            // (buffer loads: a NULL operand has a zero-length descriptor and reads zeros -- no branch, and a branch around
            // loads makes the compiler wait for every load in flight at its end, the weight prefetch included)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = min(32 * tile + 8 * g + 4 * h, LN - 4);
                if constexpr (GATE) {
                    const f32x4 hq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgate, (mc * L.ldh + n) * 4, 0, 0));
#pragma unroll
                    for (int t = 0; t < 4; ++t) T.hv[4 * g + t] = hq[t];   // (whole-vector cast: a per-element __builtin_bit_cast(float, v[t])
                }                                                          //  of an integer vector reads element 0 four times with this compiler)
                if constexpr (RES && !(SL && last)) {
                    const f32x4 rq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, (mc * L.ldr + n) * 4, 0, 0));
#pragma unroll
                    for (int t = 0; t < 4; ++t) T.rv[4 * g + t] = rq[t];
                }
                if constexpr (RES && SL && last) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int nn = 32 * tile + 8 * g + 4 * h + t;
                        T.rv[4 * g + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, nn < LN ? (mc * L.ldr + nn) * 4 : -1, 0, 0));
                    }
                }
            }
        };
        // epilogue unit u of tile T: u < 16 element (g, t) = accumulator register u; 16, 17: the two fragment pairs
        auto unit = [&](auto uc, C2Tile& T) {
            MPG_CI(u, uc);
            if constexpr (u < 16) {
                constexpr int g = u >> 2, t = u & 3;
                const int n0 = 32 * T.tile + 8 * g + 4 * h;
                if constexpr (t == 0) {
                    const float4 b4 = *reinterpret_cast<const float4*>(sbias + 256 * l + n0);
                    T.b4[0] = b4.x; T.b4[1] = b4.y; T.b4[2] = b4.z; T.b4[3] = b4.w;
                    if constexpr (DROP == 1) {   // byte mode: one word per group of four features (thr = 0 keeps everything)
                        T.wdrop = drop_word(seed_lo, seed_hi, drop_tag, (uint32_t)m, (uint32_t)(8 * T.tile + 2 * g + h));
                        if constexpr (GATE) T.wgate = drop_word(seed_lo, seed_hi, gate_tag, (uint32_t)m, (uint32_t)(8 * T.tile + 2 * g + h));
                    }
                }
                if constexpr (u == 0 && DROP == 2) {   // bit mode: one word per tile; a site without dropout keeps every bit
                    const uint32_t wd = drop_word(seed_lo, seed_hi, drop_tag, (uint32_t)m, DROP_BIT_GRP + (uint32_t)T.tile) >> (4 * h);
                    T.wdrop = drop_on ? wd : 0xffffffffu;
                    if constexpr (GATE) {
                        const uint32_t wg = drop_word(seed_lo, seed_hi, gate_tag, (uint32_t)m, DROP_BIT_GRP + (uint32_t)T.tile) >> (4 * h);
                        T.wgate = gdrop_on ? wg : 0xffffffffu;
                    }
                }
                float x = fmaf(T.acc[u], zf, T.b4[t]);
                x = lrelu(x, alpha_eff);
                if constexpr (DROP != 0) x = drop_apply<DROP>(x, T.wdrop, 8 * g + t, t, drop_thr);
                if constexpr (GATE) {   // derivative of the forward layer's Dropout o LeakyReLU (slope alpha at h <= 0)
                    x *= T.hv[u] > 0.f ? g_pos : g_ng;
                    if constexpr (DROP != 0) x = drop_apply<DROP>(x, T.wgate, 8 * g + t, t, gate_thr);
                }
                if constexpr (RES) x += T.rv[u];
                // (rows >= M and features >= N need no zeroing: their products are finite -- packed images and staged
                // biases are zero there -- they meet only zero weights downstream, and their stores are dropped)
                T.v[u] = x;
                if constexpr (SL && last) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x), rout,
                                                          (mvalid && n0 + t < LN) ? rowoff + 4 * (n0 + t) : -1, 0, 0);
                } else if constexpr (t == 3) {
                    const bool st = mvalid && n0 + 4 <= LN;
                    __builtin_amdgcn_raw_buffer_store_b128(
                        c2_u32x4{__builtin_bit_cast(uint32_t, T.v[4 * g]), __builtin_bit_cast(uint32_t, T.v[4 * g + 1]),
                                 __builtin_bit_cast(uint32_t, T.v[4 * g + 2]), __builtin_bit_cast(uint32_t, T.v[4 * g + 3])},
                        rout, st ? rowoff + 4 * n0 : -1, 0, 0);
                }
            } else if constexpr (!last) {
                constexpr int s = u - 16;   // registers 8s .. 8s+7 are the B fragment of k-step (2 tile + s) of the next layer
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = T.v[8 * s + j] * ascale;
                V hi, lo;
                split8(vv, hi, lo);
                fout[((2 * T.tile + s) * 2 + 0) * 64 + lane] = hi;
                fout[((2 * T.tile + s) * 2 + 1) * 64 + lane] = lo;
            }
        };
        // k loop of one tile (weights wb[B]); the units of tile E's epilogue ride in its 3 KSC MFMA slots (NUE = 0: none)
        // ... and each k-step, once its three MFMAs are issued, requests the same k-step of the slot's tile of the NEXT
        // layer into the registers it has just freed: the 2 KiB per k-step and wave then trickle through the CU's L1
        // (64 B/clk for four waves) behind the MFMAs instead of blocking the wave for ~2,500 clk when asked for at once
        // (a wave without a tile there reads through a zero-length descriptor: zeros, no traffic, no branch).
        constexpr int KSN = last ? 0 : (l == 0 ? KS1 : KS2);
        int nfragn = 0;
        if constexpr (!last) nfragn = ((p.L[last ? l : l + 1].N + 31) / 32) * KSN;
        auto next_frag = [&](auto bc, auto kc, const __amdgpu_buffer_rsrc_t& rwn, int tile_n) {
            MPG_CI(B, bc);
            MPG_CI(ks, kc);
            wb[B][ks][0] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rwn, lane16, (tile_n * KSN + ks) * 1024, 0));
            wb[B][ks][1] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rwn, lane16, (nfragn + tile_n * KSN + ks) * 1024, 0));
        };
        constexpr int GOFF = l == 0 ? 0 : (l == 1 ? OFF1 : OFF2);   // (NSL == 1: the layer's first item of the ring's sequence)
        auto kloop1 = [&](C2Tile& T) {   // NSL == 1: the wave's one tile, weights from the ring
            V bh = fin[0 * 64 + lane], bl = fin[1 * 64 + lane];
            static_for<0, KSC>([&](auto kc) {
                MPG_CI(ks, kc);
                constexpr int kn = ks + 1 < KSC ? ks + 1 : KSC - 1, g = GOFF + ks;
                const V nh = fin[(kn * 2 + 0) * 64 + lane], nl = fin[(kn * 2 + 1) * 64 + lane];
                T.acc = c2_mma<F16>(wb[0][g % RING][1], bh, T.acc);
                T.acc = c2_mma<F16>(wb[0][g % RING][0], bl, T.acc);
                T.acc = c2_mma<F16>(wb[0][g % RING][0], bh, T.acc);
                load_item(std::integral_constant<int, g + RING>{});
                __builtin_amdgcn_sched_barrier(0);
                bh = nh; bl = nl;
            });
        };
        auto kloop = [&](auto bc, auto nue, C2Tile& T, C2Tile& E, const __amdgpu_buffer_rsrc_t& rwn, int tile_n) {
            MPG_CI(B, bc);
            MPG_CI(NUE, nue);
            V bh = fin[0 * 64 + lane], bl = fin[1 * 64 + lane];
            static_for<0, KSC>([&](auto kc) {
                MPG_CI(ks, kc);
                constexpr int kn = ks + 1 < KSC ? ks + 1 : KSC - 1;
                const V nh = fin[(kn * 2 + 0) * 64 + lane], nl = fin[(kn * 2 + 1) * 64 + lane];
                T.acc = c2_mma<F16>(wb[B][ks][1], bh, T.acc);
                c2_slot<NUE, 3 * KSC, 3 * ks>([&](auto uc) { unit(uc, E); });
                __builtin_amdgcn_sched_barrier(0);
                T.acc = c2_mma<F16>(wb[B][ks][0], bl, T.acc);
                c2_slot<NUE, 3 * KSC, 3 * ks + 1>([&](auto uc) { unit(uc, E); });
                __builtin_amdgcn_sched_barrier(0);
                T.acc = c2_mma<F16>(wb[B][ks][0], bh, T.acc);
                c2_slot<NUE, 3 * KSC, 3 * ks + 2>([&](auto uc) { unit(uc, E); });
                if constexpr (ks < KSN) next_frag(bc, kc, rwn, tile_n);
                __builtin_amdgcn_sched_barrier(0);
                bh = nh; bl = nl;
            });
            static_for<(KSC < KSN ? KSC : KSN), KSN>([&](auto kc) { next_frag(bc, kc, rwn, tile_n); });   // a longer next layer
        };
        int MTn = 0;
        if constexpr (!last) MTn = (p.L[l + 1].N + 31) / 32;

        // descriptors of the next layer's image for this wave's two slots (zero length where it has no tile there)
        const void* imgn = last ? p.L[l].Wimg : p.L[last ? l : l + 1].Wimg;
        const __amdgpu_buffer_rsrc_t rwnA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(imgn), 0, (!last && w < MTn) ? 2 * nfragn * 1024 : 0, 0x00020000);
        C2Tile TA;
        if constexpr (NSL == 1) {
            // (the ring's sequence advances in every wave: one without a tile in this layer only issues the layer's requests --
            // empty descriptors for its own missing tiles, the next layers' fragments where it has one)
            if (actA) {
                request(TA, w);
                kloop1(TA);
            } else {
                static_for<0, KSC>([&](auto kc) { load_item(std::integral_constant<int, GOFF + decltype(kc)::value + RING>{}); });
            }
            C2_STAMP(3 + 5 * l);
            C2_STAMP(4 + 5 * l);
            if (actA) static_for<0, NU>([&](auto uc) { unit(uc, TA); });
        } else {
        if (actA) {
            request(TA, w);
            kloop(I0{}, I0{}, TA, TA, rwnA, w);
        } else if constexpr (!last) {
            if (w < MTn) load_tile(std::integral_constant<int, l + 1>{}, I0{}, w);            // (no loop to ride in)
        }
        C2_STAMP(3 + 5 * l);
        }
        if constexpr (NSL == 2) {
            const __amdgpu_buffer_rsrc_t rwnB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(imgn), 0, (!last && w + 4 < MTn) ? 2 * nfragn * 1024 : 0, 0x00020000);
            C2Tile TB;
            if (actB) {
                request(TB, w + 4);
                kloop(I1{}, std::integral_constant<int, NU>{}, TB, TA, rwnB, w + 4);          // ... with tile A's epilogue
            } else if constexpr (!last) {
                if (w + 4 < MTn) load_tile(std::integral_constant<int, l + 1>{}, I1{}, w + 4);
            }
            C2_STAMP(4 + 5 * l);
            if (actB) static_for<0, NU>([&](auto uc) { unit(uc, TB); });
            else if (actA) static_for<0, NU>([&](auto uc) { unit(uc, TA); });
        }
        C2_STAMP(5 + 5 * l);
        __syncthreads();
        C2_STAMP(6 + 5 * l);
    });
}

// Stage rows m0 .. m0 + 31 of the chain's input (A | A2, summed slabs, optional input dropout gate and its in_out copy) as
// B fragments into fb0: unit = (k-step, lane) = 8 features of one row.  Only the first ``nrows`` rows are this workgroup's
// (the others are staged as zeros and their in_out is not written).  The ``stage`` argument of c2_body for callers whose
// rows come from memory: the stand-alone kernel and the fused data-gradient kernel's prologue.
template <bool F16, int KS0, int DROP, int NW = 4, typename FT, typename BR, typename BS>
MPG_DEV void c2_stage_rows(const MpgChain& p, const int m0, const int nrows, char* fb0, FT&& first_tile, BR&& bias_request, BS&& bias_store,
                           const uint32_t seed_lo, const uint32_t seed_hi, const float ascale, unsigned long long* c2_st = nullptr) {
    typedef typename FragT<F16>::type V;
    const int tid = threadIdx.x;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // ---- stage the input rows as B fragments: unit = (k-step, lane) = 8 features of one row
    {
        const int K = p.L[0].K;
        V* fb = reinterpret_cast<V*>(fb0);
        const bool fast = (p.lda % 4 == 0) && (p.K1 % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0) &&
                          p.a_slabs == 1 && (p.K1 == K || ((p.lda2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.A2) & 15) == 0))) &&
                          (p.in_out == nullptr || ((p.ld_in_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in_out) & 15) == 0)));
        if (fast) {
            // every load of the thread's (up to four) units first, then the arithmetic
            constexpr int NTH = 64 * NW;
            constexpr int NI = (KS0 * 64 + NTH - 1) / NTH;
            float4 x[NI][2];
            const float* a2 = p.A2 != nullptr ? p.A2 : p.A;
            // unit u = (row rr, k-step ks, half hh), row-major: the 2 KS0 units of a row sit on adjacent lanes, so a wave
            // instruction takes whole 128-byte lines of two or three rows (lane = row made every lane its own line: 64 line
            // requests per instruction, 2k clk of the texture path per workgroup before the first weight tile could be asked for)
            static_for<0, NI>([&](auto ic) {
                MPG_CI(i, ic);
                const int u = min(tid + NTH * i, KS0 * 64 - 1), rr = u / (2 * KS0), rem = u - rr * (2 * KS0), ks = rem >> 1, hh = rem & 1;
                const size_t row = (size_t)min(m0 + rr, p.M - 1);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int f = 16 * ks + 8 * half + 4 * hh;
                    const int fc = min(f, K - 4);
                    x[i][half] = fc < p.K1 ? c2_ld4(p.A + row * p.lda + fc) : c2_ld4(a2 + row * p.lda2 + (fc - p.K1));
                }
            });
            C2_STAMP(19);
            first_tile(I0{});     // (its 28+ KiB per wave arrive while the rows are converted)
            bias_request();
            C2_STAMP(20);
            const uint32_t in_thr = p.in_thr;
            const bool in_on = in_thr != 0u;
            const float in_s = in_on ? p.in_scale : 1.f;
            const __amdgpu_buffer_rsrc_t rio = __builtin_amdgcn_make_buffer_rsrc(
                p.in_out, 0, p.in_out != nullptr ? (int)((size_t)p.M * p.ld_in_out * 4) : 0, 0x00020000);
            static_for<0, NI>([&](auto ic) {
                MPG_CI(i, ic);
                const int u = tid + NTH * i, uc = min(u, KS0 * 64 - 1), rr = uc / (2 * KS0), rem = uc - rr * (2 * KS0), ks = rem >> 1, hh = rem & 1;
                const int ln = rr + 32 * hh;
                const int mm = m0 + rr;
                const bool live = rr < nrows && mm < p.M;
                float v[8];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int f = 16 * ks + 8 * half + 4 * hh;
                    float x4[4] = {x[i][half].x, x[i][half].y, x[i][half].z, x[i][half].w};
                    uint32_t wd = 0;
                    if constexpr (DROP == 2) wd = drop_word(seed_lo, seed_hi, p.in_tag, (uint32_t)mm, DROP_BIT_GRP + (uint32_t)(f >> 5)) >> (f & 31);
                    if constexpr (DROP == 1) wd = drop_word(seed_lo, seed_hi, p.in_tag, (uint32_t)mm, (uint32_t)(f >> 2));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float xv = (live && f + e < K) ? x4[e] : 0.f;
                        if constexpr (DROP != 0) {
                            const bool keep = (DROP == 2 ? ((wd >> e) & 1u) != 0u : drop_keep(wd, e, in_thr)) || !in_on;
                            xv = keep ? xv * in_s : 0.f;
                        }
                        x4[e] = xv;
                        v[4 * half + e] = xv * ascale;
                    }
                    const bool st = live && f + 4 <= K && u < KS0 * 64;
                    __builtin_amdgcn_raw_buffer_store_b128(
                        c2_u32x4{__builtin_bit_cast(uint32_t, x4[0]), __builtin_bit_cast(uint32_t, x4[1]), __builtin_bit_cast(uint32_t, x4[2]),
                                 __builtin_bit_cast(uint32_t, x4[3])},
                        rio, st ? (int)(((size_t)mm * p.ld_in_out + f) * 4) : -1, 0, 0);
                }
                V hi, lo;
                split8(v, hi, lo);
                if (u < KS0 * 64) {
                    fb[(ks * 2 + 0) * 64 + ln] = hi;
                    fb[(ks * 2 + 1) * 64 + ln] = lo;
                }
            });
            first_tile(I1{});
            bias_store();
        } else {
            bias_request();
            bias_store();
            for (int u = tid; u < KS0 * 64; u += 64 * NW) {
                const int ks = u >> 6, ln = u & 63, rr = ln & 31, hh = ln >> 5;
                const int mm = m0 + rr;
                float v[8];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int f = 16 * ks + 8 * half + 4 * hh;
                    float x4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (rr < nrows && mm < p.M) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = f + e;
                            if (k < p.K1) {
                                for (int sl = 0; sl < p.a_slabs; ++sl) x4[e] += p.A[sl * p.a_slab_stride + (size_t)mm * p.lda + k];
                            } else if (k < K) {
                                x4[e] = p.A2[(size_t)mm * p.lda2 + (k - p.K1)];
                            }
                        }
                        if (p.in_thr) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                x4[e] = (f + e < K && drop_keep_f(seed_lo, seed_hi, p.in_tag, (uint32_t)mm, f + e, p.in_thr)) ? x4[e] * p.in_scale : 0.f;
                        }
                        if (p.in_out != nullptr) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (f + e < K) p.in_out[(size_t)mm * p.ld_in_out + f + e] = x4[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * half + e] = x4[e] * ascale;
                }
                V hi, lo;
                split8(v, hi, lo);
                fb[(ks * 2 + 0) * 64 + ln] = hi;
                fb[(ks * 2 + 1) * 64 + ln] = lo;
            }
            first_tile(I0{});
            first_tile(I1{});
        }
    }
}

}  // namespace
