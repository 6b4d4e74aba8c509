// Fused edge network, forward -- the EIGHT-WAVE form: two waves per SIMD, every wave one sender at a time.
//
// Same function, same memory formats (a | c rows in, agg rows, sign words and parked E2 fragments out) and the same
// per-element arithmetic as edge_fwd2_impl.h (bias first, then per k-step lo*hi, hi*lo, hi*hi, k ascending); what changes
// is who does what.  The four-wave kernel gives one wave per SIMD all 512 registers and walks its senders in pairs; it is
// bound by VALU ISSUE: ~2,700 vector instructions against 540 MFMAs per pair, one wave issuing one vector instruction per
// ~5-7 clk (tools/ubench/valu_rate2.hip), where a second wave on the SIMD issues beside the first at the same rate.  Here a
// workgroup has eight waves of <= 256 registers; a wave owns ONE sender per round (agg 96 + layer-2 accumulators 80 or E2
// fragments 80 + two layer-3 tiles 32 + weight fragments), so that one wave's MFMA-free stretches (the E2 split, the e1
// build) run beside its SIMD partner's MFMAs.  Weight delivery: W3 hi | lo from LDS, 2 KiB per three MFMAs per wave
// (85 B/clk/CU of 256); W2 hi | lo streams from L2 through a five-tile register ring, one k-step ahead.
// With 256 registers there are no AGPR copies: the accumulators are VGPRs the epilogues read in place (the four-wave form
// spends 500 of its 2,700 vector instructions per pair on v_accvgpr_read / _write).
#pragma once
#include "edge_fwd2_impl.h"
#include <stdlib.h>

#ifndef MPG_F1_STAGGER
#define MPG_F1_STAGGER 0   // experiment (tools/ubench/fwd_bench.hip): waves 4..7 start this many s_sleep(16) (~1k clk each) late
#endif

#ifdef MPG_F1_STAMP   // diagnostic build (tools/ubench/fwd_bench.hip): s_memtime per section of a wave's senders, summed per wave
__device__ unsigned long long f1_stamps[4096 * 8 * 8];
#define F1_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); f1_acc[i] += t_ - f1_t; f1_t = t_; }
#else
#define F1_STAMP(i)
#endif

namespace {

// W2 streams from L2 through the CU's L1 (64 B/clk): 60 KiB per sender, and the eight waves of a workgroup walk layer 2 in lock step
// (a round's layer 2 took 7.9k clk for 90 MFMAs per wave: 480 KiB through 64 B/clk).  The LDS left over beside W3 holds 21 of
// W2's 60 fragments -- the hi halves of tiles 0..2 for every k-step and of tile 3 for k-steps 0..2 -- and those are read from there.
constexpr int F1_W2L_OFF = F2_LDS_BYTES, F1_W2L_N = 21;
constexpr int F1_LDS_BYTES = F1_W2L_OFF + F1_W2L_N * 1024;
static_assert(F1_LDS_BYTES <= 163840, "LDS plan exceeds 160 KiB");
#ifdef MPG_F1_NO_W2L   // (experiment: everything of W2 streamed)
constexpr bool f1_w2_in_lds(int, int) { return false; }
#else
constexpr bool f1_w2_in_lds(int m, int k) { return m < 3 || (m == 3 && k < 3); }
#endif
constexpr int f1_w2_slot(int m, int k) { return m < 3 ? m * (T1 * 2) + k : 3 * (T1 * 2) + k; }

constexpr int F1_NW = 8;
static_assert(F1_NW * H1 * 4 == F2_C_BYTES, "one row of c per wave in the four-wave kernel's two-rows-per-wave area");

MPG_DEV void fill_lds_dma8(void* dst, const void* src, int bytes, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int c = wave; c < bytes / 1024; c += F1_NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(src) + c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(static_cast<char*>(dst) + c * 1024), 16, 0, 0);
}

// FN / cp / cp2: as edge_fwd_body's (the node network and the next layer's a | c projection as the workgroup's epilogue), run
// by c2_body's eight-wave form.
// LT: which of the hi/lo cross terms the two dense layers issue (the sender loop is bound by MFMA issue:
// 270 MFMAs per sender in the three-term form, two waves to a SIMD).  0 = all three terms in both layers (fp32-level products).
// Bit 0: layer 3 on TWO terms -- E2 as the ONE fp16 value that is parked anyway times W3 hi + lo (the E2 lo fragments, their
// 40 registers and their split arithmetic are gone; 210 MFMAs); bit 1: layer 2 likewise, E1 as one fp16 value times W2 hi + lo
// (180 MFMAs with both).  An activation rounded to fp16 carries a relative error of 2^-12 rms that is independent from edge to
// edge, so it averages out over the senders of agg and over the edges of every gradient sum; a pre-activation of layer 3 / 2
// is then known to ~1e-4 of its scale instead of ~5e-7, i.e. the LeakyReLU branch of elements that close to zero may differ from
// fp32's -- the backward takes the forward's own sign bits either way.
template <int DROP, bool SIGN, int NQ, int FN, int LT = 0>
MPG_DEV void edge_fwd1_body(const MpgEdgeFwd& p, const MpgChain* const cp = nullptr, const MpgChain* const cp2 = nullptr) {
    typedef f16x8 V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MPG_F1_STAMP
    const unsigned long long f1_k0 = __builtin_amdgcn_s_memtime(), f1_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int RB = (p.N + 31) / 32;
    int bid = blockIdx.x;
    const int sc = bid % p.SC; bid /= p.SC;
    const int rb = bid % RB;
    const int b = p.order != nullptr ? p.order[bid / RB] : bid / RB;
    const int i = rb * 32 + r;
    const bool vi = i < p.N;
    const int JC = (p.N + p.SC - 1) / p.SC;
    const int jbeg = sc * JC, jend = min(p.N, jbeg + JC);
    const int ldac = p.ld_ac ? p.ld_ac : H1;

    const __amdgpu_buffer_rsrc_t r2 = img_rsrc(p.W2img, 2 * NF2);   // W2 hi | lo
    const int lane16 = lane * 16;
    const V* g3 = reinterpret_cast<const V*>(p.W3img);
    V* l3 = reinterpret_cast<V*>(smem);
    float4* la = reinterpret_cast<float4*>(smem + F2_W_BYTES);
    float* lb2 = reinterpret_cast<float*>(smem + F2_W_BYTES + F2_A_BYTES);
    float* lb3 = lb2 + H2;
    float* lcw = lb3 + H3 + w * H1;                                            // this wave's row of c
    unsigned short* lst = reinterpret_cast<unsigned short*>(smem + F2_W_BYTES + F2_A_BYTES + F2_B_BYTES + F2_C_BYTES);
    int* lnv = reinterpret_cast<int*>(lst + 180);   // (behind the list's 360 bytes)
    float* lmk = reinterpret_cast<float*>(smem + F2_MK_OFF);   // the listed senders' mask entries (edge_fwd2_impl.h)

    // ---- prologue (as the four-wave kernel's, on 512 threads): small loads first, W3's image by LDS-DMA behind them
    static_assert(H2 + H3 <= 512 && T1 * 4 * 64 == 512 + 256, "the prologue's register sets");
    const int tb = min(tid, H2 + H3 - 1);
    const float bv = tb < H2 ? p.b2[tb] : p.b3[tb - H2];
    float4 av[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = min(tid + 512 * u, T1 * 4 * 64 - 1);
        const int ln = t & 63, qsu = t >> 6, rr = ln & 31, hh = ln >> 5, ii = min(rb * 32 + rr, p.N - 1);
        av[u] = ld4(p.a + (size_t)(b * p.N + ii) * ldac + 8 * qsu + 4 * hh);
    }
    fill_lds_dma8(l3, g3, 2 * NF3 * 1024, tid);
    for (int c = w; c < F1_W2L_N; c += F1_NW) {   // W2's resident fragments: slot c <- fragment (tile m, k-step k) of the hi image
        const int m = c < 3 * (T1 * 2) ? c / (T1 * 2) : 3, k = c - m * (T1 * 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(p.W2img) + (m * (T1 * 2) + k) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + F1_W2L_OFF + c * 1024), 16, 0, 0);
    }
    if constexpr (NQ > 0)
        for (int t = tid; t < NQ * H1; t += 512) reinterpret_cast<float*>(smem + F2_Q_OFF)[t] = p.wq[t] * SC_A;
    if (tid < H2 + H3) lb2[tid] = bv * (tid < H2 ? SC_E2 : SC_E3);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + 512 * u;
        const bool in = rb * 32 + (t & 31) < p.N;
        if (t < T1 * 4 * 64)
            la[t] = in ? make_float4(av[u].x * SC_A, av[u].y * SC_A, av[u].z * SC_A, av[u].w * SC_A) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool whole = p.N <= F2_LIST_MAX;
    const int lbeg = whole ? 0 : jbeg, lend = whole ? p.N : jend;
    if (w == 0) {
        int cnt = 0;
        for (int j0 = lbeg; j0 < lend; j0 += 64) {
            const int j = j0 + lane;
            const float mv = (j < lend && p.mask != nullptr) ? p.mask[b * p.N + j] : 1.f;
            const bool ok = j < lend && (!(p.skip_masked & 1) || mv != 0.f);
            const unsigned long long bits = __ballot(ok);
            const int pos = cnt + __popcll(bits & ((1ull << lane) - 1ull));
            if (ok) { lst[pos] = (unsigned short)j; lmk[pos] = mv; }
            cnt += __popcll(bits);
        }
        if (lane == 0) *lnv = cnt;
    }
    __syncthreads();
    int nvalid = __builtin_amdgcn_readfirstlane(*lnv);
    if (whole) {
        const int per = (nvalid + p.SC - 1) / p.SC, l0 = min(nvalid, sc * per);
        lst += l0;
        lmk += l0;
        nvalid = min(per, nvalid - l0);
    }

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    const uint32_t lb3hi = lds_base(smem, lane16), lb3lo = lds_base(smem, NF3 * 1024 + lane16);
    const uint32_t lbla = lds_base(smem, F2_W_BYTES + lane16);
    const uint32_t lbc = lds_base(smem, F2_W_BYTES + F2_A_BYTES + F2_B_BYTES + w * (H1 * 4) + 16 * h);
    const uint32_t lbq = lds_base(smem, F2_Q_OFF + 16 * h);
    // (the bias columns through one opaque base + immediate offsets: plain pointer arithmetic makes the compiler keep one
    // address register per 16-byte group, hoisted out of the sender loop -- 44 registers this kernel does not have)
    const uint32_t lbb = lds_base(smem, F2_W_BYTES + F2_A_BYTES + 16 * h);
    const uint32_t lbw2 = lds_base(smem, F1_W2L_OFF + lane16);

    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(p.sign3, 0, SIGN ? p.B * RB * p.N * (T3 * 32 * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(p.stageE2, 0, (SIGN && p.stageE2 != nullptr) ? p.B * RB * p.N * (NFR2 * 1024) : 0, 0x00020000);

    f32x16 agg[T3];
#pragma unroll
    for (int m = 0; m < T3; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) agg[m][k] = 0.f;

    // the sender's row of c is requested one round ahead
    float pc0, pc1;
    float pes[NQ > 0 ? NQ : 1];
    auto prefetch = [&](int sn) {
        const int jn = __builtin_amdgcn_readfirstlane((int)lst[max(0, min(sn, nvalid - 1))]);
        const float* cj = p.c + (size_t)(b * p.N + jn) * ldac;
        pc0 = cj[lane];
        pc1 = cj[64 + (lane & 31)];
        if constexpr (NQ > 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) pes[q] = p.es[((size_t)(b * p.N + jn) * NQ + q) * p.N + (vi ? i : 0)];
        }
    };
    prefetch(w);
#if MPG_F1_STAGGER
    if (w >= 4)
        for (int t = 0; t < MPG_F1_STAGGER; ++t) __builtin_amdgcn_s_sleep(16);
#endif
#ifdef MPG_F1_STAMP
    unsigned long long f1_acc[8] = {}, f1_t = __builtin_amdgcn_s_memtime();
    const unsigned long long f1_t0 = f1_t;
#endif
    for (int s = w; s < nvalid; s += F1_NW) {
        F1_STAMP(4)
        const int jj = __builtin_amdgcn_readfirstlane((int)lst[s]);
        const float mj = lmk[s];
        float mjs = mj * p.dscale * (1.f / SC_E3);   // (the layer-3 output carries SC_E3)
        if (p.nbr != nullptr) {
            // (the lane's row index behind an opaque copy: left to itself the compiler hoists the 64-bit row address out of the
            // sender loop, has no register pair for it across layer 2, and reloads it from scratch -- with a vmcnt(0) -- every sender)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int irow = min(rb * 32 + (ln & 31), p.N - 1);   // (= i for the lanes whose receiver exists; the others' sums are never stored)
            const unsigned int wb = p.nbr[(size_t)(b * p.N + irow) * ((p.N + 31) >> 5) + (jj >> 5)];
            mjs = ((wb >> (jj & 31)) & 1u) ? mjs : 0.f;
        }
        // (the dropout row of this (receiver, sender) from the lane id behind an opaque copy: (b N + i) N is loop-invariant, the
        // compiler hoists it, finds no register for it across layer 2 in the dropout variants and reloads it from scratch at the
        // top of every sender -- behind a vmcnt(0) that also waits for the W2 fragments just requested)
        int lnr = lane;
        asm volatile("" : "+v"(lnr));
        const uint32_t erow = (uint32_t)((b * p.N + rb * 32 + (lnr & 31)) * p.N + jj);
        lcw[lane] = pc0 * SC_A;
        if (lane < H1 - 64) lcw[64 + lane] = pc1 * SC_A;
        float esv[NQ > 0 ? NQ : 1];
        if constexpr (NQ > 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) esv[q] = pes[q];
        }
        prefetch(s + F1_NW);

        F1_STAMP(0)
        // ---- layer 2: Z2 = W2' E1 + b2, k-outer; e1 = drop(lrelu(a_i + c_j)) built one k-step ahead; W2's fragments in a
        //      ring of five tiles: tile m's pair of k-step k + 1 is requested behind tile m's MFMAs of k-step k
        f32x16 acc[T2];
        {
            constexpr int KS = T1 * 2;
            V eh[2], el[2];          // [buffer] e1 fragment (hi, lo) of a k-step
            V wh[T2], wl[T2];        // [tile] W2 fragments of the k-step in flight
            float v1[8];
            PairSplit<V> ps1[4];
            f32x4 a4[2], c4[2];
            f32x4 q4[NQ > 0 ? NQ : 1][2];
            auto load_w = [&](auto kc, auto mc) {
                MPG_CI(k, kc); MPG_CI(m, mc);
                // (a resident hi fragment is read from LDS one TILE ahead, load_wl below: a whole k-step ahead it would sit in
                // registers for ~1k clk -- twelve the dropout variants do not have)
                if constexpr (!f1_w2_in_lds(m, k)) wh[m] = img_frag<V>(r2, lane16, m * KS + k);
                wl[m] = img_frag<V>(r2, lane16, NF2 + m * KS + k);
            };
            V whl[2];   // resident hi fragments: [tile parity], the next tile's requested in front of this tile's MFMAs
            auto load_wl = [&](auto kc, auto mc) {
                MPG_CI(k, kc); MPG_CI(m, mc);
                if constexpr (k < KS && m < T2) {
                    if constexpr (f1_w2_in_lds(m, k)) whl[m & 1] = lds_frag<V>(lbw2, f1_w2_slot(m, k) * 1024);
                }
            };
            auto load_a = [&](auto kc) {
                MPG_CI(k, kc);
                a4[0] = lds_frag<f32x4>(lbla, (k * 2 + 0) * 1024);
                a4[1] = lds_frag<f32x4>(lbla, (k * 2 + 1) * 1024);
#pragma unroll
                for (int uh = 0; uh < 2; ++uh) c4[uh] = lds_frag<f32x4>(lbc, (16 * k + 8 * uh) * 4);
                if constexpr (NQ > 0) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int uh = 0; uh < 2; ++uh) q4[q][uh] = lds_frag<f32x4>(lbq, (q * H1 + 16 * k + 8 * uh) * 4);
                }
            };
            // build units of the e1 fragment of k-step k = (q, s): 8 element units + 4 pairs x 2 halves = 16
            auto buildA = [&](auto kc, auto uc) {
                MPG_CI(k, kc); MPG_CI(u, uc);
                constexpr int q = k >> 1, s2 = k & 1;
                if constexpr (u < 8) {
                    constexpr int uh = u >> 2, t = u & 3;  // element 4 uh + t  <->  feature 32q + 16s + 8uh + 4h + t
                    float cc = c4[uh][t] + a4[uh][t];
                    if constexpr (NQ > 0) {
#pragma unroll
                        for (int qq = 0; qq < NQ; ++qq) cc = fmaf(esv[qq], q4[qq][uh][t], cc);
                    }
                    const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, q, 4 * s2 + 2 * uh + h, h);
                    v1[u] = drop_apply<DROP>(lrelu(cc, p.alpha), wd, 16 * s2 + 8 * uh + t, t, p.thr);
                } else {
                    constexpr int pr = (u - 8) >> 1;
                    if constexpr (LT & 2) {   // E1 as one fp16 value: no lo half
                        if constexpr (((u - 8) & 1) == 0) {
                            const f16x2 hp = {(_Float16)v1[2 * pr], (_Float16)v1[2 * pr + 1]};
                            eh[k & 1][2 * pr] = hp[0]; eh[k & 1][2 * pr + 1] = hp[1];
                        }
                    } else {
                        if constexpr (((u - 8) & 1) == 0) ps1[pr].first(v1[2 * pr], v1[2 * pr + 1]);
                        else ps1[pr].second(v1[2 * pr + 1], eh[k & 1], el[k & 1], 2 * pr);
                    }
                }
            };
            using K0 = std::integral_constant<int, 0>;
            static_for<0, T2>([&](auto mc) { load_w(K0{}, mc); });
            load_wl(K0{}, K0{});
            load_a(K0{});
            static_for<0, 16>([&](auto uc) { buildA(K0{}, uc); });
#pragma unroll
            for (int m = 0; m < T2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = lds_frag<f32x4>(lbb, (32 * m + 8 * g) * 4);
                    acc[m][4 * g + 0] = b4[0]; acc[m][4 * g + 1] = b4[1]; acc[m][4 * g + 2] = b4[2]; acc[m][4 * g + 3] = b4[3];
                }
            static_for<0, KS>([&](auto kc) {
                MPG_CI(k, kc);
                if constexpr (k + 1 < KS) load_a(std::integral_constant<int, k + 1>{});
                const V bh0 = eh[k & 1], bl0 = el[k & 1];
                static_for<0, T2>([&](auto mc) {
                    MPG_CI(m, mc);
                    auto slot = [&](auto slc) {
                        MPG_CI(SL, slc);
                        if constexpr (k + 1 < KS) f2_slot<16, 13, SL - 2>([&](auto uc) { buildA(std::integral_constant<int, k + 1>{}, uc); });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    // next tile's resident fragment (the first tile of the next k-step behind the last tile)
                    if constexpr (m + 1 < T2) load_wl(kc, std::integral_constant<int, m + 1>{});
                    else load_wl(std::integral_constant<int, k + 1>{}, std::integral_constant<int, 0>{});
                    V a_h;
                    if constexpr (f1_w2_in_lds(m, k)) a_h = whl[m & 1]; else a_h = wh[m];
                    const V a_l = wl[m];
                    acc[m] = f2_mma(a_l, bh0, acc[m]); slot(std::integral_constant<int, 3 * m + 0>{});
                    if constexpr (!(LT & 2)) acc[m] = f2_mma(a_h, bl0, acc[m]);
                    slot(std::integral_constant<int, 3 * m + 1>{});
                    acc[m] = f2_mma(a_h, bh0, acc[m]);
                    if constexpr (k + 1 < KS) load_w(std::integral_constant<int, k + 1>{}, mc);
                    slot(std::integral_constant<int, 3 * m + 2>{});
                });
            });
        }

        F1_STAMP(1)
        // ---- E2 = drop(lrelu(Z2)) as B fragments (hi, lo): tile mm, k-step half s -> fragment 2 mm + s
        f2_u32x4 e2h[T2 * 2], e2l[(LT & 1) ? 1 : T2 * 2];
        const int sts = ((b * RB + rb) * p.N + jj) * (NFR2 * 1024);
        static_for<0, T2>([&](auto mc) {
            MPG_CI(mm, mc);
            uint32_t wd2 = 0u;
            if constexpr (DROP == 2) wd2 = drop_tile_word<2>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow, mm, 0, h);
            float x2[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int g = u >> 2, t = u & 3;
                uint32_t wd = wd2;
                if constexpr (DROP == 1) wd = drop_tile_word<1>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow, mm, 2 * g + h, h);
                x2[u] = drop_apply<DROP>(lrelu(acc[mm][u], p.alpha), wd, 8 * g + t, t, p.thr);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                V hi, lo;
                if constexpr (LT & 1) {   // E2 as the one fp16 value that is parked
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f16x2 hp = {(_Float16)x2[8 * s2 + j], (_Float16)x2[8 * s2 + j + 1]};
                        hi[j] = hp[0]; hi[j + 1] = hp[1];
                    }
                } else {
                    split8(x2 + 8 * s2, hi, lo);
                    e2l[2 * mm + s2] = __builtin_bit_cast(f2_u32x4, lo);
                }
                e2h[2 * mm + s2] = __builtin_bit_cast(f2_u32x4, hi);
                if constexpr (SIGN) __builtin_amdgcn_raw_buffer_store_b128(e2h[2 * mm + s2], rsE, lane16, sts + (2 * mm + s2) * 1024, 0);
            }
        });

        F1_STAMP(2)
        // ---- layer 3: Z3 = W3' E2 + b3 tile by tile; the epilogue of tile m - 1 (sign bit, LeakyReLU, dropout, m_j-weighted sum
        //      into agg) rides in the MFMA slots of tile m
        {
            constexpr int KS = T2 * 2;
            f32x16 a3[2];   // [tile parity]
            uint32_t sgn[T3 / 2] = {0u, 0u, 0u};
            auto bias_init = [&](auto mc) {
                MPG_CI(m, mc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = lds_frag<f32x4>(lbb, (H2 + 32 * m + 8 * g) * 4);
                    a3[m & 1][4 * g + 0] = b4[0]; a3[m & 1][4 * g + 1] = b4[1]; a3[m & 1][4 * g + 2] = b4[2]; a3[m & 1][4 * g + 3] = b4[3];
                }
            };
            auto epi3 = [&](auto mc, auto uc) {
                MPG_CI(mm, mc); MPG_CI(e, uc);
                constexpr int g = e >> 2, t = e & 3;
                const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, mm, 2 * g + h, h);
                const float z = a3[mm & 1][e];
                if constexpr (SIGN) sgn[mm >> 1] = __builtin_amdgcn_alignbit(sgn[mm >> 1], __builtin_bit_cast(uint32_t, z), 31);
                const float x = drop_apply<DROP>(lrelu(z, p.alpha), wd, 8 * g + t, t, p.thr);
                agg[mm][e] += mjs * x;
            };
            bias_init(std::integral_constant<int, 0>{});
            static_for<0, T3>([&](auto mc) {
                MPG_CI(m, mc);
                V ah[2], al[2];
                ah[0] = lds_frag<V>(lb3hi, (m * KS) * 1024);
                al[0] = lds_frag<V>(lb3lo, (m * KS) * 1024);
                static_for<0, KS>([&](auto kc) {
                    MPG_CI(k, kc);
                    if constexpr (k + 1 < KS) {
                        ah[(k + 1) & 1] = lds_frag<V>(lb3hi, (m * KS + k + 1) * 1024);
                        al[(k + 1) & 1] = lds_frag<V>(lb3lo, (m * KS + k + 1) * 1024);
                    }
                    const V bh0 = __builtin_bit_cast(V, e2h[k]), bl0 = __builtin_bit_cast(V, e2l[(LT & 1) ? 0 : k]);
                    auto slot = [&](auto slc) {
                        MPG_CI(SL, slc);   // 3 KS - 3 slots: the last k-step's are left to the next tile's bias load
                        if constexpr (m > 0) f2_slot<16, 3 * KS - 3, SL>([&](auto uc) { epi3(std::integral_constant<int, m - 1>{}, uc); });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const V a_h = ah[k & 1], a_l = al[k & 1];
                    a3[m & 1] = f2_mma(a_l, bh0, a3[m & 1]); slot(std::integral_constant<int, 3 * k + 0>{});
                    if constexpr (!(LT & 1)) a3[m & 1] = f2_mma(a_h, bl0, a3[m & 1]);
                    slot(std::integral_constant<int, 3 * k + 1>{});
                    a3[m & 1] = f2_mma(a_h, bh0, a3[m & 1]); slot(std::integral_constant<int, 3 * k + 2>{});
                    if constexpr (k == KS - 1 && m + 1 < T3) {
                        bias_init(std::integral_constant<int, m + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
            static_for<0, 16>([&](auto uc) { epi3(std::integral_constant<int, T3 - 1>{}, uc); });
            if constexpr (SIGN) {
                const int soff = ((b * RB + rb) * p.N + jj) * (T3 * 32 * 4);
#pragma unroll
                for (int q = 0; q < T3 / 2; ++q) __builtin_amdgcn_raw_buffer_store_b32(sgn[q], rsS, lane * 4, soff + q * 256, 0);
            }
        }
        F1_STAMP(3)
    }

    {   // ---- behind the sender loop (a scope of its own: see the ids below)
#ifdef MPG_F1_STAMP
    if (lane == 0) {
        unsigned long long* o = f1_stamps + ((size_t)blockIdx.x * 8 + w) * 8;
        for (int q = 0; q < 5; ++q) o[q] = f1_acc[q];
        o[5] = (nvalid - w + F1_NW - 1) / F1_NW; o[6] = f1_t0; o[7] = __builtin_amdgcn_s_memtime();
    }
#endif
    // ---- reduce the eight waves' partial sums through LDS (the weight area holds four waves' worth: waves 4..7 hand theirs to
    //      waves 0..3 first) and write agg[b, i, :]
    // (the thread's ids are derived afresh behind an opaque copy: what the epilogue makes of them -- lane * 16, row offsets -- is
    // otherwise computed in the prologue, kept across a sender loop that has no register to spare, and spilled to scratch)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int tid = tid_e, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int i = rb * 32 + r;
    const bool vi = i < p.N;
    __syncthreads();  // everyone is done with the weight copy
    float* red = reinterpret_cast<float*>(smem);
    if (w >= 4) {
#pragma unroll
        for (int m = 0; m < T3; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) red[(((w - 4) * T3 + m) * 16 + k) * 64 + lane] = agg[m][k];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
        for (int m = 0; m < T3; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) agg[m][k] += red[((w * T3 + m) * 16 + k) * 64 + lane];
#pragma unroll
        for (int m = 0; m < T3; ++m)
#pragma unroll
            for (int k = 0; k < 16; ++k) red[((w * T3 + m) * 16 + k) * 64 + lane] = agg[m][k];
    }
    __syncthreads();
    if constexpr (FN == 0) {
        // (16 bytes per lane: registers 4g .. 4g+3 of a tile are four consecutive features of the lane's receiver -- element by
        // element every lane of a store instruction wrote 4 bytes into a 128-byte line of its own, 6,144 line requests per
        // workgroup; the LDS reads stay lane-consecutive)
        float* out = p.agg + ((size_t)sc * p.B + b) * p.N * H3;
        for (int u = tid; u < T3 * 4 * 64; u += 512) {
            const int ln = u & 63, mg = u >> 6, m = mg >> 2, g = mg & 3;
            float4 v;
            float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = (m * 16 + 4 * g + t) * 64 + ln;
                vp[t] = (red[e] + red[e + T3 * 1024] + red[e + 2 * T3 * 1024] + red[e + 3 * T3 * 1024]) * p.agg_scale;
            }
            const int ii = rb * 32 + (ln & 31);
            if (ii < p.N) *reinterpret_cast<float4*>(out + (size_t)ii * H3 + 32 * m + 8 * g + 4 * (ln >> 5)) = v;
        }
    } else {
        // ---- the node network on these 32 receivers (edge_fwd2_impl.h: registers 8s .. 8s+7 of accumulator tile m ARE the B
        //      fragment of k-step 2m + s of fn's first layer).  Eight waves: wave w sums k-steps w and w + 8 (< 12) of the four
        //      slabs -- in the plain path's order, times agg_scale: the values that path writes -- and lays them down as hi/lo
        //      fragments beside the slabs; waves 4, 5 also stage the x columns (k-steps 12, 13).
        static_assert(F2_FN_BIAS + C2_BIAS * 4 <= F2_LDS_BYTES, "fn's buffers must fit the edge kernel's LDS");
        const MpgChain& c = *cp;
        const int m0 = b * p.N + rb * 32, nrows = min(32, p.N - rb * 32);
        if (p.SC > 1) {
            // ---- sender chunks: this workgroup holds ONE of the SC partial sums of its (jet, receiver block).  It publishes its
            //      slab, draws a ticket, and the workgroup that draws the last one adds the slabs up IN CHUNK ORDER (the same sum
            //      whoever is last), leaves the total in slab 0 (the backward's agg) and goes on to the epilogue; the others are done.
            // Hand-off as MI355X_MICROARCH.md prices it (inter-workgroup visibility, the sc1 row of its table): every byte of the slab
            // goes out by a 16-byte sc1 (write-through) store, every storing wave waits for its stores (vmcnt(0)) in front of the
            // workgroup's barrier, ONE lane adds to the (jet, receiver block)'s counter with an agent-scope atomic, the workgroup whose
            // add came last -- told by the value returned -- reads every slab by 16-byte sc1 loads (they bypass this CU's L1) behind a
            // barrier.  (The textbook form -- plain stores, agent-scope release, acquire -- made every workgroup write back its XCD's
            // whole L2, with 59 MB of freshly parked E2 in it: the N = 150 iteration went from 2.6 to 4.5 ms.)
            const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(p.agg, 0, p.SC * p.B * p.N * (H3 * 4), 0x00020000);
            constexpr int SC1 = 16;   // cache-policy bit of the buffer instructions: sc1
            const int slab0 = (sc * p.B + b) * p.N * (H3 * 4);
            for (int u = tid; u < T3 * 4 * 64; u += 512) {
                const int ln = u & 63, mg = u >> 6, m = mg >> 2, g = mg & 3;
                f2_u32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int e = (m * 16 + 4 * g + t) * 64 + ln;
                    v[t] = __builtin_bit_cast(uint32_t, (red[e] + red[e + T3 * 1024] + red[e + 2 * T3 * 1024] + red[e + 3 * T3 * 1024]) * p.agg_scale);
                }
                const int ii = rb * 32 + (ln & 31);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsl, ii < p.N ? slab0 + (ii * H3 + 32 * m + 8 * g + 4 * (ln >> 5)) * 4 : -1, 0, SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                unsigned int* tk = p.tickets + (b * RB + rb);
                const unsigned int old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool last = old == (unsigned int)(p.SC - 1);
                if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (zero again for the next launch)
                *lnv = last ? 1 : 0;   // (the list's count: dead behind the sender loop)
            }
            __syncthreads();
            if (*lnv == 0) return;
            for (int u = tid; u < T3 * 4 * 64; u += 512) {
                const int ln = u & 63, mg = u >> 6, m = mg >> 2, g = mg & 3;
                const int ii = rb * 32 + (ln & 31);
                const int off = ii < p.N ? (ii * H3 + 32 * m + 8 * g + 4 * (ln >> 5)) * 4 : -1;   // (out of range: zeros)
                f32x4 tot = {0.f, 0.f, 0.f, 0.f};
                for (int q = 0; q < p.SC; ++q)   // in chunk order: the same sum whichever workgroup arrived last
                    tot += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsl, off >= 0 ? (q * p.B + b) * p.N * (H3 * 4) + off : -1, 0, SC1));
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int e = (m * 16 + 4 * g + t) * 64 + ln;
                    red[e] = tot[t];
                    red[e + T3 * 1024] = 0.f; red[e + 2 * T3 * 1024] = 0.f; red[e + 3 * T3 * 1024] = 0.f;
                }
            }
            __syncthreads();
        }
        const float red_scale = p.SC > 1 ? 1.f : p.agg_scale;   // (the chunks' slabs carry agg_scale already)
        using I0 = std::integral_constant<int, 0>;
        auto stage = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t, const uint32_t, const float ascale) {
            first_tile(I0{});   // (every register of the sender loop is free: the wave's weight tile is on its way during the staging)
            bias_request();
            V* fb = reinterpret_cast<V*>(smem + F2_FN_FB0);
            const __amdgpu_buffer_rsrc_t ragg = __builtin_amdgcn_make_buffer_rsrc(p.agg, 0, p.agg != nullptr ? p.B * p.N * (H3 * 4) : 0, 0x00020000);
            const int rowoff = vi ? (b * p.N + i) * (H3 * 4) : -1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ks = w + 8 * q, mm = ks >> 1, s2 = ks & 1;
                if (ks < 2 * T3) {
                    float av8[8], v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int e = (mm * 16 + 8 * s2 + j) * 64 + lane;
                        const float sum = red[e] + red[e + T3 * 1024] + red[e + 2 * T3 * 1024] + red[e + 3 * T3 * 1024];
                        av8[j] = sum * red_scale;
                        v[j] = av8[j] * ascale;
                    }
#pragma unroll
                    for (int half = 0; half < 2; ++half)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            f2_u32x4{__builtin_bit_cast(uint32_t, av8[4 * half]), __builtin_bit_cast(uint32_t, av8[4 * half + 1]),
                                     __builtin_bit_cast(uint32_t, av8[4 * half + 2]), __builtin_bit_cast(uint32_t, av8[4 * half + 3])},
                            ragg, vi ? rowoff + (32 * mm + 16 * s2 + 8 * half + 4 * h) * 4 : -1, 0, 0);
                    V hi, lo;
                    split8(v, hi, lo);
                    fb[(ks * 2 + 0) * 64 + lane] = hi;
                    fb[(ks * 2 + 1) * 64 + lane] = lo;
                }
            }
            if (w == 4 || w == 5) {   // the x columns [192, K) of cat((agg, x)): k-steps 12 and 13
                const int ks = 8 + w, KX = c.L[0].K - H3;
                const float* xr = c.A2 + (size_t)(m0 + r) * c.lda2;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = 16 * (w - 4) + 8 * (j >> 2) + 4 * h + (j & 3);
                    v[j] = (vi && f < KX) ? xr[f] * ascale : 0.f;
                }
                V hi, lo;
                split8(v, hi, lo);
                fb[(ks * 2 + 0) * 64 + lane] = hi;
                fb[(ks * 2 + 1) * 64 + lane] = lo;
            }
            bias_store();
        };
        c2_body<true, 14, 16, 16, DROP, 0, 0, FN == 2, F1_NW>(c, m0, nrows, smem + F2_FN_FB0, smem, reinterpret_cast<float*>(smem + F2_FN_BIAS), stage);
        if (cp2->nlayers > 0) {
            // the next layer's a | c on the rows just written (this workgroup's own stores: ordered within the workgroup)
            const MpgChain& c2 = *cp2;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            auto stage2 = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t s_lo, const uint32_t s_hi, const float ascale) {
                c2_stage_rows<true, 2, 0, F1_NW>(c2, m0, nrows, smem, first_tile, bias_request, bias_store, s_lo, s_hi, ascale);
            };
            c2_body<true, 2, 0, 0, 0, 0, 0, false, F1_NW>(c2, m0, nrows, smem, smem + C2_FB, reinterpret_cast<float*>(smem + 2 * C2_FB), stage2);
        }
    }
#ifdef MPG_F1_STAMP
    if (lane == 0) {   // whole-kernel figures in the slots the per-section sums do not use: prologue, kernel length, 100 MHz ticks
        unsigned long long* o = f1_stamps + ((size_t)blockIdx.x * 8 + w) * 8;
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        o[4] = o[6] - f1_k0;                                    // prologue: kernel start -> loop start  (loop top sum dropped)
        o[6] = t1 - o[7];                                       // epilogue: loop end -> kernel end
        o[7] = ((t1 - f1_k0) << 20) | (__builtin_amdgcn_s_memrealtime() - f1_r0);   // kernel clk | realtime ticks
    }
#endif
    }
}

#ifndef MPG_F1_LT
#define MPG_F1_LT 0   // (tools/ubench/fwd_bench.hip: the harness builds the plain kernel in one LT form)
#endif
template <int DROP, bool SIGN, int NQ, int LT = MPG_F1_LT>
__global__ __launch_bounds__(512) void edge_fwd1_kernel(const MpgEdgeFwd p) { edge_fwd1_body<DROP, SIGN, NQ, 0, LT>(p); }

template <int DROP, bool SIGN, bool SL, int LT = 0>
__global__ __launch_bounds__(512) void edge_fwd1_fn_kernel(const MpgEdgeFwd p, const MpgChain c, const MpgChain c2) {
    edge_fwd1_body<DROP, SIGN, 0, SL ? 2 : 1, LT>(p, &c, &c2);
}

// Which product form a launch takes (MpgEdgeFwd.two_term): 0 = three 16-bit terms in both dense layers, 1 = layer 3 on two terms
// (LT = 1 above).  Anything else is not built into the library: the entry points answer -8.
inline bool f1_terms_ok(const MpgEdgeFwd* p) { return p->two_term == 0 || p->two_term == 1; }

template <int D, int NQ, int LT>
int f1_launch_lt(const MpgEdgeFwd* p, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB * p->SC), block(512);
    if (p->sign3 != nullptr) {
        MPG_ENSURE_LDS((edge_fwd1_kernel<D, true, NQ, LT>), F1_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd1_kernel<D, true, NQ, LT>), grid, block, F1_LDS_BYTES, st, *p);
    } else {
        MPG_ENSURE_LDS((edge_fwd1_kernel<D, false, NQ, LT>), F1_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd1_kernel<D, false, NQ, LT>), grid, block, F1_LDS_BYTES, st, *p);
    }
    return (int)hipGetLastError();
}
template <int D, int NQ = 0>
int f1_launch(const MpgEdgeFwd* p, hipStream_t st) {
#ifdef MPG_SINGLE_VARIANT   // (tools/ubench/fwd_bench.hip: one form, -DMPG_F1_LT=n)
    return f1_launch_lt<D, NQ, MPG_F1_LT>(p, st);
#else
    return p->two_term == 1 ? f1_launch_lt<D, NQ, 1>(p, st) : f1_launch_lt<D, NQ, 0>(p, st);
#endif
}

// the fused forward + node network of one dropout mode / SIGN (edge_fwd_fn_*.hip: one translation unit each)
template <int D, bool SIGN, int LT>
int f1_launch_fn_lt(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, bool sl, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB * p->SC), block(512);
    MpgChain none = {};   // nlayers = 0: no second chain
    if (c2 == nullptr) c2 = &none;
    if (sl) {
        MPG_ENSURE_LDS((edge_fwd1_fn_kernel<D, SIGN, true, LT>), F1_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd1_fn_kernel<D, SIGN, true, LT>), grid, block, F1_LDS_BYTES, st, *p, *c, *c2);
    } else {
        MPG_ENSURE_LDS((edge_fwd1_fn_kernel<D, SIGN, false, LT>), F1_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd1_fn_kernel<D, SIGN, false, LT>), grid, block, F1_LDS_BYTES, st, *p, *c, *c2);
    }
    return (int)hipGetLastError();
}
template <int D, bool SIGN>
int f1_launch_fn(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, bool sl, hipStream_t st) {
    return p->two_term == 1 ? f1_launch_fn_lt<D, SIGN, 1>(p, c, c2, sl, st) : f1_launch_fn_lt<D, SIGN, 0>(p, c, c2, sl, st);
}

}  // namespace
