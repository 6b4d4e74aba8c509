// Fused fully-connected message-passing edge network, forward (MPLayer's fe + mask + aggregation).
//
// Replaces, for the default MPLayer configuration, the reference's
//   _getA_fully_connected  mpgan/model.py:284-317  (x.repeat / cat -> [B*N*N, 2F] edge tensor)
//   self.fe(A)             mpgan/model.py:256 -> LinearNet.forward :70-85 (3x Linear+LeakyReLU+Dropout)
//   A * mask ; sum/mean    mpgan/model.py:257-267
// The [B,N,N,*] edge activations never exist in memory.
//
// Layout ("chain" layout, see common.h): every activation tile is [features x receivers]: the receiver i sits on the
// MFMA column (= lane & 31), features sit in accumulator registers.  A layer's accumulator tile, converted to fp16 hi/lo,
// IS the B operand of the next layer's MFMA (no LDS round trip, no lane movement), weights are the A operand, pre-packed by
// pack_weights() into per-lane fragment images.  The sum over senders is a per-lane register accumulation over the sender
// loop, and layer 1 is the exact factorisation  W1 [x_i ; x_j] = a_i + c_j  (SURVEY.md A.3).
//
// A workgroup owns one (jet, 32 receivers, sender chunk); its four waves split the chunk's senders and are independent of
// each other between the prologue and the final reduction of agg.  What bounds a wave is operand delivery: every MFMA
// needs a 1 KiB weight fragment, and a CU moves 128 B/clk from LDS for its four SIMDs -- exactly one fragment per MFMA
// and wave, with nothing to spare.  So a wave walks its senders in PAIRS (as the data-gradient kernel does,
// edge_bwd2_impl.h): every weight fragment -- W3 hi+lo from LDS, W2 hi+lo streamed from L2 -- feeds the MFMAs of two
// senders.
//   layer 2  Z2 = W2 E1 + b2   k-outer (6 k-steps x 5 tiles x 2 senders): the e1 fragment of k-step k+1 is built behind
//                              the MFMAs of k-step k; all five accumulator tiles of both senders are live
//   E2       = drop(lrelu(Z2)) as hi/lo fragments (10 per sender)
//   layer 3  Z3 = W3 E2 + b3   tile by tile (6 tiles x 10 k-steps x 2 senders); the epilogue of a tile -- LeakyReLU,
//                              dropout, sign bit for the backward, m_j-weighted sum into agg -- sits behind the MFMAs of
//                              the next one
// Arithmetic per accumulator element: bias first, then per k-step lo*hi, hi*lo, hi*hi, k ascending -- the order the
// backward's recomputation of layer 2 repeats bit for bit.
#pragma once
#include "edge_common.h"
#include "chain2_impl.h"

#ifndef MPG_EXP
#define MPG_EXP 0  // experiment bits (tools/ubench/fwd_bench.hip): 1 streamed fragments all from k-step 0 (L1 hits)
#endif

namespace {

// LDS plan: W3 hi|lo (fp16) | a tile of the 32 receivers | b2 | b3 | per-wave rows of c for the two senders in flight |
// list of the chunk's senders.  W2 streams from L2.
constexpr int F2_W_BYTES = 2 * NF3 * 1024;       // 122,880
constexpr int F2_A_BYTES = T1 * 4 * 64 * 16;     //  12,288
constexpr int F2_B_BYTES = (H2 + H3) * 4;        //   1,408
constexpr int F2_C_BYTES = 4 * 2 * H1 * 4;       //   3,072
constexpr int F2_LIST_MAX = 160;                 // senders per chunk (uint16 entries + count: 384 B)
constexpr int F2_Q_OFF = F2_W_BYTES + F2_A_BYTES + F2_B_BYTES + F2_C_BYTES + 384;   // edge-scalar columns wq [3][96] (NQ > 0)
// ... and the listed senders' MASK ENTRIES, in list order (fp32): the sender loop takes m_j from here.  Loaded from memory at the
// top of a sender's round it cost a full drain of the wave's vector-memory queue -- vmcnt counts the parking stores of the round
// before too, and they complete in order -- 1.3-1.9k clk of a round's 20-27k (tools/ubench/fwd_bench.hip, -DMPG_F1_STAMP).
constexpr int F2_MK_OFF = F2_Q_OFF + MPG_EDGE_SCALARS * H1 * 4;
constexpr int F2_LDS_BYTES = F2_MK_OFF + F2_LIST_MAX * 4;
static_assert(F2_LDS_BYTES <= 163840, "LDS plan exceeds 160 KiB");
static_assert(4 * T3 * 16 * 64 * 4 <= F2_W_BYTES, "the final reduction reuses the weight area");

typedef unsigned int f2_u32x4 __attribute__((ext_vector_type(4)));

template <int NU, int NS, int SL, typename F>
MPG_DEV void f2_slot(F&& unit) {   // units u of [0, NU) that fall into slot SL of NS
    if constexpr (SL >= 0 && SL < NS) {
        constexpr int u0 = (SL * NU) / NS, u1 = ((SL + 1) * NU) / NS;
        static_for<u0, u1>(unit);
    }
}

MPG_DEV f32x16 f2_mma(const f16x8 a, const f16x8 b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// DROP: 0 off, 1 byte-threshold dropout, 2 one-bit (p = 1/2) dropout.  SIGN: also write the per-lane sign words of Z3 the
// backward needs (a run-time test here would put a branch after every accumulator register).
// NQ: edge scalars per edge (0, or MPG_EDGE_SCALARS: Z1 = a_i + c_j + sum_q es(i, j, q) wq[q])
// FN: 0 = agg goes to memory; 1 / 2 = the node network fn (mpgan/model.py:268-279: cat((agg, x)) -> three layers) runs as
// this workgroup's EPILOGUE on the 32 receivers it has just aggregated -- their agg rows never leave the CU as operands
// (agg itself is still written when a backward will need it) -- with chain2's schedule (chain2_impl.h; 2: the last layer's
// rows are not whole 16-byte groups).  Needs the whole jet in one workgroup (SC = 1).
constexpr int F2_RED_BYTES = 4 * T3 * 16 * 64 * 4;      // the four waves' partial sums: 98,304
constexpr int F2_FN_FB0 = F2_RED_BYTES;                 // fn's input fragments are laid down beside them ...
constexpr int F2_FN_BIAS = F2_FN_FB0 + C2_FB;           // ... its second fragment buffer over them, once they are dead

// ``cp2`` (FN only; nlayers = 0: none): one more chain on the rows fn has just written -- the NEXT MPLayer's layer-1 node terms
// a | c = [W1a ; W1c] y + [b1 ; 0] (its mpg_chain call, with fn's output rows as input) -- so that the next layer starts with
// its edge launch.
template <int DROP, bool SIGN, int NQ, int FN>
MPG_DEV void edge_fwd_body(const MpgEdgeFwd& p, const MpgChain* const cp, const MpgChain* const cp2 = nullptr) {
    typedef f16x8 V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int RB = (p.N + 31) / 32;
    int bid = blockIdx.x;
    const int sc = bid % p.SC; bid /= p.SC;
    const int rb = bid % RB;
    const int b = p.order != nullptr ? p.order[bid / RB] : bid / RB;   // (heaviest jets first: mpg_jet_order)
    const int i = rb * 32 + r;
    const bool vi = i < p.N;
    const int JC = (p.N + p.SC - 1) / p.SC;
    const int jbeg = sc * JC, jend = min(p.N, jbeg + JC);
    const int ldac = p.ld_ac ? p.ld_ac : H1;

    const __amdgpu_buffer_rsrc_t r2 = img_rsrc(p.W2img, 2 * NF2);   // W2 hi | lo
    const int lane16 = lane * 16;
    const V* g3 = reinterpret_cast<const V*>(p.W3img);
    V* l3 = reinterpret_cast<V*>(smem);
    float4* la = reinterpret_cast<float4*>(smem + F2_W_BYTES);                 // [(q*2+s)*2+u][lane]
    float* lb2 = reinterpret_cast<float*>(smem + F2_W_BYTES + F2_A_BYTES);
    float* lb3 = lb2 + H2;
    float* lcw = lb3 + H3 + w * (2 * H1);                                      // this wave's two rows of c
    unsigned short* lst = reinterpret_cast<unsigned short*>(smem + F2_W_BYTES + F2_A_BYTES + F2_B_BYTES + F2_C_BYTES);
    int* lnv = reinterpret_cast<int*>(lst + 180);   // (behind the list's 360 bytes)
    float* lmk = reinterpret_cast<float*>(smem + F2_MK_OFF);

    // ---- prologue (whole workgroup): W3 and the receivers' layer-1 terms into LDS, biases in the accumulators' scales,
    //      the list of the chunk's senders (the unmasked ones: a zero-masked sender adds exactly 0)
    // Order of issue: the small loads first (biases, the receivers' rows of a: into registers), then W3's image by LDS-DMA;
    // the registers go to LDS once everything has landed -- ONE wait for the whole prologue instead of a round trip per piece
    // (two copy passes, biases, rows: four, each behind the one before).
    static_assert(H2 + H3 <= 2 * 256 && T1 * 4 * 64 == 3 * 256, "the prologue's register sets");
    float bv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = min(tid + 256 * u, H2 + H3 - 1);
        bv[u] = t < H2 ? p.b2[t] : p.b3[t - H2];
    }
    float4 av[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = tid + 256 * u;
        const int ln = t & 63, qsu = t >> 6, rr = ln & 31, hh = ln >> 5, ii = min(rb * 32 + rr, p.N - 1);
        av[u] = ld4(p.a + (size_t)(b * p.N + ii) * ldac + 8 * qsu + 4 * hh);
    }
    fill_lds_dma(l3, g3, 2 * NF3 * 1024, tid);
    if constexpr (NQ > 0)
        for (int t = tid; t < NQ * H1; t += 256) reinterpret_cast<float*>(smem + F2_Q_OFF)[t] = p.wq[t] * SC_A;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + 256 * u;
        if (t < H2 + H3) lb2[t] = bv[u] * (t < H2 ? SC_E2 : SC_E3);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = tid + 256 * u;
        const bool in = rb * 32 + (t & 31) < p.N;
        la[t] = in ? make_float4(av[u].x * SC_A, av[u].y * SC_A, av[u].z * SC_A, av[u].w * SC_A) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // A jet whose senders all fit the list (N <= F2_LIST_MAX) is listed WHOLE and the list cut into SC equal parts: chunks by
    // sender index are as uneven as the mask (150 particles, 117 unmasked, 3 chunks: 50 / 50 / 17 senders -- the launch lasts
    // as long as the fullest).  Which chunk a sender lands in is this kernel's business alone: the by-products it leaves
    // (sign words, parked E2) are indexed by sender, and the backward partitions its own list.
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
    // (the sign words of skipped senders are never written, and never read: the backward skips the same blocks)
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
    const uint32_t lbc = lds_base(smem, F2_W_BYTES + F2_A_BYTES + F2_B_BYTES + w * (2 * H1 * 4) + 16 * h);
    const uint32_t lbq = lds_base(smem, F2_Q_OFF + 16 * h);

    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(p.sign3, 0, SIGN ? p.B * RB * p.N * (T3 * 32 * 4) : 0, 0x00020000);
    // E2 parked for the backward (mpg_edge_bwd takes phi'(Z2) from its signs, mpg_edge_dw multiplies with it): the hi halves
    // of the fragments built below, as they are -- one 16-byte store per lane and fragment, no arithmetic
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(p.stageE2, 0, (SIGN && p.stageE2 != nullptr) ? p.B * RB * p.N * (NFR2 * 1024) : 0, 0x00020000);

    f32x16 agg[T3];
#pragma unroll
    for (int m = 0; m < T3; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) agg[m][k] = 0.f;

    // the rows of c of a pair are requested one pair ahead (vector memory operations complete in order: requested at its own
    // top a pair's loads would queue behind the sign-word stores of the pair before)
    float pc0[2], pc1[2];
    float pes[2][NQ > 0 ? NQ : 1];   // the lane's receiver's edge scalars with the pair's senders
    auto prefetch = [&](int pq2) {
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const int jn = __builtin_amdgcn_readfirstlane((int)lst[max(0, min(2 * pq2 + sd, nvalid - 1))]);
            const float* cj = p.c + (size_t)(b * p.N + jn) * ldac;
            pc0[sd] = cj[lane];
            pc1[sd] = cj[64 + (lane & 31)];
            if constexpr (NQ > 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) pes[sd][q] = p.es[((size_t)(b * p.N + jn) * NQ + q) * p.N + (vi ? i : 0)];
            }
        }
    };
    prefetch(w);
    for (int pq = w; 2 * pq < nvalid; pq += 4) {
        const bool has2 = 2 * pq + 1 < nvalid;
        int jj[2];
        jj[0] = __builtin_amdgcn_readfirstlane((int)lst[2 * pq]);
        jj[1] = has2 ? __builtin_amdgcn_readfirstlane((int)lst[2 * pq + 1]) : jj[0];
        float mjs[2];
        uint32_t erow[2];
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const float mj = lmk[min(2 * pq + sd, nvalid - 1)];
            mjs[sd] = (sd == 0 || has2) ? mj * p.dscale * (1.f / SC_E3) : 0.f;   // (the layer-3 output carries SC_E3)
            if (p.nbr != nullptr) {  // k-nearest-neighbour graph: sender j counts for this lane's receiver only if its bit is set
                const unsigned int wb = p.nbr[(size_t)(b * p.N + (vi ? i : 0)) * ((p.N + 31) >> 5) + (jj[sd] >> 5)];
                mjs[sd] = ((wb >> (jj[sd] & 31)) & 1u) ? mjs[sd] : 0.f;
            }
            erow[sd] = (uint32_t)((b * p.N + i) * p.N + jj[sd]);
            lcw[sd * H1 + lane] = pc0[sd] * SC_A;
            if (lane < H1 - 64) lcw[sd * H1 + 64 + lane] = pc1[sd] * SC_A;
        }
        float esv[2][NQ > 0 ? NQ : 1];
        if constexpr (NQ > 0) {
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                for (int q = 0; q < NQ; ++q) esv[sd][q] = pes[sd][q];
        }
        prefetch(pq + 4);

        // ---- layer 2: Z2 = W2' E1 + b2, k-outer; e1 = drop(lrelu(a_i + c_j)) built one k-step ahead
        f32x16 acc[T2][2];       // [tile][sender]
        {
            constexpr int KS = T1 * 2;
            V eh[2][2], el[2][2];    // [buffer][sender] e1 fragment (hi, lo) of a k-step
            V wh[2][T2], wl[2][T2];  // [buffer][tile] W2 fragments of a k-step
            float v1[2][8];
            PairSplit<V> ps1[2][4];
            f32x4 a4[2], c4[2][2];
            f32x4 q4[NQ > 0 ? NQ : 1][2];
            auto load_w = [&](auto kc) {
                MPG_CI(k, kc);
#pragma unroll
                for (int m = 0; m < T2; ++m) {
                    wh[k & 1][m] = img_frag<V>(r2, lane16, m * KS + ((MPG_EXP & 1) ? 0 : k));
                    wl[k & 1][m] = img_frag<V>(r2, lane16, NF2 + m * KS + ((MPG_EXP & 1) ? 0 : k));
                }
            };
            auto load_a = [&](auto kc) {
                MPG_CI(k, kc);
                a4[0] = lds_frag<f32x4>(lbla, (k * 2 + 0) * 1024);
                a4[1] = lds_frag<f32x4>(lbla, (k * 2 + 1) * 1024);
#pragma unroll
                for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                    for (int uh = 0; uh < 2; ++uh) c4[sd][uh] = lds_frag<f32x4>(lbc, (sd * H1 + 16 * k + 8 * uh) * 4);
                if constexpr (NQ > 0) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int uh = 0; uh < 2; ++uh) q4[q][uh] = lds_frag<f32x4>(lbq, (q * H1 + 16 * k + 8 * uh) * 4);
                }
            };
            // build units of the e1 fragment of k-step k = (q, s): per sender 8 element units + 4 pairs x 2 halves = 16
            auto buildA = [&](auto kc, auto uc) {
                MPG_CI(k, kc); MPG_CI(uu, uc);
                constexpr int sd = uu / 16, u = uu % 16;
                constexpr int q = k >> 1, s = k & 1;
                if constexpr (u < 8) {
                    constexpr int uh = u >> 2, t = u & 3;  // element 4 uh + t  <->  feature 32q + 16s + 8uh + 4h + t
                    float cc = c4[sd][uh][t] + a4[uh][t];
                    if constexpr (NQ > 0) {
#pragma unroll
                        for (int q = 0; q < NQ; ++q) cc = fmaf(esv[sd][q], q4[q][uh][t], cc);
                    }
                    const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow[sd], q, 4 * s + 2 * uh + h, h);
                    v1[sd][u] = drop_apply<DROP>(lrelu(cc, p.alpha), wd, 16 * s + 8 * uh + t, t, p.thr);
                } else {
                    constexpr int pr = (u - 8) >> 1;
                    if constexpr (((u - 8) & 1) == 0) ps1[sd][pr].first(v1[sd][2 * pr], v1[sd][2 * pr + 1]);
                    else ps1[sd][pr].second(v1[sd][2 * pr + 1], eh[k & 1][sd], el[k & 1][sd], 2 * pr);
                }
            };
            load_w(std::integral_constant<int, 0>{});
            load_a(std::integral_constant<int, 0>{});
            static_for<0, 32>([&](auto uc) { buildA(std::integral_constant<int, 0>{}, uc); });
            // accumulators start as the bias column
#pragma unroll
            for (int m = 0; m < T2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = ld4(lb2 + 32 * m + 8 * g + 4 * h);
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd) {
                        acc[m][sd][4 * g + 0] = b4.x; acc[m][sd][4 * g + 1] = b4.y;
                        acc[m][sd][4 * g + 2] = b4.z; acc[m][sd][4 * g + 3] = b4.w;
                    }
                }
            static_for<0, KS>([&](auto kc) {
                MPG_CI(k, kc);
                if constexpr (k + 1 < KS) {
                    load_w(std::integral_constant<int, k + 1>{});
                    load_a(std::integral_constant<int, k + 1>{});
                }
                const V bh0 = eh[k & 1][0], bl0 = el[k & 1][0], bh1 = eh[k & 1][1], bl1 = el[k & 1][1];
                static_for<0, T2>([&](auto mc) {
                    MPG_CI(m, mc);
                    auto slot = [&](auto slc) {
                        MPG_CI(SL, slc);
                        if constexpr (k + 1 < KS) f2_slot<32, 28, SL - 2>([&](auto uc) { buildA(std::integral_constant<int, k + 1>{}, uc); });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const V a_h = wh[k & 1][m], a_l = wl[k & 1][m];
                    acc[m][0] = f2_mma(a_l, bh0, acc[m][0]); slot(std::integral_constant<int, 6 * m + 0>{});
                    acc[m][1] = f2_mma(a_l, bh1, acc[m][1]); slot(std::integral_constant<int, 6 * m + 1>{});
                    acc[m][0] = f2_mma(a_h, bl0, acc[m][0]); slot(std::integral_constant<int, 6 * m + 2>{});
                    acc[m][1] = f2_mma(a_h, bl1, acc[m][1]); slot(std::integral_constant<int, 6 * m + 3>{});
                    acc[m][0] = f2_mma(a_h, bh0, acc[m][0]); slot(std::integral_constant<int, 6 * m + 4>{});
                    acc[m][1] = f2_mma(a_h, bh1, acc[m][1]); slot(std::integral_constant<int, 6 * m + 5>{});
                });
            });
        }

        // ---- E2 = drop(lrelu(Z2)) as B fragments (hi, lo): tile mm, k-step half s -> fragment 2 mm + s
        f2_u32x4 e2h[2][T2 * 2], e2l[2][T2 * 2];
        int stv[2], sts[2];   // the senders' parking blocks: per-lane offset (out of range for the idle half of an odd pair), block offset
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            stv[sd] = (sd == 0 || has2) ? lane16 : (int)0x7ffffff0;
            sts[sd] = ((b * RB + rb) * p.N + jj[sd]) * (NFR2 * 1024);
        }
        static_for<0, T2>([&](auto mc) {
            MPG_CI(mm, mc);
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                uint32_t wd2 = 0u;
                if constexpr (DROP == 2) wd2 = drop_tile_word<2>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow[sd], mm, 0, h);
                float x2[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int g = u >> 2, t = u & 3;
                    uint32_t wd = wd2;
                    if constexpr (DROP == 1) wd = drop_tile_word<1>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow[sd], mm, 2 * g + h, h);
                    x2[u] = drop_apply<DROP>(lrelu(acc[mm][sd][u], p.alpha), wd, 8 * g + t, t, p.thr);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    V hi, lo;
                    split8(x2 + 8 * s, hi, lo);
                    e2h[sd][2 * mm + s] = __builtin_bit_cast(f2_u32x4, hi);
                    e2l[sd][2 * mm + s] = __builtin_bit_cast(f2_u32x4, lo);
                    if constexpr (SIGN) __builtin_amdgcn_raw_buffer_store_b128(e2h[sd][2 * mm + s], rsE, stv[sd], sts[sd] + (2 * mm + s) * 1024, 0);
                }
            }
        });

        // ---- layer 3: Z3 = W3' E2 + b3 tile by tile; the epilogue of tile m - 1 (both senders: sign bit, LeakyReLU, dropout,
        //      m_j-weighted sum into agg) rides in the MFMA slots of tile m
        {
            constexpr int KS = T2 * 2;
            f32x16 a3[2][2];   // [tile parity][sender]
            uint32_t sgn[2][T3 / 2] = {{0u, 0u, 0u}, {0u, 0u, 0u}};
            auto bias_init = [&](auto mc) {
                MPG_CI(m, mc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = ld4(lb3 + 32 * m + 8 * g + 4 * h);
#pragma unroll
                    for (int sd = 0; sd < 2; ++sd) {
                        a3[m & 1][sd][4 * g + 0] = b4.x; a3[m & 1][sd][4 * g + 1] = b4.y;
                        a3[m & 1][sd][4 * g + 2] = b4.z; a3[m & 1][sd][4 * g + 3] = b4.w;
                    }
                }
            };
            // epilogue units of tile mm: uu = 16 sender + element
            auto epi3 = [&](auto mc, auto uc) {
                MPG_CI(mm, mc); MPG_CI(uu, uc);
                constexpr int sd = uu / 16, e = uu % 16, g = e >> 2, t = e & 3;
                const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow[sd], mm, 2 * g + h, h);
                const float z = a3[mm & 1][sd][e];
                // sign words: this lane shifts in the sign bit of each of its 96 Z3 registers in (tile, register) order ->
                // word tile >> 1, bit 31 - (16 (tile & 1) + register); one v_alignbit each
                if constexpr (SIGN) sgn[sd][mm >> 1] = __builtin_amdgcn_alignbit(sgn[sd][mm >> 1], __builtin_bit_cast(uint32_t, z), 31);
                const float x = drop_apply<DROP>(lrelu(z, p.alpha), wd, 8 * g + t, t, p.thr);
                agg[mm][e] += mjs[sd] * x;
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
                    const V bh0 = __builtin_bit_cast(V, e2h[0][k]), bl0 = __builtin_bit_cast(V, e2l[0][k]);
                    const V bh1 = __builtin_bit_cast(V, e2h[1][k]), bl1 = __builtin_bit_cast(V, e2l[1][k]);
                    auto slot = [&](auto slc) {
                        MPG_CI(SL, slc);   // 6 KS - 6 slots: the last k-step's are left to the next tile's bias load
                        if constexpr (m > 0) f2_slot<32, 6 * KS - 6, SL>([&](auto uc) { epi3(std::integral_constant<int, m - 1>{}, uc); });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const V a_h = ah[k & 1], a_l = al[k & 1];
                    a3[m & 1][0] = f2_mma(a_l, bh0, a3[m & 1][0]); slot(std::integral_constant<int, 6 * k + 0>{});
                    a3[m & 1][1] = f2_mma(a_l, bh1, a3[m & 1][1]); slot(std::integral_constant<int, 6 * k + 1>{});
                    a3[m & 1][0] = f2_mma(a_h, bl0, a3[m & 1][0]); slot(std::integral_constant<int, 6 * k + 2>{});
                    a3[m & 1][1] = f2_mma(a_h, bl1, a3[m & 1][1]); slot(std::integral_constant<int, 6 * k + 3>{});
                    a3[m & 1][0] = f2_mma(a_h, bh0, a3[m & 1][0]); slot(std::integral_constant<int, 6 * k + 4>{});
                    a3[m & 1][1] = f2_mma(a_h, bh1, a3[m & 1][1]); slot(std::integral_constant<int, 6 * k + 5>{});
                    if constexpr (k == KS - 1 && m + 1 < T3) {
                        bias_init(std::integral_constant<int, m + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
            static_for<0, 32>([&](auto uc) { epi3(std::integral_constant<int, T3 - 1>{}, uc); });
            if constexpr (SIGN) {
                // (buffer stores: the idle second half of an odd pair gets an offset beyond the buffer and the hardware
                // drops its stores -- no branch)
#pragma unroll
                for (int sd = 0; sd < 2; ++sd) {
                    const int voff = (sd == 0 || has2) ? lane * 4 : (int)0x7ffffff0;
                    const int soff = ((b * RB + rb) * p.N + jj[sd]) * (T3 * 32 * 4);
#pragma unroll
                    for (int q = 0; q < T3 / 2; ++q) __builtin_amdgcn_raw_buffer_store_b32(sgn[sd][q], rsS, voff, soff + q * 256, 0);
                }
            }
        }
    }

    // ---- reduce the four waves' partial sums through LDS and write agg[b, i, :]
    __syncthreads();  // everyone is done with the weight copy
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int m = 0; m < T3; ++m)
#pragma unroll
        for (int k = 0; k < 16; ++k) red[((w * T3 + m) * 16 + k) * 64 + lane] = agg[m][k];
    __syncthreads();
    if constexpr (FN == 0) {
        float* out = p.agg + ((size_t)sc * p.B + b) * p.N * H3;
        for (int e = tid; e < T3 * 16 * 64; e += 256) {
            const int ln = e & 63, k = (e >> 6) & 15, m = e >> 10;
            const float sum = red[e] + red[e + T3 * 1024] + red[e + 2 * T3 * 1024] + red[e + 3 * T3 * 1024];
            const int ii = rb * 32 + (ln & 31);
            const int f = 32 * m + 8 * (k >> 2) + 4 * (ln >> 5) + (k & 3);
            if (ii < p.N) out[(size_t)ii * H3 + f] = sum * p.agg_scale;
        }
    } else {
        // ---- the node network on these 32 receivers.  Registers 8s .. 8s+7 of accumulator tile m ARE the B fragment of
        //      k-step 2m + s of fn's first layer (same chain layout): wave w sums k-steps w, w + 4, w + 8 of the four waves'
        //      partials in the order of the plain path, times agg_scale -- the very fp32 values that path writes and
        //      chain2 reads back, so both routes give the same bits -- and lays them down as hi/lo fragments beside the
        //      partials; the x columns (k-steps 12, 13) come from memory.
        static_assert(F2_FN_BIAS + C2_BIAS * 4 <= F2_LDS_BYTES, "fn's buffers must fit the edge kernel's LDS");
        const MpgChain& c = *cp;
        const int m0 = b * p.N + rb * 32, nrows = min(32, p.N - rb * 32);
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        auto stage = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t, const uint32_t, const float ascale) {
            first_tile(I0{});   // (every register of the sender loop is free: the first weight tile is on its way during the staging)
            bias_request();
            V* fb = reinterpret_cast<V*>(smem + F2_FN_FB0);
            const __amdgpu_buffer_rsrc_t ragg = __builtin_amdgcn_make_buffer_rsrc(p.agg, 0, p.agg != nullptr ? p.B * p.N * (H3 * 4) : 0, 0x00020000);
            const int rowoff = vi ? (b * p.N + i) * (H3 * 4) : -1;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ks = w + 4 * q, mm = ks >> 1, s2 = ks & 1;
                float av[8], v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = (mm * 16 + 8 * s2 + j) * 64 + lane;
                    const float sum = red[e] + red[e + T3 * 1024] + red[e + 2 * T3 * 1024] + red[e + 3 * T3 * 1024];
                    av[j] = sum * p.agg_scale;
                    v[j] = av[j] * ascale;
                }
                // features 32 mm + 16 s2 + 4 h + {0..3} and + 8: two 16-byte stores of the receiver's agg row (kept for the
                // backward: fn.net.0's weight gradient; a NULL agg has an empty descriptor and the stores are dropped)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        f2_u32x4{__builtin_bit_cast(uint32_t, av[4 * half]), __builtin_bit_cast(uint32_t, av[4 * half + 1]),
                                 __builtin_bit_cast(uint32_t, av[4 * half + 2]), __builtin_bit_cast(uint32_t, av[4 * half + 3])},
                        ragg, vi ? rowoff + (32 * mm + 16 * s2 + 8 * half + 4 * h) * 4 : -1, 0, 0);
                V hi, lo;
                split8(v, hi, lo);
                fb[(ks * 2 + 0) * 64 + lane] = hi;
                fb[(ks * 2 + 1) * 64 + lane] = lo;
            }
            if (w < 2) {   // the x columns [192, K) of cat((agg, x)): k-steps 12 and 13
                const int ks = 12 + w, KX = c.L[0].K - H3;
                const float* xr = c.A2 + (size_t)(m0 + r) * c.lda2;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = 16 * w + 8 * (j >> 2) + 4 * h + (j & 3);
                    v[j] = (vi && f < KX) ? xr[f] * ascale : 0.f;
                }
                V hi, lo;
                split8(v, hi, lo);
                fb[(ks * 2 + 0) * 64 + lane] = hi;
                fb[(ks * 2 + 1) * 64 + lane] = lo;
            }
            first_tile(I1{});
            bias_store();
        };
        c2_body<true, 14, 16, 16, DROP, 0, 0, FN == 2>(c, m0, nrows, smem + F2_FN_FB0, smem, reinterpret_cast<float*>(smem + F2_FN_BIAS), stage);
        if (cp2->nlayers > 0) {
            // the next layer's a | c on the rows just written (every buffer of fn is dead behind its last barrier; the rows
            // are this workgroup's own stores: ordered within the workgroup)
            const MpgChain& c2 = *cp2;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            auto stage2 = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t s_lo, const uint32_t s_hi, const float ascale) {
                c2_stage_rows<true, 2, 0>(c2, m0, nrows, smem, first_tile, bias_request, bias_store, s_lo, s_hi, ascale);
            };
            c2_body<true, 2, 0, 0, 0, 0, 0, false>(c2, m0, nrows, smem, smem + C2_FB, reinterpret_cast<float*>(smem + 2 * C2_FB), stage2);
        }
    }
}

template <int DROP, bool SIGN, int NQ>
__global__ __launch_bounds__(256, 1) void edge_fwd_kernel(const MpgEdgeFwd p) { edge_fwd_body<DROP, SIGN, NQ, 0>(p, nullptr); }

template <int DROP, bool SIGN, bool SL>
__global__ __launch_bounds__(256, 1) void edge_fwd_fn_kernel(const MpgEdgeFwd p, const MpgChain c, const MpgChain c2) {
    edge_fwd_body<DROP, SIGN, 0, SL ? 2 : 1>(p, &c, &c2);
}

// the SIGN pair of one dropout mode and edge-scalar count (the combinations compile as separate translation units: edge.hip,
// edge_fwd_d1.hip, edge_fwd_d2.hip; edge_fwd_q{0,1,2}.hip with the edge scalars)
template <int D, int NQ = 0>
int f2_launch(const MpgEdgeFwd* p, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB * p->SC), block(256);
    if (p->sign3 != nullptr) {
        MPG_ENSURE_LDS((edge_fwd_kernel<D, true, NQ>), F2_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd_kernel<D, true, NQ>), grid, block, F2_LDS_BYTES, st, *p);
    } else {
        MPG_ENSURE_LDS((edge_fwd_kernel<D, false, NQ>), F2_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd_kernel<D, false, NQ>), grid, block, F2_LDS_BYTES, st, *p);
    }
    return (int)hipGetLastError();
}

// the fused forward + node network of one dropout mode / SIGN (edge_fwd_fn_*.hip: one translation unit each)
template <int D, bool SIGN>
int f2_launch_fn(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, bool sl, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB), block(256);
    MpgChain none = {};   // nlayers = 0: no second chain
    if (c2 == nullptr) c2 = &none;
    if (sl) {
        MPG_ENSURE_LDS((edge_fwd_fn_kernel<D, SIGN, true>), F2_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd_fn_kernel<D, SIGN, true>), grid, block, F2_LDS_BYTES, st, *p, *c, *c2);
    } else {
        MPG_ENSURE_LDS((edge_fwd_fn_kernel<D, SIGN, false>), F2_LDS_BYTES);
        hipLaunchKernelGGL((edge_fwd_fn_kernel<D, SIGN, false>), grid, block, F2_LDS_BYTES, st, *p, *c, *c2);
    }
    return (int)hipGetLastError();
}

}  // namespace
