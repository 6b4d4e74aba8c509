// Backward of the fused edge network, data-gradient path -- the EIGHT-WAVE form: two waves per SIMD, one sender per wave.
//
// Same function, memory formats and per-element arithmetic as edge_bwd2_impl.h (see its head for the products, the gradient
// units and the dither); what changes is the shape of the work.  The four-wave kernel walks its senders in pairs with one
// wave per SIMD and is bound by VALU ISSUE (~3,500 vector instructions against 360 MFMAs per pair, 1 wave issuing one
// vector instruction per ~5-7 clk): it hides the operand builds (dZ3, dZ2) in the MFMA slots of a k-outer schedule and still
// spends half of a pair on instructions the matrix pipe waits for.  Here a workgroup has eight waves of <= 256 registers, a wave
// owns ONE sender per round, and its phases are plain:
//   B     dE2 = W3'^T dZ3          120 MFMAs, the dZ3 fragment of k-step k + 1 built in the slots of k-step k (as before)
//   gate  dZ2 = dE2 * keep2 * phi'(Z2) -> ten fp16 fragments (and parked for mpg_edge_dw)        VALU only
//   C     dE1 = W2'^T dZ2           60 MFMAs, W2^T streamed through a three-k-step ring
//   dZ1   gate, da_i += dZ1, dc_j = sum_i dZ1 (DPP halving reduction)                            VALU only
// A wave's VALU-only phases run beside its SIMD partner's MFMA phases (the second wave issues beside the first at the same
// rate: tools/ubench/valu_rate2.hip), and with 256 registers the accumulators are VGPRs the gates read in place: no
// v_accvgpr_read / _write (a tenth of the four-wave kernel's instructions).  Peak registers: phase B  dacc 48 + accB 80 + the
// first E2 fragments 12 + build 24; phase C  dacc 48 + dZ2 fragments 40 + accC 48 + ring 72.
// Plain form only (no edge scalars, no epilogue chains): mpg_edge_bwd; mpg_edge_bwd_fn answers MPG_FN_NA in this mode and the
// caller launches the chains itself.
#pragma once
#include "edge_bwd2_impl.h"
#include <stdlib.h>

#ifdef MPG_B1_STAMP   // diagnostic build (tools/ubench/bwd2_bench.hip): s_memtime per section of a wave's senders, summed per wave
__device__ unsigned long long b1_stamps[4096 * 8 * 10];
#define B1_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); b1_acc[i] += t_ - b1_t; b1_t = t_; }
#else
#define B1_STAMP(i)
#endif

namespace {

constexpr int B1_NW = 8;
static_assert(B1_NW * H1 * 4 == B2_C_BYTES, "one row of c per wave in the four-wave kernel's two-rows-per-wave area");
static_assert(T3 * 4 * 64 == 3 * 512 && T1 * 4 * 64 == 512 + 256, "the prologue's register sets");
static_assert(B1_NW * T1 * 16 * 64 * 4 <= B2_W_BYTES, "the final reduction reuses the weight area");

// EPI / cdxp / cnxp: the epilogue chains of edge_bwd_body (the layer's dx chain and the lower layer's node-network input-gradient
// chain on this workgroup's own jet), run by c2_body's eight-wave form
template <int DROP, bool NEEDW, int EPI>
MPG_DEV void edge_bwd1_body(const MpgEdgeBwd& p, const MpgChain* const cdxp = nullptr, const MpgChain* const cnxp = nullptr) {
    typedef f16x8 V;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MPG_B1_STAMP
    const unsigned long long b1_k0 = __builtin_amdgcn_s_memtime(), b1_r0 = __builtin_amdgcn_s_memrealtime();
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
    const int JC = (p.N + p.SC - 1) / p.SC;
    const int jbeg = sc * JC, jend = min(p.N, jbeg + JC);
    const int ldac = p.ld_ac ? p.ld_ac : H1;

    const __amdgpu_buffer_rsrc_t r2t = img_rsrc(p.W2Timg, 2 * NF2T);  // W2^T hi | lo (fp16, operand scale SC_W2)
    const int lane16 = lane * 16;
    const f16x8* t3g = reinterpret_cast<const f16x8*>(p.W3Timg);
    f16x8* l3t = reinterpret_cast<f16x8*>(smem);
    float4* ldg = reinterpret_cast<float4*>(smem + B2_W_BYTES);                // [(m*4+g)][lane]
    float4* la = reinterpret_cast<float4*>(smem + B2_W_BYTES + B2_DG_BYTES);   // [(q*2+s)*2+u][lane]
    float* lmk = reinterpret_cast<float*>(smem + B2_Q_OFF);                    // the listed senders' mask entries (edge_bwd2_impl.h)
    float* lcw = reinterpret_cast<float*>(smem + B2_Q_OFF + B2_B2_BYTES) + w * H1;   // this wave's row of c
    unsigned short* lst = reinterpret_cast<unsigned short*>(smem + B2_Q_OFF + B2_B2_BYTES + B2_C_BYTES);
    int* lnv = reinterpret_cast<int*>(lst + 180);
    float* lmx = reinterpret_cast<float*>(lst + 180) - 8;                      // wave maxima of |dagg|: 32 bytes behind the list's 320
    static_assert(2 * B2_LIST_MAX + 32 <= 360, "the eight maxima sit between the list and its count");

    // ---- prologue (the four-wave kernel's, on 512 threads)
    float4 dv[3], av[2];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = tid + 512 * u;
        const int ln = t & 63, mg = t >> 6, rr = ln & 31, hh = ln >> 5, ii = min(rb * 32 + rr, p.N - 1);
        const float* di = p.dagg + (size_t)(b * p.N + ii) * p.ld_dagg + 32 * (mg >> 2) + 8 * (mg & 3) + 4 * hh;
        dv[u] = make_float4(di[0], di[1], di[2], di[3]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = min(tid + 512 * u, T1 * 4 * 64 - 1);
        const int ln = t & 63, qsu = t >> 6, rr = ln & 31, hh = ln >> 5, ii = min(rb * 32 + rr, p.N - 1);
        av[u] = ld4(p.a + (size_t)(b * p.N + ii) * ldac + 8 * qsu + 4 * hh);
    }
    for (int c = w; c < 2 * NF3T; c += B1_NW)   // W3^T's image by LDS-DMA: 1 KiB per wave-instruction
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(t3g) + c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(l3t) + c * 1024), 16, 0, 0);
    float amax = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = tid + 512 * u;
        const bool in = rb * 32 + (t & 31) < p.N;
        const float4 v4 = in ? make_float4(dv[u].x * p.agg_scale, dv[u].y * p.agg_scale, dv[u].z * p.agg_scale, dv[u].w * p.agg_scale)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        ldg[t] = v4;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v4.x), fabsf(v4.y))), fmaxf(fabsf(v4.z), fabsf(v4.w)));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) lmx[w] = amax;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = tid + 512 * u;
        const bool in = rb * 32 + (t & 31) < p.N;
        if (t < T1 * 4 * 64)
            la[t] = in ? make_float4(av[u].x * SC_A, av[u].y * SC_A, av[u].z * SC_A, av[u].w * SC_A) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool whole = p.N <= B2_LIST_MAX;
    const int lbeg = whole ? 0 : jbeg, lend = whole ? p.N : jend;
    if (w == 0) {
        int cnt = 0;
        for (int j0 = lbeg; j0 < lend; j0 += 64) {
            const int j = j0 + lane;
            const float mv = (j < lend && p.mask != nullptr) ? p.mask[b * p.N + j] : 1.f;
            const bool ok = j < lend && mv != 0.f;
            const unsigned long long bits = __ballot(ok);
            const int pos = cnt + __popcll(bits & ((1ull << lane) - 1ull));
            if (ok) { lst[pos] = (unsigned short)j; lmk[pos] = mv; }
            cnt += __popcll(bits);
        }
        if (lane == 0) *lnv = cnt;
    }
    // masked senders: every gradient through their edges is exactly zero (mpg_edge_dw skips those blocks too)
    if (p.mask != nullptr)
        for (int t = tid; t < (jend - jbeg) * (H1 / 4); t += 512) {
            const int j = jbeg + t / (H1 / 4);
            if (p.mask[b * p.N + j] == 0.f)
                reinterpret_cast<float4*>(p.dc + ((size_t)rb * p.B * p.N + (size_t)(b * p.N + j)) * H1)[t % (H1 / 4)] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    __syncthreads();
    int nvalid = __builtin_amdgcn_readfirstlane(*lnv);
    if (whole) {
        const int per = (nvalid + p.SC - 1) / p.SC, l0 = min(nvalid, sc * per);
        lst += l0;
        lmk += l0;
        nvalid = min(per, nvalid - l0);
    }
    float mx8 = 0.f;
#pragma unroll
    for (int q = 0; q < B1_NW; ++q) mx8 = fmaxf(mx8, lmx[q]);
    const int gexp = __builtin_amdgcn_readfirstlane(grad_unit_exp(mx8 * p.dscale));
    const float gunit = __builtin_bit_cast(float, (uint32_t)(gexp + 127) << 23);            // 2^e
    if (NEEDW && tid == 0) p.gexp[b * RB + rb] = gexp;

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    const uint32_t lb3t = lds_base(smem, lane16);                 // W3^T hi fragments; lo at + NF3T KiB
    const uint32_t lbdg = lds_base(smem, B2_W_BYTES + lane16);    // dagg tile
    const uint32_t lbla = lds_base(smem, B2_W_BYTES + B2_DG_BYTES + lane16);  // a tile
    const uint32_t lbc = lds_base(smem, B2_Q_OFF + B2_B2_BYTES + w * (H1 * 4) + 16 * h);  // this wave's row of c
    const int nblk = p.B * RB * p.N;
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageE2), 0, nblk * (NFR2 * 1024), 0x00020000);   // (read)
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(p.stageZ2, 0, NEEDW ? nblk * (NFR2 * 1024) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(p.sign3), 0, nblk * (T3 * 32 * 4), 0x00020000);

    const float sone2 = 1.f / SC_W3;
    const float sone1 = __builtin_bit_cast(float, (uint32_t)(127 - gexp - 4) << 23);   // 2^-e / SC_W2 (SC_W2 = 2^4)
    static_assert(SC_W2 == 16.f, "sone1 assumes SC_W2 = 2^4");
    float dacc[T1][16];
#pragma unroll
    for (int q = 0; q < T1; ++q)
#pragma unroll
        for (int k = 0; k < 16; ++k) dacc[q][k] = 0.f;

    // what a sender needs from memory -- its sign words, its row of c, its mask entry -- is requested one round ahead
    uint32_t psw[T3 / 2];
    float pc0, pc1;
    auto prefetch = [&](int sn) {
        const int jn = __builtin_amdgcn_readfirstlane((int)lst[max(0, min(sn, nvalid - 1))]);
        const int blk = (b * RB + rb) * p.N + jn;
#pragma unroll
        for (int q = 0; q < T3 / 2; ++q) psw[q] = __builtin_amdgcn_raw_buffer_load_b32(rsS, lane * 4, blk * (T3 * 32 * 4) + q * 256, 0);
        const float* cj = p.c + (size_t)(b * p.N + jn) * ldac;
        pc0 = cj[lane];
        pc1 = cj[64 + (lane & 31)];
    };
    prefetch(w);
#ifdef MPG_B1_STAMP
    unsigned long long b1_acc[6] = {}, b1_t = __builtin_amdgcn_s_memtime();
    const unsigned long long b1_l0 = b1_t;
#endif
#ifdef MPG_B1_STAGGER   // experiment: waves 4..7 (the SIMD partners of waves 0..3) start this many s_sleep(16) (~1k clk each) late
    if (w >= 4)
        for (int t = 0; t < MPG_B1_STAGGER; ++t) __builtin_amdgcn_s_sleep(16);
#endif
    for (int s = w; s < nvalid; s += B1_NW) {
        B1_STAMP(5)
        const int jj = __builtin_amdgcn_readfirstlane((int)lst[s]);
        int oln0 = lane;
        asm volatile("" : "+v"(oln0));
        const uint32_t erow = (uint32_t)((b * p.N + rb * 32 + (oln0 & 31)) * p.N) + (uint32_t)jj;
        float in_set = 1.f;
        if (p.nbr != nullptr) {  // k-nearest-neighbour graph: the edge (i, j) exists only if j's bit is set in i's row
            const unsigned int wb = p.nbr[(size_t)(b * p.N + (i < p.N ? i : 0)) * ((p.N + 31) >> 5) + (jj >> 5)];
            in_set = ((wb >> (jj & 31)) & 1u) ? 1.f : 0.f;
        }
        const int blk = (b * RB + rb) * p.N + jj;
        const float dth = dither_of((uint32_t)blk);   // this sender's unit within the workgroup's
        const float mj = lmk[s];   // (from LDS: a load at the top of the round would wait for every store of the round before)
        const float mjs = mj * p.dscale * gunit;
        const float cpos = mjs * in_set * dth, cneg = mjs * p.alpha * in_set * dth;
        const int stsc = blk * (NFR2 * 1024);
        uint32_t sw[T3 / 2];
#pragma unroll
        for (int q = 0; q < T3 / 2; ++q) sw[q] = psw[q];
        lcw[lane] = pc0 * SC_A;
        if (lane < H1 - 64) lcw[64 + lane] = pc1 * SC_A;
        prefetch(s + B1_NW);

        // the parked E2 fragments (their SIGN is phi'(Z2)) come from HBM: all ten are requested here, a whole phase B ahead (a
        // ring of three, refilled as the gate consumed them, left the gate waiting for memory: 5.2k clk for 400 instructions)
#ifndef MPG_B1_E2D
#define MPG_B1_E2D 10
#endif
#ifndef MPG_B1_WD
#define MPG_B1_WD 3
#endif
        constexpr int E2D = MPG_B1_E2D;
        b2_u32x4 e2g[E2D];
        auto load_e2 = [&](auto kc) {
            MPG_CI(k, kc);
            e2g[k % E2D] = __builtin_amdgcn_raw_buffer_load_b128(rsE, lane16, stsc + k * 1024, 0);
        };
        static_for<0, E2D>([&](auto kc) { load_e2(kc); });

        B1_STAMP(0)
        // ---- phase B: dE2 = W3'^T dZ3, k-outer, two fp16 terms; dZ3 = dagg * slope(sign word) * keep3 built one k-step ahead
        f32x16 accB[T2];
        {
            constexpr int KS = T3 * 2;  // 12 k-steps of 16 features of layer 3
            b2_u32x4 zz[2];               // [buffer] dZ3 fragment of a k-step
            float v3[8];
            f32x4 dg[2];                  // dagg of the k-step being built
            uint32_t wd3 = 0u;
            V ah[MPG_B1_WD], al[MPG_B1_WD];
            auto load_dg = [&](auto kc) {
                MPG_CI(k, kc);
                constexpr int m3 = k >> 1, s2 = k & 1;
                dg[0] = lds_frag<f32x4>(lbdg, ((m3 * 4 + 2 * s2) * 64) * 16);
                dg[1] = lds_frag<f32x4>(lbdg, ((m3 * 4 + 2 * s2 + 1) * 64) * 16);
            };
            // build units of the dZ3 fragment of k-step k: 8 element units + 4 pair conversions = 12
            auto buildB = [&](auto kc, auto uc) {
                MPG_CI(k, kc); MPG_CI(u, uc);
                constexpr int m3 = k >> 1, s2 = k & 1;
                if constexpr (u < 8) {
                    constexpr int r16 = 8 * s2 + u, g = r16 >> 2, t = r16 & 3;
                    if constexpr (DROP == 2) { if constexpr (u == 0 && s2 == 0) wd3 = drop_tile_word<2>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, m3, 0, h); }
                    uint32_t wd = wd3;
                    if constexpr (DROP == 1) wd = drop_tile_word<1>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, m3, 2 * g + h, h);
                    const float dd = dg[u >> 2][t];
                    const float sel = dd * sel_by_bit<31 - (16 * (m3 & 1) + r16)>(sw[m3 >> 1], cneg, cpos);
                    v3[u] = drop_apply<DROP>(sel, wd, 8 * g + t, t, p.thr);
                } else {
                    constexpr int pr = u - 8;
                    zz[k & 1][pr] = cvt_pk_f16(v3[2 * pr], v3[2 * pr + 1]);
                }
            };
            load_dg(std::integral_constant<int, 0>{});
            static_for<0, 12>([&](auto uc) { buildB(std::integral_constant<int, 0>{}, uc); });
            static_for<0, KS>([&](auto kc) {
                MPG_CI(k, kc);
                if constexpr (k + 1 < KS) load_dg(std::integral_constant<int, k + 1>{});
                const V b0 = __builtin_bit_cast(V, zz[k & 1]);
                // W3^T fragments from LDS through a ring of B1_WD tiles: tile m + B1_WD - 1 is requested in front of tile m's MFMAs
                // (one tile ahead -- two MFMAs, 64 clk -- is less than an LDS read takes: every tile then waits for its operands)
                constexpr int WD = MPG_B1_WD;
                static_for<0, WD - 1>([&](auto mc) {
                    MPG_CI(m, mc);
                    if constexpr (k == 0) {
                        ah[m] = lds_frag<V>(lb3t, (m * KS + k) * 1024);
                        al[m] = lds_frag<V>(lb3t, (NF3T + m * KS + k) * 1024);
                    }
                });
                static_for<0, T2>([&](auto mc) {
                    MPG_CI(m, mc);
                    {   // the ring runs on across k-steps: request (k, m + WD - 1), or the first tiles of k-step k + 1
                        constexpr int mn = m + WD - 1, kn = mn < T2 ? k : k + 1, mm = mn < T2 ? mn : mn - T2;
                        if constexpr (kn < KS) {
                            ah[(kn * T2 + mm) % WD] = lds_frag<V>(lb3t, (mm * KS + kn) * 1024);
                            al[(kn * T2 + mm) % WD] = lds_frag<V>(lb3t, (NF3T + mm * KS + kn) * 1024);
                        }
                    }
                    auto slot = [&](auto slc) {
                        MPG_CI(SL, slc);   // (the first slot leaves the LDS reads of the next k-step time to land)
                        if constexpr (k + 1 < KS && SL >= 1) run_slot<12, 9, SL - 1>([&](auto uc) { buildB(std::integral_constant<int, k + 1>{}, uc); });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const V a_h = ah[(k * T2 + m) % WD], a_l = al[(k * T2 + m) % WD];
                    if constexpr (k == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        accB[m] = mma(a_l, b0, z);
                    } else {
                        accB[m] = mma(a_l, b0, accB[m]);
                    }
                    slot(std::integral_constant<int, 2 * m + 0>{});
                    accB[m] = mma(a_h, b0, accB[m]); slot(std::integral_constant<int, 2 * m + 1>{});
                });
            });
        }

        B1_STAMP(1)
        // ---- gate: dZ2 = dE2 * keep2 * phi'(Z2) (still in the sender's unit) as ten fp16 fragments, parked for mpg_edge_dw as they
        //      are; W2^T's first k-steps are requested first (they arrive while the fragments are built)
        constexpr int KSC = T2 * 2;  // 10 k-steps of 16 features of layer 2
        V ah[3][T1], al[3][T1];       // ring of three k-steps of W2^T fragments (L2 is more than one k-step away)
        auto load_w = [&](auto kc) {
            MPG_CI(k, kc);
#pragma unroll
            for (int m = 0; m < T1; ++m) {
                ah[k % 3][m] = img_frag<V>(r2t, lane16, m * KSC + k);
                al[k % 3][m] = img_frag<V>(r2t, lane16, NF2T + m * KSC + k);
            }
        };
        b2_u32x4 z2f[KSC];
        {
            float valpha2 = p.alpha, vone2 = sone2;
            asm volatile("" : "+v"(valpha2), "+v"(vone2));   // (opaque copies made HERE: not hoisted out of the sender loop)
            valpha2 *= vone2;
            const int stoff = stsc + lane16;
            uint32_t wd2c = 0u;
            static_for<0, KSC>([&](auto kc) {
                MPG_CI(k, kc);
                constexpr int m2 = k >> 1, s2 = k & 1;
                float v2[8];
                static_for<0, 8>([&](auto uc) {
                    MPG_CI(u, uc);
                    constexpr int r16 = 8 * s2 + u, g = r16 >> 2, t = r16 & 3;
                    // sign of Z2 (register r16 of tile m2) = sign of element u of the parked E2 fragment 2 m2 + s2
                    float gt = sel_by_bit<16 * (u & 1) + 15>(e2g[k % E2D][u >> 1], valpha2, vone2);
                    if constexpr (DROP != 0) {
                        uint32_t wd;
                        if constexpr (DROP == 2) {
                            if constexpr (u == 0 && s2 == 0) wd2c = drop_tile_word<2>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow, m2, 0, h);
                            wd = wd2c;
                        } else wd = drop_tile_word<1>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow, m2, 2 * g + h, h);
                        gt = drop_apply<DROP>(gt, wd, 8 * g + t, t, p.thr);
                    }
                    v2[u] = accB[m2][r16] * gt;
                });
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) z2f[k][pr] = cvt_pk_f16(v2[2 * pr], v2[2 * pr + 1]);
                if constexpr (NEEDW) __builtin_amdgcn_raw_buffer_store_b128(z2f[k], rsZ, stoff, k * 1024, 0);
                if constexpr (k + E2D < KSC) load_e2(std::integral_constant<int, k + E2D>{});
                if constexpr (k < 3) load_w(kc);   // (behind the build of the first fragments: the ring's three k-steps)
            });
        }

        B1_STAMP(2)
        // ---- phase C: dE1 = W2'^T dZ2, k-outer, two fp16 terms
        f32x16 accC[T1];
        static_for<0, KSC>([&](auto kc) {
            MPG_CI(k, kc);
            const V b0 = __builtin_bit_cast(V, z2f[k]);
            static_for<0, T1>([&](auto mc) {
                MPG_CI(m, mc);
                const V a_h = ah[k % 3][m], a_l = al[k % 3][m];
                if constexpr (k == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    accC[m] = mma(a_l, b0, z);
                } else {
                    accC[m] = mma(a_l, b0, accC[m]);
                }
                accC[m] = mma(a_h, b0, accC[m]);
            });
            if constexpr (k + 3 < KSC) load_w(std::integral_constant<int, k + 3>{});
            __builtin_amdgcn_sched_barrier(0);
        });

        B1_STAMP(3)
        // ---- dZ1 = dE1 * keep1 * phi'(Z1) (back in plain units: the gate constants carry 2^-e / SC_W2) ; da_i += dZ1 ; dc_j = sum_i dZ1
        {
            const bool lb0 = lane & 1, lb1 = lane & 2, lbb2 = lane & 4, lb3 = lane & 8;
            float vone1 = sone1 * __builtin_amdgcn_rcpf(dither_of((uint32_t)blk));
            asm volatile("" : "+v"(vone1));
            const float valpha1 = p.alpha * vone1;
            int oln = lane;
            asm volatile("" : "+v"(oln));
            const int dcslot = 16 * ((oln >> 3) & 1) + 8 * ((oln >> 2) & 1) + 4 * (oln >> 5) + (oln & 3);
            const bool dcown = !(oln & 16);
            float* dcj = p.dc + ((size_t)rb * p.B * p.N + (size_t)(b * p.N + jj)) * H1;
            static_for<0, T1>([&](auto mmc) {
                MPG_CI(mm, mmc);
                float ured[4];
                static_for<0, 4>([&](auto suc) {
                    MPG_CI(su, suc);
                    constexpr int s2 = su >> 1, u = su & 1;
                    const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, mm, 4 * s2 + 2 * u + h, h);
                    // Z1 = a_i + c_j of the four features 32 mm + 16 s + 8 u + 4 h + t, as the forward adds them
                    const f32x4 a4 = lds_frag<f32x4>(lbla, ((mm * 2 + s2) * 2 + u) * 1024);
                    const f32x4 c4 = lds_frag<f32x4>(lbc, (32 * mm + 16 * s2 + 8 * u) * 4);
                    float dz[4];
                    static_for<0, 4>([&](auto tc) {
                        MPG_CI(t, tc);
                        const float z1 = c4[t] + a4[t];
                        float gt = sel_by_bit<31>(__builtin_bit_cast(uint32_t, z1), valpha1, vone1);
                        gt = drop_apply<DROP>(gt, wd, 16 * s2 + 8 * u + t, t, p.thr);
                        dz[t] = accC[mm][8 * s2 + 4 * u + t] * gt;
                        dacc[mm][8 * s2 + 4 * u + t] += dz[t];
                    });
                    const float w0 = halve_add<0xB1>(lb0, dz[0], dz[1]), w1 = halve_add<0xB1>(lb0, dz[2], dz[3]);
                    ured[su] = halve_add<0x4E>(lb1, w0, w1);
                });
                const float x0 = halve_add<0x124>(lbb2, ured[0], ured[1]), x1 = halve_add<0x124>(lbb2, ured[2], ured[3]);
                float y = halve_add<0x128>(lb3, x0, x1);
                y += __shfl_xor(y, 16, 64);
                if (dcown) dcj[32 * mm + dcslot] = y;
            });
        }
        B1_STAMP(4)
    }
#ifdef MPG_B1_STAMP
    const unsigned long long b1_l1 = __builtin_amdgcn_s_memtime();
#endif

    // ---- da: reduce over the eight waves (disjoint sender subsets), in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int q = 0; q < T1; ++q)
#pragma unroll
        for (int k = 0; k < 16; ++k) red[((w * T1 + q) * 16 + k) * 64 + lane] = dacc[q][k];
    __syncthreads();
    // (16 bytes per lane: registers 8s + 4u .. + 3 of a tile are four consecutive features of the lane's receiver)
    float* out = p.da + ((size_t)sc * p.B + b) * p.N * H1;
    for (int u = tid; u < T1 * 4 * 64; u += 512) {
        const int ln = u & 63, qg = u >> 6, q = qg >> 2, g = qg & 3;
        float4 v;
        float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = (q * 16 + 4 * g + t) * 64 + ln;
            float sum = red[e];
#pragma unroll
            for (int ww = 1; ww < B1_NW; ++ww) sum += red[e + ww * T1 * 1024];
            vp[t] = sum;
        }
        const int ii = rb * 32 + (ln & 31);
        if (ii < p.N) *reinterpret_cast<float4*>(out + (size_t)ii * H1 + 32 * q + 8 * g + 4 * (ln >> 5)) = v;
    }
    if constexpr (EPI != 0) {
        // ---- epilogue chains on this jet's nodes (edge_bwd2_impl.h): the rows of da (just written) and of dc (written sender by
        //      sender in the loop, zeros for masked senders in the prologue) are this workgroup's own stores
        const int m0 = b * p.N + rb * 32, nrows = min(32, p.N - rb * 32);
        {
            const MpgChain& c = *cdxp;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            auto stage = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t s_lo, const uint32_t s_hi, const float ascale) {
                c2_stage_rows<false, 12, 0, B1_NW>(c, m0, nrows, smem, first_tile, bias_request, bias_store, s_lo, s_hi, ascale);
            };
            c2_body<false, 12, 0, 0, 0, 0, 1, EPI == 3, B1_NW>(c, m0, nrows, smem, smem + C2_FB, reinterpret_cast<float*>(smem + 2 * C2_FB), stage);
        }
        if constexpr (EPI != 3) {
            if (cnxp->nlayers > 0) {
                const MpgChain& c = *cnxp;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                auto stage = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t s_lo, const uint32_t s_hi, const float ascale) {
                    c2_stage_rows<false, 2, DROP, B1_NW>(c, m0, nrows, smem, first_tile, bias_request, bias_store, s_lo, s_hi, ascale);
                };
                c2_body<false, 2, 16, 16, DROP, 3, 0, EPI == 2, B1_NW>(c, m0, nrows, smem, smem + C2_FB, reinterpret_cast<float*>(smem + 2 * C2_FB), stage);
            }
        }
    }
#ifdef MPG_B1_STAMP
    if (lane == 0) {
        unsigned long long* o = b1_stamps + ((size_t)blockIdx.x * 8 + w) * 10;
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        for (int q = 0; q < 5; ++q) o[q] = b1_acc[q];
        o[5] = (nvalid - w + B1_NW - 1) / B1_NW; o[6] = b1_l0 - b1_k0; o[7] = t1 - b1_l1; o[8] = t1 - b1_k0; o[9] = __builtin_amdgcn_s_memrealtime() - b1_r0;
    }
#endif
}

template <int DROP, bool NEEDW>
__global__ __launch_bounds__(512) void edge_bwd1_kernel(const MpgEdgeBwd p) { edge_bwd1_body<DROP, NEEDW, 0>(p); }

template <int DROP, bool NEEDW, int EPI>
__global__ __launch_bounds__(512) void edge_bwd1_fn_kernel(const MpgEdgeBwd p, const MpgChain cdx, const MpgChain cnx) {
    edge_bwd1_body<DROP, NEEDW, EPI>(p, &cdx, &cnx);
}

// the epilogue forms of one dropout mode / NEEDW (edge_bwd_fn_*.hip); epi = 1, 2, 3 as in edge_bwd2_impl.h
template <int D, bool NEEDW>
int b1_launch_fn(const MpgEdgeBwd* p, const MpgChain* cdx, const MpgChain* cnx, int epi, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB), block(512);
    MpgChain none = {};   // nlayers = 0: no second chain
    if (cnx == nullptr) cnx = &none;
#define MPG_B1FN(E)                                                                                         \
    do {                                                                                                    \
        MPG_ENSURE_LDS((edge_bwd1_fn_kernel<D, NEEDW, E>), B2_LDS_BYTES);                                   \
        hipLaunchKernelGGL((edge_bwd1_fn_kernel<D, NEEDW, E>), grid, block, B2_LDS_BYTES, st, *p, *cdx, *cnx); \
    } while (0)
    if (epi == 1) MPG_B1FN(1);
    else if (epi == 2) MPG_B1FN(2);
    else MPG_B1FN(3);
#undef MPG_B1FN
    return (int)hipGetLastError();
}

template <int D>
int b1_launch(const MpgEdgeBwd* p, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB * p->SC), block(512);
    if (p->stageZ2 != nullptr) {
        MPG_ENSURE_LDS((edge_bwd1_kernel<D, true>), B2_LDS_BYTES);
        hipLaunchKernelGGL((edge_bwd1_kernel<D, true>), grid, block, B2_LDS_BYTES, st, *p);
    } else {
        MPG_ENSURE_LDS((edge_bwd1_kernel<D, false>), B2_LDS_BYTES);
        hipLaunchKernelGGL((edge_bwd1_kernel<D, false>), grid, block, B2_LDS_BYTES, st, *p);
    }
    return (int)hipGetLastError();
}

}  // namespace
