// Backward of the fused edge network, data-gradient path (see edge_fwd2_impl.h for the forward and the chain layout).
//
//   dZ3 = m_j * dagg_i * keep3 * phi'(Z3)      phi'(Z3) from the forward's saved sign words
//   dE2 = W3'^T dZ3 ;  dZ2 = dE2 * keep2 * phi'(Z2)     phi'(Z2) from the SIGN of the E2 fragments the forward parked
//   dE1 = W2'^T dZ2 ;  dZ1 = dE1 * keep1 * phi'(Z1) ;  da_i = sum_j dZ1 ;  dc_j = sum_i dZ1     (Z1 = a_i + c_j, one add)
//
// Nothing of the forward is recomputed: E2 = fe.net.1's output is parked by the forward anyway (fp16 fragments, for the
// weight-gradient kernel), and all this kernel needs of layer 2 is its sign pattern -- which LeakyReLU preserves.  (The
// recomputation -- 180 MFMAs and 2,600 VALU instructions per pair of senders, a third of this kernel -- bought 59 MB
// less HBM traffic per launch in a kernel that is bound by VALU issue.)
//
// A workgroup owns one (jet, 32 receivers, sender chunk); its four waves split the chunk's UNMASKED senders and are
// independent of each other between the prologue and the final reduction of da.  Every MFMA needs a 1 KiB weight
// fragment, and a CU moves 64 B/clk from L1/L2 and 128 B/clk from LDS.  So a wave walks its senders in PAIRS: each
// weight fragment (W3^T from LDS, W2^T streamed from L2) feeds the MFMAs of two senders, and the pair's operands are
// built just in time:
//   phase B  dE2 = W3'^T dZ3     k-outer (12 k-steps x 5 tiles): the dZ3 fragment of k-step k+1 is built behind the
//                                MFMAs of k-step k, so dZ3 never exists as a whole
//   phase C  dE1 = W2'^T dZ2     k-outer (10 k-steps x 3 tiles): dZ2 = gate * dE2 built (and parked for the
//                                weight-gradient kernel) the same way from phase B's accumulators
// Arithmetic.  The two gradient products have no kink behind their rounding and run as TWO fp16
// terms: the weight image keeps hi and lo, the operand built on the fly (dZ3, dZ2) is rounded to ONE fp16 value
// (2^-12 relative per element).  Left alone that rounding would be the same for every sender of a receiver -- dZ3_ij is
// dagg_i times one of two constants, dZ2_ij nearly so -- i.e. coherent along the very axis da sums over (measured:
// 2.8e-4 on dx at any batch size).  So every sender works in ITS OWN unit: the pair's gradients are multiplied by a
// dither factor c in [1, 2) hashed from the block index (free: it rides in the slope constants of layer 3), which puts
// the roundings of different senders at unrelated places of the mantissa, and 1/c is folded into the gate constants of
// layer 1 (and into mpg_edge_dw's rescaling of the parked dZ2).  The roundings then average out like independent noise in
// every sum over edges (tests/test_gpu_mplayer.py, tests/probe_precision.py).
// fp16 has 30 binades, gradients have any magnitude, so the workgroup works in units of 2^-e of its own receivers'
// upstream gradient: e is chosen in the prologue such that max |dZ3| * 2^e lies in [2^8, 2^9), the factor rides for free
// in constants that already multiply every element (the slope constants of layer 3; the gates of layers 2 and 1 undo
// the images' operand scales and 2^e), and gexp[(b, rb)] = e tells mpg_edge_dw the unit of the staged dZ2.
#pragma once
#include "edge_common.h"
#include "chain2_impl.h"

#ifndef MPG_B2EXP
#define MPG_B2EXP 0  // experiment bits (tools/ubench/bwd2_bench.hip): 1 streamed fragments all from k-step 0 (L1 hits), 2 no parking stores
#endif

#ifdef MPG_B2STAMP  // diagnostic build: s_memtime at the phase boundaries of the first pair of every wave of the first 64 workgroups
__device__ unsigned long long g_b2_stamps[64 * 4 * 8];
#define B2_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); b2_st[i] = pq == w ? t_ : b2_st[i]; } while (0)
#else
#define B2_STAMP(i) do {} while (0)
#endif

namespace {

// LDS plan (all of the 160 KiB): W3^T hi|lo (fp16) | dagg tile of the 32 receivers | a tile | (640 B unused) | per-wave rows of
// c for the two senders in flight | list of the chunk's unmasked senders.  W2 and W2^T stream from L2.
constexpr int B2_W_BYTES = 2 * NF3T * 1024;      // 122,880
constexpr int B2_DG_BYTES = T3 * 4 * 64 * 16;    //  24,576
constexpr int B2_A_BYTES = T1 * 4 * 64 * 16;     //  12,288
constexpr int B2_B2_BYTES = H2 * 4;              //     640
constexpr int B2_C_BYTES = 4 * 2 * H1 * 4;       //   3,072
constexpr int B2_LIST_MAX = 160;                 // senders per chunk (uint16 entries + count + the wave maxima of the prologue: 384 B)
// Without edge scalars the 640 bytes b2 once had hold the listed senders' MASK ENTRIES in list order (160 floats), and the
// sender loop takes m_j from there instead of from memory at the top of a pair (see F2_MK_OFF, edge_fwd2_impl.h)
constexpr int B2_LDS_BYTES = B2_W_BYTES + B2_DG_BYTES + B2_A_BYTES + B2_B2_BYTES + B2_C_BYTES + 384;
static_assert(B2_LDS_BYTES <= 163840, "LDS plan exceeds 160 KiB");
// with edge scalars (NQ > 0) the columns wq [2][96] take the 640 bytes left over from b2 (nothing is recomputed any more)
// and the first 128 bytes of the list area: a chunk then holds at most 116 senders
constexpr int B2_Q_OFF = B2_W_BYTES + B2_DG_BYTES + B2_A_BYTES;
constexpr int B2_LIST_MAX_Q = 116;
constexpr int B2_Q_BYTES = MPG_EDGE_SCALARS * H1 * 4;   // 768: the rows of c and the list move up by 128 bytes
static_assert(2 * B2_LIST_MAX_Q + 4 + 16 <= 384 - (B2_Q_BYTES - B2_B2_BYTES), "list area with edge scalars");

typedef unsigned int b2_u32x4 __attribute__((ext_vector_type(4)));

// two floats -> one word of an fp16x8 fragment (v_cvt_pk_f16_f32, round to nearest even)
MPG_DEV uint32_t cvt_pk_f16(float v0, float v1) {
    const f16x2 hp = {(_Float16)v0, (_Float16)v1};
    return __builtin_bit_cast(uint32_t, hp);
}

// units u of [0, NU) that fall into slot SL of NS
template <int NU, int NS, int SL, typename F>
MPG_DEV void run_slot(F&& unit) {
    constexpr int u0 = (SL * NU) / NS, u1 = ((SL + 1) * NU) / NS;
    static_for<u0, u1>(unit);
}

// e with m * 2^e in [2^8, 2^9) for a positive normal m, clamped to +-100 so that 2^e, 2^-e / SC_W2 stay normal floats
// (gradients below 2^-92 are flushed by the fp16 rounding, as RMSprop's eps would do anyway).  fp16 is normal from 2^-14
// to 2^16: with the largest |dZ3| (times its dither factor < 2) below 2^10, elements down to 2^-23 of the largest keep
// their 11 bits, and dZ2 = W3^T dZ3 -- typically 1 .. 10 times the largest |dZ3| -- stays two binades and more below the
// overflow.  m = 0 (no gradient at all: nothing to scale) gives the largest exponent, so that it never decides the
// minimum mpg_edge_dw takes over a launch.
MPG_DEV int grad_unit_exp(float m) {
    const int ex = (int)((__builtin_bit_cast(uint32_t, m) >> 23) & 0xffu);
    if (ex == 0) return 100;
    return max(-100, min(100, 8 - (ex - 127)));
}

MPG_DEV f32x16 mma(const f16x8 a, const f16x8 b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// DROP: 0 off, 1 byte mode, 2 bit mode (see common.h); NQ: edge scalars (0 or MPG_EDGE_SCALARS): Z1 = a_i + c_j + sum_q es wq[q],
// and the kernel also returns des = dZ1 . wq[q] per edge and daq = sum_j es dZ1 per receiver
// EPI: what the workgroup does BEHIND its data-gradient work, as an epilogue on its own 32 nodes (a whole jet per workgroup:
// N <= 32, SC = 1 -- then the jet's rows of da AND dc are this workgroup's own stores):
//   0  nothing;
//   1 / 2 / 3  the layer's input gradient  dx = [da | dc] [W1a ; W1c] + dx(node path)  (the chain `cdx`: mpg_chain's "dx from da | dc"
//      call) and -- 1, 2, when `cnx` has layers -- the NEXT-LOWER MPLayer's node-network input-gradient chain on those dx rows
//      (the backward of its fn, mpgan/model.py:279, as mpg_chain takes it: dz2 = gate(V3^T dy), dz1 = gate(V2^T dz2),
//      [dagg | dx] = V1^T dz1), with chain2's schedule (chain2_impl.h).  2: that chain's last rows are not whole 16-byte groups
//      (a 195-column [dagg | dx]); 3: dx's own rows are not (3 features) and nothing follows.
// A workgroup whose jet has few senders does this while the fullest jets' workgroups are still in their sender loops: the
// launch is as long as its slowest workgroup, and the chains of most jets hide in its tail (measured for the forward's epilogue,
// edge_fwd2_impl.h: +2.8 %; as a PROLOGUE -- on every workgroup's critical path -- the same chain was 1.2 % slower than its own launch).
static_assert(C2_LDS <= B2_LDS_BYTES, "the chains' buffers must fit the data-gradient kernel's LDS");
template <int DROP, bool NEEDW, int NQ, int EPI>
MPG_DEV void edge_bwd_body(const MpgEdgeBwd& p, const MpgChain* const cdxp, const MpgChain* const cnxp) {
    constexpr int QB = NQ > 0 ? B2_Q_BYTES : B2_B2_BYTES;   // bytes between the a tile and the rows of c
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
    const int JC = (p.N + p.SC - 1) / p.SC;
    const int jbeg = sc * JC, jend = min(p.N, jbeg + JC);
    const int ldac = p.ld_ac ? p.ld_ac : H1;

    const __amdgpu_buffer_rsrc_t r2t = img_rsrc(p.W2Timg, 2 * NF2T);  // W2^T hi | lo (fp16, operand scale SC_W2)
    const int lane16 = lane * 16;
    const f16x8* t3g = reinterpret_cast<const f16x8*>(p.W3Timg);
    f16x8* l3t = reinterpret_cast<f16x8*>(smem);
    float4* ldg = reinterpret_cast<float4*>(smem + B2_W_BYTES);                // [(m*4+g)][lane]
    float4* la = reinterpret_cast<float4*>(smem + B2_W_BYTES + B2_DG_BYTES);   // [(q*2+s)*2+u][lane]
    float* lwq = reinterpret_cast<float*>(smem + B2_Q_OFF);                    // wq [NQ][96] (times SC_A, like a and c)
    float* lcw = reinterpret_cast<float*>(smem + B2_Q_OFF + QB) + w * (2 * H1); // this wave's two rows of c
    unsigned short* lst = reinterpret_cast<unsigned short*>(smem + B2_Q_OFF + QB + B2_C_BYTES);
    int* lnv = reinterpret_cast<int*>(lst + (NQ > 0 ? B2_LIST_MAX_Q : 180));
    float* lmk = lwq;   // (NQ == 0 only)
    float* lmx = reinterpret_cast<float*>(smem + B2_LDS_BYTES - 16);                // wave maxima of |dagg|

    // ---- prologue (whole workgroup): weights and the per-receiver tiles into LDS, the list of unmasked senders
    // Order of issue as in the forward (edge_fwd2_impl.h): the receivers' rows of dagg and a into registers, then W3^T's image by
    // LDS-DMA; the registers go to LDS when everything has landed -- one wait for the whole prologue.
    // upstream gradient dagg (scaled) and the layer-1 receiver term a, both in the register order the chain layout
    // wants (zeros for padding lanes: they carry exact zeros all the way down)
    static_assert(T3 * 4 * 64 == 6 * 256 && T1 * 4 * 64 == 3 * 256, "the prologue's register sets");
    float4 dv[6], av[3];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int t = tid + 256 * u;
        const int ln = t & 63, mg = t >> 6, rr = ln & 31, hh = ln >> 5, ii = min(rb * 32 + rr, p.N - 1);
        const float* di = p.dagg + (size_t)(b * p.N + ii) * p.ld_dagg + 32 * (mg >> 2) + 8 * (mg & 3) + 4 * hh;
        dv[u] = make_float4(di[0], di[1], di[2], di[3]);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = tid + 256 * u;
        const int ln = t & 63, qsu = t >> 6, rr = ln & 31, hh = ln >> 5, ii = min(rb * 32 + rr, p.N - 1);
        av[u] = ld4(p.a + (size_t)(b * p.N + ii) * ldac + 8 * qsu + 4 * hh);
    }
    fill_lds_dma(l3t, t3g, 2 * NF3T * 1024, tid);
    float amax = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int t = tid + 256 * u;
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
    for (int u = 0; u < 3; ++u) {
        const int t = tid + 256 * u;
        const bool in = rb * 32 + (t & 31) < p.N;
        la[t] = in ? make_float4(av[u].x * SC_A, av[u].y * SC_A, av[u].z * SC_A, av[u].w * SC_A) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (NQ > 0)
        for (int t = tid; t < NQ * H1; t += 256) lwq[t] = p.wq[t] * SC_A;
    // unmasked senders, in order: of the whole jet when they all fit the list -- which is then cut into SC equal parts (chunks
    // by sender index are as uneven as the mask; see edge_fwd2_impl.h) --, else of the chunk's index range
    const bool whole = p.N <= (NQ > 0 ? B2_LIST_MAX_Q : B2_LIST_MAX);
    const int lbeg = whole ? 0 : jbeg, lend = whole ? p.N : jend;
    if (w == 0) {
        int cnt = 0;
        for (int j0 = lbeg; j0 < lend; j0 += 64) {
            const int j = j0 + lane;
            const float mv = (j < lend && p.mask != nullptr) ? p.mask[b * p.N + j] : 1.f;
            const bool ok = j < lend && mv != 0.f;
            const unsigned long long bits = __ballot(ok);
            const int pos = cnt + __popcll(bits & ((1ull << lane) - 1ull));
            if (ok) {
                lst[pos] = (unsigned short)j;
                if constexpr (NQ == 0) lmk[pos] = mv;
            }
            cnt += __popcll(bits);
        }
        if (lane == 0) *lnv = cnt;
    }
    // masked senders: every gradient through their edges is exactly zero (mpg_edge_dw skips those blocks too)
    if (p.mask != nullptr)
        for (int t = tid; t < (jend - jbeg) * (H1 / 4); t += 256) {
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
    // the workgroup's gradient unit 2^-e (see the head of this file): max |dZ3| <= max |dagg * agg_scale| * dscale (mask <= 1)
    const int gexp = __builtin_amdgcn_readfirstlane(grad_unit_exp(fmaxf(fmaxf(lmx[0], lmx[1]), fmaxf(lmx[2], lmx[3])) * p.dscale));
    const float gunit = __builtin_bit_cast(float, (uint32_t)(gexp + 127) << 23);            // 2^e
    if (NEEDW && tid == 0) p.gexp[b * RB + rb] = gexp;   // (the sender chunks of a receiver block all write the same value)

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    const uint32_t lb3t = lds_base(smem, lane16);                 // W3^T hi fragments; lo at + NF3T KiB
    const uint32_t lbdg = lds_base(smem, B2_W_BYTES + lane16);    // dagg tile
    const uint32_t lbla = lds_base(smem, B2_W_BYTES + B2_DG_BYTES + lane16);  // a tile
    const uint32_t lbc = lds_base(smem, B2_Q_OFF + QB + w * (2 * H1 * 4) + 16 * h);  // this wave's rows of c
    const uint32_t lbq = lds_base(smem, B2_Q_OFF + 16 * h);
    // staging: block blk = (b*RB + rb)*N + j ; the 10 B-operand fragments (tile, k-step) of the 160-feature tensor,
    // rounded to fp16, exactly as the lanes hold them: one coalesced 16-byte store per lane and fragment.
    // Buffer stores: a block offset beyond the buffer (the idle second half of an odd pair) is dropped by the hardware.
    const int nblk = p.B * RB * p.N;
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageE2), 0, nblk * (NFR2 * 1024), 0x00020000);   // (read)
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(p.stageZ2, 0, NEEDW ? nblk * (NFR2 * 1024) : 0, 0x00020000);

    uint32_t opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    // the two slopes in vector registers (operands of v_bfi), times what the gate has to undo: layer 2's gate takes
    // phase B's accumulators (W3^T image: SC_W3) to dZ2 in gradient units, layer 1's gate takes phase C's
    // (W2^T image: SC_W2, gradient unit 2^e) to plain dZ1
    // (wave-uniform values; each phase moves its pair into vector registers for its own duration only)
    // (times 1 / c_j of the sender, there).  gfx950 has no scalar float ALU: what is not integer arithmetic is recomputed where it is
    // used -- kept across the pair loop such a value is the first thing the register allocator spills
    const float sone2 = 1.f / SC_W3;
    const float sone1 = __builtin_bit_cast(float, (uint32_t)(127 - gexp - 4) << 23);   // 2^-e / SC_W2 (SC_W2 = 2^4): integer arithmetic, a scalar register
    static_assert(SC_W2 == 16.f, "sone1 assumes SC_W2 = 2^4");
    float dacc[T1][16];
#pragma unroll
    for (int q = 0; q < T1; ++q)
#pragma unroll
        for (int k = 0; k < 16; ++k) dacc[q][k] = 0.f;

#ifdef MPG_B2STAMP
    unsigned long long b2_st[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    // What a pair needs from memory -- its senders' sign words and rows of c -- is requested one pair ahead, at the top of
    // the previous pair: vector memory operations complete in order, and requested at its own top a pair's loads queue up
    // behind all the staging stores of the pair before (measured: ~1.2k clk of every pair).  Past the end the indices are
    // clamped (the last pair is fetched again, unused).
    uint32_t psw[2][T3 / 2];
    float pc0[2], pc1[2];
    float pes[2][NQ > 0 ? NQ : 1];
    auto prefetch = [&](int pq2) {
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const int jn = __builtin_amdgcn_readfirstlane((int)lst[max(0, min(2 * pq2 + sd, nvalid - 1))]);
            const int blk = (b * RB + rb) * p.N + jn;
#pragma unroll
            for (int q = 0; q < T3 / 2; ++q) psw[sd][q] = p.sign3[(size_t)blk * (T3 * 32) + q * 64 + lane];
            const float* cj = p.c + (size_t)(b * p.N + jn) * ldac;
            pc0[sd] = cj[lane];
            pc1[sd] = cj[64 + (lane & 31)];
            if constexpr (NQ > 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) pes[sd][q] = p.es[((size_t)(b * p.N + jn) * NQ + q) * p.N + (i < p.N ? i : 0)];
            }
        }
    };
    // (with edge scalars) per receiver: sum over this wave's senders of es(i, j, q) * dZ1(i, j) -- the gradient of wq's column
    // once summed over receivers
    float dqacc[NQ > 0 ? NQ : 1][T1][16];
    if constexpr (NQ > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int m = 0; m < T1; ++m)
#pragma unroll
                for (int k = 0; k < 16; ++k) dqacc[q][m][k] = 0.f;
    }
    prefetch(w);
    for (int pq = w; 2 * pq < nvalid; pq += 4) {
        const bool has2 = 2 * pq + 1 < nvalid;
        B2_STAMP(0);
        int jj[2];
        jj[0] = __builtin_amdgcn_readfirstlane((int)lst[2 * pq]);
        jj[1] = has2 ? __builtin_amdgcn_readfirstlane((int)lst[2 * pq + 1]) : jj[0];
        // (the lane's first edge row, from an opaque copy of the lane id: a loop-invariant per-lane value kept across the
        // pair loop is what the register allocator spills first, and every scratch reload drains vmcnt)
        int oln0 = lane;
        asm volatile("" : "+v"(oln0));
        const uint32_t ebase = (uint32_t)((b * p.N + rb * 32 + (oln0 & 31)) * p.N);
        float cpos[2], cneg[2];     // slope of layer 3 times m_j * dscale * 2^e * c_j: one select between two uniform constants
        // erow2 == erow, in a form the optimiser cannot prove equal: every dropout word is needed twice per pair, phases
        // apart (forward recomputation, then the gate of the matching gradient), and with one visible value it keeps
        // all the one-instruction keep-masks of the first use alive for the second (hundreds of registers, spilled)
        uint32_t erow[2], erow2[2];
        int stoff[2];               // byte offset of the sender's staging block (out of range for an idle half)
        int stsc[2];                // the block's offset as a scalar (reads of the parked E2)
        uint32_t sw[2][T3 / 2];     // this lane's 96 sign bits of Z3 per sender
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            float mj;
            if constexpr (NQ == 0) mj = lmk[min(2 * pq + sd, nvalid - 1)];
            else mj = p.mask ? p.mask[b * p.N + jj[sd]] : 1.f;
            const float mjs = (sd == 0 || has2) ? mj * p.dscale * gunit : 0.f;
            float in_set = 1.f;
            if (p.nbr != nullptr) {  // k-nearest-neighbour graph: the edge (i, j) exists only if j's bit is set in i's row
                const unsigned int wb = p.nbr[(size_t)(b * p.N + (i < p.N ? i : 0)) * ((p.N + 31) >> 5) + (jj[sd] >> 5)];
                in_set = ((wb >> (jj[sd] & 31)) & 1u) ? 1.f : 0.f;
            }
            const int blk = (b * RB + rb) * p.N + jj[sd];
            const float dth = dither_of((uint32_t)blk);  // this sender's unit within the workgroup's (see the head of this file)
            cpos[sd] = mjs * in_set * dth; cneg[sd] = mjs * p.alpha * in_set * dth;
            erow[sd] = ebase + (uint32_t)jj[sd];
            erow2[sd] = erow[sd] + opaque_zero;
            stoff[sd] = (sd == 0 || has2) ? blk * (NFR2 * 1024) + lane16 : (int)0x7ffffff0;
            stsc[sd] = blk * (NFR2 * 1024);
#pragma unroll
            for (int q = 0; q < T3 / 2; ++q) sw[sd][q] = psw[sd][q];
            // the sender's row of c into this wave's LDS slot (96 floats: lanes 0..63, then lanes 0..31)
            lcw[sd * H1 + lane] = pc0[sd] * SC_A;
            if (lane < H1 - 64) lcw[sd * H1 + 64 + lane] = pc1[sd] * SC_A;
        }
        float esv[2][NQ > 0 ? NQ : 1];
        if constexpr (NQ > 0) {
#pragma unroll
            for (int sd = 0; sd < 2; ++sd)
#pragma unroll
                for (int q = 0; q < NQ; ++q) esv[sd][q] = (sd == 0 || has2) ? pes[sd][q] : 0.f;
        }
        prefetch(pq + 4);

        // (no recomputation of layer 2: E2 = fe.net.1's output comes back from memory, parked by the forward as the fp16
        //  fragments the lanes hold -- requested above, first used by phase C's gate)
        // the parked E2 fragments (10 x 16 B per lane and sender, one per k-step of phase C) come from HBM: a ring of
        // E2D + 1, the first E2D requested here -- a whole phase B ahead -- and one more per k-step of phase C
        constexpr int E2D = 4;
        b2_u32x4 e2g[E2D + 1][2];
        auto load_e2 = [&](auto kc) {
            MPG_CI(k, kc);
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) e2g[k % (E2D + 1)][sd] = __builtin_amdgcn_raw_buffer_load_b128(rsE, lane16, stsc[sd] + k * 1024, 0);
        };
        static_for<0, E2D>([&](auto kc) { load_e2(kc); });
        B2_STAMP(1); B2_STAMP(2); B2_STAMP(3);
        // ---- phase B: dE2 = W3'^T dZ3, k-outer, two fp16 terms (W3^T lo, hi x dZ3 rounded to fp16).  dZ3 = dagg * slope(sign
        //      word) * keep3 in the sender's gradient unit is built one k-step ahead.
        f32x16 accB[T2][2];
        {
            constexpr int KS = T3 * 2;  // 12 k-steps of 16 features of layer 3
            b2_u32x4 zz[2][2];            // [buffer][sender] dZ3 fragment of a k-step
            float v3[2][8];
            f32x4 dg[2];                  // dagg of the k-step being built (shared by the two senders)
            uint32_t wd3[2] = {0u, 0u};
            V ah[2][T2], al[2][T2];       // [buffer][tile] W3^T fragments of a k-step
            auto load_w = [&](auto kc) {
                MPG_CI(k, kc);
#pragma unroll
                for (int m = 0; m < T2; ++m) {
                    ah[k & 1][m] = lds_frag<V>(lb3t, (m * KS + k) * 1024);
                    al[k & 1][m] = lds_frag<V>(lb3t, (NF3T + m * KS + k) * 1024);
                }
            };
            auto load_dg = [&](auto kc) {
                MPG_CI(k, kc);
                constexpr int m3 = k >> 1, s = k & 1;
                dg[0] = lds_frag<f32x4>(lbdg, ((m3 * 4 + 2 * s) * 64) * 16);
                dg[1] = lds_frag<f32x4>(lbdg, ((m3 * 4 + 2 * s + 1) * 64) * 16);
            };
            // build units of the dZ3 fragment of k-step k: per sender 8 element units + 4 pair conversions = 12
            auto buildB = [&](auto kc, auto uc) {
                MPG_CI(k, kc); MPG_CI(uu, uc);
                constexpr int sd = uu / 12, u = uu % 12;
                constexpr int m3 = k >> 1, s = k & 1;
                if constexpr (u < 8) {
                    constexpr int r16 = 8 * s + u, g = r16 >> 2, t = r16 & 3;
                    if constexpr (DROP == 2) { if constexpr (u == 0 && s == 0) wd3[sd] = drop_tile_word<2>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow[sd], m3, 0, h); }
                    uint32_t wd = wd3[sd];
                    if constexpr (DROP == 1) wd = drop_tile_word<1>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow[sd], m3, 2 * g + h, h);
                    const float dd = dg[u >> 2][t];
                    // sign bit of this lane's Z3 register (tile m3, r16): see the forward's epi3
                    const float sel = dd * sel_by_bit<31 - (16 * (m3 & 1) + r16)>(sw[sd][m3 >> 1], cneg[sd], cpos[sd]);
                    v3[sd][u] = drop_apply<DROP>(sel, wd, 8 * g + t, t, p.thr);
                } else {
                    constexpr int pr = u - 8;
                    zz[k & 1][sd][pr] = cvt_pk_f16(v3[sd][2 * pr], v3[sd][2 * pr + 1]);
                }
            };
            load_w(std::integral_constant<int, 0>{});
            load_dg(std::integral_constant<int, 0>{});
            static_for<0, 24>([&](auto uc) { buildB(std::integral_constant<int, 0>{}, uc); });
            static_for<0, KS>([&](auto kc) {
                MPG_CI(k, kc);
                if constexpr (k + 1 < KS) {
                    load_dg(std::integral_constant<int, k + 1>{});  // (first: the build units behind the next MFMAs wait for it)
                    load_w(std::integral_constant<int, k + 1>{});
                }
                const V b0 = __builtin_bit_cast(V, zz[k & 1][0]), b1 = __builtin_bit_cast(V, zz[k & 1][1]);
                static_for<0, T2>([&](auto mc) {
                    MPG_CI(m, mc);
                    auto slot = [&](auto slc) {
                        MPG_CI(SL, slc);
                        // the first slots leave the LDS reads of the next k-step time to land
                        if constexpr (k + 1 < KS && SL >= 2) run_slot<24, 18, SL - 2>([&](auto uc) { buildB(std::integral_constant<int, k + 1>{}, uc); });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const V a_h = ah[k & 1][m], a_l = al[k & 1][m];
                    if constexpr (k == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        accB[m][0] = mma(a_l, b0, z); slot(std::integral_constant<int, 4 * m + 0>{});
                        accB[m][1] = mma(a_l, b1, z); slot(std::integral_constant<int, 4 * m + 1>{});
                    } else {
                        accB[m][0] = mma(a_l, b0, accB[m][0]); slot(std::integral_constant<int, 4 * m + 0>{});
                        accB[m][1] = mma(a_l, b1, accB[m][1]); slot(std::integral_constant<int, 4 * m + 1>{});
                    }
                    accB[m][0] = mma(a_h, b0, accB[m][0]); slot(std::integral_constant<int, 4 * m + 2>{});
                    accB[m][1] = mma(a_h, b1, accB[m][1]); slot(std::integral_constant<int, 4 * m + 3>{});
                });
            });
        }

        B2_STAMP(4);
        // ---- phase C: dE1 = W2'^T dZ2, k-outer, two fp16 terms; dZ2 = dE2 * keep2 * phi'(Z2) (still in the sender's unit) is
        //      built -- and staged -- one k-step ahead from phase B's accumulators.  W2^T streams from L2 two k-steps ahead.
        f32x16 accC[T1][2];
        {
            constexpr int KS = T2 * 2;  // 10 k-steps of 16 features of layer 2
            float valpha2 = p.alpha, vone2 = sone2;
            asm volatile("" : "+v"(valpha2), "+v"(vone2));   // (opaque copies made HERE: the product below cannot be hoisted out of the pair loop)
            valpha2 *= vone2;
            b2_u32x4 zz[2][2];
            float v2[2][8];
            V ah[3][T1], al[3][T1];       // ring of three k-steps of W2^T fragments (L2 is more than one k-step away)
            // ... and of the parked E2 fragments (10 x 16 B per lane and sender, one per k-step): their SIGN is phi'(Z2) for
            // dZ2's gate (LeakyReLU keeps the sign; a dropped element is +0 and is zeroed by the regenerated keep mask)
            // (declared and first requested before phase B: e2g, E2D)
            uint32_t wd2c[2] = {0u, 0u};  // bit-mode dropout word of the layer-2 tile being gated (hashed again: see erow2)
            auto load_w = [&](auto kc) {
                MPG_CI(k, kc);
#pragma unroll
                for (int m = 0; m < T1; ++m) {
                    ah[k % 3][m] = img_frag<V>(r2t, lane16, m * KS + ((MPG_B2EXP & 1) ? 0 : k));
                    al[k % 3][m] = img_frag<V>(r2t, lane16, NF2T + m * KS + ((MPG_B2EXP & 1) ? 0 : k));
                }
            };

            // build units of the dZ2 fragment of k-step k: per sender 8 element units + 4 pair conversions + 1 store = 13
            auto buildC = [&](auto kc, auto uc) {
                MPG_CI(k, kc); MPG_CI(uu, uc);
                constexpr int sd = uu / 13, u = uu % 13;
                constexpr int m2 = k >> 1, s = k & 1;
                if constexpr (u < 8) {
                    constexpr int r16 = 8 * s + u, g = r16 >> 2, t = r16 & 3;
                    // sign of Z2 (register r16 of tile m2) = sign of element u of the parked E2 fragment 2 m2 + s
                    float gt = sel_by_bit<16 * (u & 1) + 15>(e2g[k % (E2D + 1)][sd][u >> 1], valpha2, vone2);
                    if constexpr (DROP != 0) {
                        uint32_t wd;
                        if constexpr (DROP == 2) {
                            if constexpr (u == 0 && s == 0) wd2c[sd] = drop_tile_word<2>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow2[sd], m2, 0, h);
                            wd = wd2c[sd];
                        } else wd = drop_tile_word<1>(seed_lo, seed_hi, p.tag_base + TAG_E1, erow2[sd], m2, 2 * g + h, h);
                        gt = drop_apply<DROP>(gt, wd, 8 * g + t, t, p.thr);
                    }
                    v2[sd][u] = accB[m2][sd][r16] * gt;
                } else if constexpr (u < 12) {
                    constexpr int pr = u - 8;
                    zz[k & 1][sd][pr] = cvt_pk_f16(v2[sd][2 * pr], v2[sd][2 * pr + 1]);
                } else {
                    if constexpr (NEEDW && !(MPG_B2EXP & 2))
                        __builtin_amdgcn_raw_buffer_store_b128(zz[k & 1][sd], rsZ, stoff[sd], k * 1024, 0);
                }
            };
            load_w(std::integral_constant<int, 0>{});
            load_w(std::integral_constant<int, 1>{});
            static_for<0, 26>([&](auto uc) { buildC(std::integral_constant<int, 0>{}, uc); });
            static_for<0, KS>([&](auto kc) {
                MPG_CI(k, kc);
                // (fragment k + 1 is read by the build units in this k-step's slots: slot (k + E2D) % (E2D + 1) is free)
                if constexpr (k + E2D < KS) load_e2(std::integral_constant<int, k + E2D>{});
                if constexpr (k + 2 < KS) load_w(std::integral_constant<int, k + 2>{});
                const V b0 = __builtin_bit_cast(V, zz[k & 1][0]), b1 = __builtin_bit_cast(V, zz[k & 1][1]);
                static_for<0, T1>([&](auto mc) {
                    MPG_CI(m, mc);
                    auto slot = [&](auto slc) {
                        MPG_CI(SL, slc);
                        if constexpr (k + 1 < KS) run_slot<26, 12, SL>([&](auto uc) { buildC(std::integral_constant<int, k + 1>{}, uc); });
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    const V a_h = ah[k % 3][m], a_l = al[k % 3][m];
                    if constexpr (k == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        accC[m][0] = mma(a_l, b0, z); slot(std::integral_constant<int, 4 * m + 0>{});
                        accC[m][1] = mma(a_l, b1, z); slot(std::integral_constant<int, 4 * m + 1>{});
                    } else {
                        accC[m][0] = mma(a_l, b0, accC[m][0]); slot(std::integral_constant<int, 4 * m + 0>{});
                        accC[m][1] = mma(a_l, b1, accC[m][1]); slot(std::integral_constant<int, 4 * m + 1>{});
                    }
                    accC[m][0] = mma(a_h, b0, accC[m][0]); slot(std::integral_constant<int, 4 * m + 2>{});
                    accC[m][1] = mma(a_h, b1, accC[m][1]); slot(std::integral_constant<int, 4 * m + 3>{});
                });
            });
        }

        B2_STAMP(5);
        // ---- dZ1 = dE1 * keep1 * phi'(Z1) (back in plain units: the gate constants carry 2^-e / SC_W2) ; da_i += dZ1 ; dc_j = sum_i dZ1
        {
            // dc_j = sum over the 32 receivers (lanes of one half) of dZ1: 16 values per tile and lane.  Halving
            // reduction: at each step a lane keeps half of its values and hands the other half to its partner
            // (DPP), so 16 values cost 15 exchanges instead of 80 and lane l ends with the total of value l & 15
            // (= accumulator register 8s + 4u + t  <->  feature 32 mm + 16 s + 8 u + 4 h + t).
            const bool lb0 = lane & 1, lb1 = lane & 2, lbb2 = lane & 4, lb3 = lane & 8;
            // (1 / c_j recomputed from the block index: two registers less across the pair)
            float vone1[2] = {sone1 * __builtin_amdgcn_rcpf(dither_of((uint32_t)((b * RB + rb) * p.N + jj[0]))),
                              sone1 * __builtin_amdgcn_rcpf(dither_of((uint32_t)((b * RB + rb) * p.N + jj[1])))};
            asm volatile("" : "+v"(vone1[0]), "+v"(vone1[1]));
            float valpha1[2] = {p.alpha * vone1[0], p.alpha * vone1[1]};
            // (the lane's slot in a row of dc, from an opaque copy of the lane id: hoisted out of the sender loop these few
            // loop-invariant values are what the register allocator spills -- and every scratch reload drains vmcnt)
            int oln = lane;
            asm volatile("" : "+v"(oln));
            const int dcslot = 16 * ((oln >> 3) & 1) + 8 * ((oln >> 2) & 1) + 4 * (oln >> 5) + (oln & 3);
            const bool dcown = !(oln & 16);
            static_for<0, 2>([&](auto sdc) {
                MPG_CI(sd, sdc);
                float* dcj = p.dc + ((size_t)rb * p.B * p.N + (size_t)(b * p.N + jj[sd])) * H1;
                float dsq[NQ > 0 ? NQ : 1];   // des(i, j, q) = dZ1(i, j) . wq[q]: this lane's half of the 96 features
#pragma unroll
                for (int q = 0; q < (NQ > 0 ? NQ : 1); ++q) dsq[q] = 0.f;
                static_for<0, T1>([&](auto mmc) {
                    MPG_CI(mm, mmc);
                    float ured[4];
                    static_for<0, 4>([&](auto suc) {
                        MPG_CI(su, suc);
                        constexpr int s = su >> 1, u = su & 1;
                        const uint32_t wd = drop_tile_word<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow2[sd], mm, 4 * s + 2 * u + h, h);
                        // Z1 = a_i + c_j of the four features 32 mm + 16 s + 8 u + 4 h + t, as the forward adds them
                        const f32x4 a4 = lds_frag<f32x4>(lbla, ((mm * 2 + s) * 2 + u) * 1024);
                        const f32x4 c4 = lds_frag<f32x4>(lbc, (sd * H1 + 32 * mm + 16 * s + 8 * u) * 4);
                        f32x4 q4[NQ > 0 ? NQ : 1];
                        if constexpr (NQ > 0) {
#pragma unroll
                            for (int q = 0; q < NQ; ++q) q4[q] = lds_frag<f32x4>(lbq, (q * H1 + 32 * mm + 16 * s + 8 * u) * 4);
                        }
                        float dz[4];
                        static_for<0, 4>([&](auto tc) {
                            MPG_CI(t, tc);
                            float z1 = c4[t] + a4[t];
                            if constexpr (NQ > 0) {   // (the forward's order of additions: the sign must be the forward's)
#pragma unroll
                                for (int q = 0; q < NQ; ++q) z1 = fmaf(esv[sd][q], q4[q][t], z1);
                            }
                            float gt = sel_by_bit<31>(__builtin_bit_cast(uint32_t, z1), valpha1[sd], vone1[sd]);
                            gt = drop_apply<DROP>(gt, wd, 16 * s + 8 * u + t, t, p.thr);
                            dz[t] = accC[mm][sd][8 * s + 4 * u + t] * gt;
                            dacc[mm][8 * s + 4 * u + t] += dz[t];
                            if constexpr (NQ > 0) {
#pragma unroll
                                for (int q = 0; q < NQ; ++q) {
                                    dsq[q] = fmaf(dz[t], q4[q][t], dsq[q]);
                                    dqacc[q][mm][8 * s + 4 * u + t] = fmaf(esv[sd][q], dz[t], dqacc[q][mm][8 * s + 4 * u + t]);
                                }
                            }
                        });
                        const float w0 = halve_add<0xB1>(lb0, dz[0], dz[1]), w1 = halve_add<0xB1>(lb0, dz[2], dz[3]);
                        ured[su] = halve_add<0x4E>(lb1, w0, w1);
                    });
                    const float x0 = halve_add<0x124>(lbb2, ured[0], ured[1]), x1 = halve_add<0x124>(lbb2, ured[2], ured[3]);
                    float y = halve_add<0x128>(lb3, x0, x1);
                    y += __shfl_xor(y, 16, 64);
                    if (dcown && (sd == 0 || has2)) dcj[32 * mm + dcslot] = y;
                });
                if constexpr (NQ > 0) {   // (wq sits in LDS times SC_A: undone here)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float tot = (dsq[q] + __shfl_xor(dsq[q], 32, 64)) * (1.f / SC_A);
                        if (oln < 32 && rb * 32 + oln < p.N && (sd == 0 || has2))
                            p.des[((size_t)(b * p.N + jj[sd]) * NQ + q) * p.N + rb * 32 + oln] = tot;
                    }
                }
            });
        }
        B2_STAMP(6);
    }

#ifdef MPG_B2STAMP
    if (blockIdx.x < 64 && lane == 0)
        for (int q = 0; q < 7; ++q) g_b2_stamps[(blockIdx.x * 4 + w) * 8 + q] = b2_st[q];
#endif
    // ---- da: reduce over the four waves (disjoint sender subsets)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int q = 0; q < T1; ++q)
#pragma unroll
        for (int k = 0; k < 16; ++k) red[((w * T1 + q) * 16 + k) * 64 + lane] = dacc[q][k];
    __syncthreads();
    float* out = p.da + ((size_t)sc * p.B + b) * p.N * H1;
    for (int e = tid; e < T1 * 16 * 64; e += 256) {
        const int ln = e & 63, k = (e >> 6) & 15, q = e >> 10;
        const float sum = red[e] + red[e + T1 * 1024] + red[e + 2 * T1 * 1024] + red[e + 3 * T1 * 1024];
        const int ii = rb * 32 + (ln & 31);
        const int f = 32 * q + 16 * (k >> 3) + 8 * ((k >> 2) & 1) + 4 * (ln >> 5) + (k & 3);
        if (ii < p.N) out[(size_t)ii * H1 + f] = sum;
    }
    if constexpr (NQ > 0) {   // daq [SC][B*N][NQ][96], the same reduction per scalar
        static_for<0, NQ>([&](auto qc) {
            MPG_CI(qq, qc);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < T1; ++q)
#pragma unroll
                for (int k = 0; k < 16; ++k) red[((w * T1 + q) * 16 + k) * 64 + lane] = dqacc[qq][q][k];
            __syncthreads();
            float* outq = p.daq + (((size_t)sc * p.B + b) * p.N * NQ + qq) * H1;
            for (int e = tid; e < T1 * 16 * 64; e += 256) {
                const int ln = e & 63, k = (e >> 6) & 15, q = e >> 10;
                const float sum = red[e] + red[e + T1 * 1024] + red[e + 2 * T1 * 1024] + red[e + 3 * T1 * 1024];
                const int ii = rb * 32 + (ln & 31);
                const int f = 32 * q + 16 * (k >> 3) + 8 * ((k >> 2) & 1) + 4 * (ln >> 5) + (k & 3);
                if (ii < p.N) outq[(size_t)ii * NQ * H1 + f] = sum;
            }
        });
    }
    if constexpr (EPI != 0) {
        // ---- epilogue chains on this jet's nodes.  The rows of da (just written above) and of dc (written sender by sender in
        //      the loop, zeros for masked senders in the prologue) are this workgroup's own stores: ordered within the workgroup.
        const int m0 = b * p.N + rb * 32, nrows = min(32, p.N - rb * 32);
        {
            const MpgChain& c = *cdxp;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            auto stage = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t s_lo, const uint32_t s_hi, const float ascale) {
                c2_stage_rows<false, 12, 0>(c, m0, nrows, smem, first_tile, bias_request, bias_store, s_lo, s_hi, ascale);
            };
            c2_body<false, 12, 0, 0, 0, 0, 1, EPI == 3>(c, m0, nrows, smem, smem + C2_FB, reinterpret_cast<float*>(smem + 2 * C2_FB), stage);
        }
        if constexpr (EPI != 3) {
            if (cnxp->nlayers > 0) {
                const MpgChain& c = *cnxp;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                auto stage = [&](auto&& first_tile, auto&& bias_request, auto&& bias_store, const uint32_t s_lo, const uint32_t s_hi, const float ascale) {
                    c2_stage_rows<false, 2, DROP>(c, m0, nrows, smem, first_tile, bias_request, bias_store, s_lo, s_hi, ascale);
                };
                c2_body<false, 2, 16, 16, DROP, 3, 0, EPI == 2>(c, m0, nrows, smem, smem + C2_FB, reinterpret_cast<float*>(smem + 2 * C2_FB), stage);
            }
        }
    }
}

template <int DROP, bool NEEDW, int NQ>
__global__ __launch_bounds__(256, 1) void edge_bwd_kernel(const MpgEdgeBwd p) { edge_bwd_body<DROP, NEEDW, NQ, 0>(p, nullptr, nullptr); }

template <int DROP, bool NEEDW, int EPI>
__global__ __launch_bounds__(256, 1) void edge_bwd_fn_kernel(const MpgEdgeBwd p, const MpgChain cdx, const MpgChain cnx) {
    edge_bwd_body<DROP, NEEDW, 0, EPI>(p, &cdx, &cnx);
}

// the epilogue forms of one dropout mode / NEEDW (edge_bwd_fn_*.hip: one translation unit each); epi = 1, 2, 3 as above
template <int D, bool NEEDW>
int b2_launch_fn(const MpgEdgeBwd* p, const MpgChain* cdx, const MpgChain* cnx, int epi, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB), block(256);
    MpgChain none = {};   // nlayers = 0: no second chain
    if (cnx == nullptr) cnx = &none;
#define MPG_B2FN(E)                                                                                        \
    do {                                                                                                   \
        MPG_ENSURE_LDS((edge_bwd_fn_kernel<D, NEEDW, E>), B2_LDS_BYTES);                                   \
        hipLaunchKernelGGL((edge_bwd_fn_kernel<D, NEEDW, E>), grid, block, B2_LDS_BYTES, st, *p, *cdx, *cnx); \
    } while (0)
    if (epi == 1) MPG_B2FN(1);
    else if (epi == 2) MPG_B2FN(2);
    else MPG_B2FN(3);
#undef MPG_B2FN
    return (int)hipGetLastError();
}

// the NEEDW pair of one dropout mode (the three modes compile as separate translation units: edge_bwd2.hip, edge_bwd2_d1.hip,
// edge_bwd2_d2.hip -- this template is slow to compile)
template <int D, int NQ = 0>
int b2_launch(const MpgEdgeBwd* p, hipStream_t st) {
    const int RB = (p->N + 31) / 32;
    dim3 grid(p->B * RB * p->SC), block(256);
    const bool needw = p->stageZ2 != nullptr;
    if (needw) {
        MPG_ENSURE_LDS((edge_bwd_kernel<D, true, NQ>), B2_LDS_BYTES);
        hipLaunchKernelGGL((edge_bwd_kernel<D, true, NQ>), grid, block, B2_LDS_BYTES, st, *p);
    } else {
        MPG_ENSURE_LDS((edge_bwd_kernel<D, false, NQ>), B2_LDS_BYTES);
        hipLaunchKernelGGL((edge_bwd_kernel<D, false, NQ>), grid, block, B2_LDS_BYTES, st, *p);
    }
    return (int)hipGetLastError();
}

}  // namespace
