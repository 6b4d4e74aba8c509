// Weight gradients of the fused edge network (see edge.hip for the forward, edge_bwd2_impl.h for the data-gradient kernel):
//
//   dW3 = s * sum_e dZ3 E2^T ;  dW2 = s * sum_e dZ2 E1^T ;  db3 = sum_e dZ3 ;  db2 = sum_e dZ2
//
// mpg_edge_bwd parks E2 and dZ2 in memory as the fp16 B fragments its lanes hold (one coalesced 16-byte store per lane
// and fragment; dZ2 in units of 2^-gexp of its (jet, receiver block)).  edge_dw_kernel streams those fragments into
// [receiver][feature] LDS images, rebuilds the cheap operands (E1 from a_i + c_j, dZ3 from dagg and the sign words),
// fetches MFMA operands with transposing LDS reads and accumulates dW3/dW2/db3/db2 in registers over its share of the
// edges; the per-workgroup partials are summed by a last small kernel that also undoes the fragment-order permutation of
// the feature indices and the launch's gradient unit.
//
// Arithmetic: ONE fp16 term per product -- every operand is a single fp16 value (2^-12 relative rounding), and every one
// of these roundings is independent from block to block: E2 and E1 differ from edge to edge by themselves, the parked
// dZ2 carries its block's dither factor (edge_bwd2_impl.h), and dZ3 -- for one receiver the same number for all senders
// up to one of two constants -- is built times that factor c while its partner E2 is divided by it.  Independent
// roundings average out over the edges a weight gradient sums (measured: tests/probe_precision.py).
// The contraction runs over receivers, senders AND jets, so everything is
// brought to ONE gradient unit 2^-eG, eG = min over the launch of gexp: dZ3 is built in it; a parked dZ2 piece stays as
// it is and its partner E1 is multiplied by 2^(eG - gexp) / c, c the block's dither factor (a jet whose gradients are
// 2^-24 of the largest jet's drops out of the fp16 range, and of any fp32 sum with that jet too).
#include "edge_common.h"
#include "reduce_group.h"
#include <stdlib.h>
#include <string.h>

#ifndef MPG_DW8_SETS
#define MPG_DW8_SETS 1   // register sets of parked pieces in the uniform kernel (edge_dw8_kernel)
#endif
#ifndef MPG_DW_EXP
#define MPG_DW_EXP 0  // experiment bits (tools/ubench/dw_bench.hip): 1 consumers idle, 2 builders idle, 4 no staged loads, 8 no LDS writes
#endif

namespace {

// The contraction runs over edges = (receiver, sender): per block (32 receivers of one jet, one sender) the
// four operand tensors are needed as [feature][receiver], but the backward (and any rebuild) naturally
// produces [receiver][8 features] pieces.  So the block's operands are laid down in LDS as plain
// [receiver][feature] fp16 images and the MFMA fragments are fetched with gfx950's transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads 4 receivers x 16 features and each lane receives one feature's
// 4 receivers).  Row strides are odd multiples of 64 B, which makes those reads bank-conflict free.
//
// A workgroup is 8 waves on 4 SIMDs: waves 0-3 are CONSUMERS (each owns 10-12 of the 45 output tiles in
// registers and only issues LDS reads + MFMAs), waves 4-7 are BUILDERS (VALU only: copy the parked dZ2, divide the
// parked E2 by the dither factor, rebuild dZ3 from dagg and the sign words and E1 from a_i + c_j, sum the biases).  The
// images are double buffered (2 x 40 KiB): builders fill block n+1 while consumers multiply block n, one barrier per block.  Every
// builder thread owns fixed (receiver, feature chunk) pieces and fetches everything it needs of a block two blocks ahead,
// into one of two register sets, right after the piece it replaces was used: no builder ever waits on another.
#ifdef MPG_DWSTAMP  // diagnostic build (tools/ubench/dw_bench.hip): clocks per section of a builder's block, summed over a workgroup's blocks
__device__ unsigned long long g_dw_stamps[64 * 4 * 8];
#define DW_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dw_acc[i] += t_ - dw_t; dw_t = t_; } while (0)
#else
#define DW_STAMP(i) do {} while (0)
#endif
constexpr int DW_RS3 = 448, DW_RS2 = 320, DW_RS1 = 192;  // image row strides (bytes)
constexpr int DW_Z3H = 0, DW_E2H = DW_Z3H + 32 * DW_RS3, DW_Z2H = DW_E2H + 32 * DW_RS2, DW_E1H = DW_Z2H + 32 * DW_RS2,
              DW_BUF = DW_E1H + 32 * DW_RS1;
constexpr int DW_LDS_BYTES = 2 * DW_BUF;  // 81,920
static_assert(DW_LDS_BYTES <= 163840, "dW images must fit the LDS twice");

struct DwTile { int prod, m, n; };  // prod 0: dW3 (A = Z3 tile m, B = E2 tile n); 1: dW2 (A = Z2, B = E1)
// consumer wave w owns tiles [0,12) [12,23) [23,34) [34,45)
__device__ constexpr DwTile DW_TILES[45] = {
    {0,0,0},{0,0,1},{0,0,2},{0,0,3},{0,0,4},{0,1,0},{0,1,1},{0,1,2},{0,1,3},{0,1,4},{1,0,0},{1,0,1},
    {0,2,0},{0,2,1},{0,2,2},{0,2,3},{0,2,4},{0,3,0},{0,3,1},{0,3,2},{0,3,3},{0,3,4},{1,0,2},
    {0,4,0},{0,4,1},{0,4,2},{0,4,3},{0,4,4},{0,5,0},{0,5,1},{0,5,2},{0,5,3},{0,5,4},{1,1,0},
    {1,1,1},{1,1,2},{1,2,0},{1,2,1},{1,2,2},{1,3,0},{1,3,1},{1,3,2},{1,4,0},{1,4,1},{1,4,2}};

// six consumers (edge_dw12_kernel): consumer w < 5 owns row w of dW3 (5 tiles) and row w of dW2 (3 tiles), consumer 5 row 5 of dW3:
// ranges [0,8) [8,16) [16,24) [24,32) [32,40) [40,45)
__device__ constexpr DwTile DW_TILES6[45] = {
    {0,0,0},{0,0,1},{0,0,2},{0,0,3},{0,0,4},{1,0,0},{1,0,1},{1,0,2},
    {0,1,0},{0,1,1},{0,1,2},{0,1,3},{0,1,4},{1,1,0},{1,1,1},{1,1,2},
    {0,2,0},{0,2,1},{0,2,2},{0,2,3},{0,2,4},{1,2,0},{1,2,1},{1,2,2},
    {0,3,0},{0,3,1},{0,3,2},{0,3,3},{0,3,4},{1,3,0},{1,3,1},{1,3,2},
    {0,4,0},{0,4,1},{0,4,2},{0,4,3},{0,4,4},{1,4,0},{1,4,1},{1,4,2},
    {0,5,0},{0,5,1},{0,5,2},{0,5,3},{0,5,4}};
template <int TBL> constexpr DwTile dw_tile(int t) { return TBL ? DW_TILES6[t] : DW_TILES[t]; }

template <int TBL> constexpr bool dw_same_rows(int t, int u) { return dw_tile<TBL>(t).prod == dw_tile<TBL>(u).prod && dw_tile<TBL>(t).m == dw_tile<TBL>(u).m; }
template <int TBL> constexpr bool dw_leader(int t, int begin) { return t == begin || !dw_same_rows<TBL>(t, t - 1); }
template <int TBL> constexpr int dw_group_end(int t, int end) {
    int e = t + 1;
    while (e < end && dw_same_rows<TBL>(t, e)) ++e;
    return e;
}

MPG_DEV int feat_of_fi(int fi) {  // fragment-order index -> feature
    const int ms = fi >> 4, hh = (fi >> 3) & 1, jj = fi & 7;
    return 32 * (ms >> 1) + 16 * (ms & 1) + 8 * (jj >> 2) + 4 * hh + (jj & 3);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
// fragment (32 features starting at byte column `col`, receivers 16s .. 16s+15) of an image: two transposed reads
MPG_DEV f16x8 dw_frag(uint32_t lane_addr, int off, int rs) {
    typedef __attribute__((address_space(3))) s16x4* P;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P)(uintptr_t)(lane_addr + (uint32_t)off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((P)(uintptr_t)(lane_addr + (uint32_t)(off + 4 * rs)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}

template <int TBL> constexpr bool dw_has_prod(int prod, int begin, int end) {
    for (int t = begin; t < end; ++t) if (dw_tile<TBL>(t).prod == prod) return true;
    return false;
}
template <int TBL> constexpr bool dw_uses_n(int prod, int n, int begin, int end) {
    for (int t = begin; t < end; ++t) if (dw_tile<TBL>(t).prod == prod && dw_tile<TBL>(t).n == n) return true;
    return false;
}

// One block into the wave's accumulators: per k-step (16 receivers) the B fragments the wave's tiles of a product need
// are read ONCE (E2: all five, E1: up to three), then every row group (same product and m) reads its A fragment and
// issues its MFMAs.  The transposing reads are what bounds this kernel (LDS bandwidth), so no fragment is read twice.
template <int BEGIN, int END, int TBL = 0>
MPG_DEV void dw_consume(f32x16* acc, uint32_t buf, int lane) {
    // lane 4q+p of 16-lane group g supplies row (8 (g>>1) + q), feature columns 16 (g&1) + 4p .. +3 of the block
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = 8 * (g >> 1) + q, col = (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // fresh (opaque) bases per k-step: everything else is an immediate offset, and no fragment of this k-step can be
        // kept for the next (the accumulators would spill)
        uint32_t bz3 = buf + DW_Z3H + row * DW_RS3 + col, be2 = buf + DW_E2H + row * DW_RS2 + col;
        uint32_t bz2 = buf + DW_Z2H + row * DW_RS2 + col, be1 = buf + DW_E1H + row * DW_RS1 + col;
        asm volatile("" : "+v"(bz3), "+v"(be2), "+v"(bz2), "+v"(be1));
        static_for<0, 2>([&](auto pc) {
            MPG_CI(prod, pc);
            if constexpr (dw_has_prod<TBL>(prod, BEGIN, END)) {
                constexpr int rsa = prod == 0 ? DW_RS3 : DW_RS2, rsb = prod == 0 ? DW_RS2 : DW_RS1, nb = prod == 0 ? T2 : T1;
                const uint32_t ba = prod == 0 ? bz3 : bz2, bb = prod == 0 ? be2 : be1;
                f16x8 bfr[nb];
                static_for<0, nb>([&](auto nc) {
                    MPG_CI(n, nc);
                    if constexpr (dw_uses_n<TBL>(prod, n, BEGIN, END)) bfr[n] = dw_frag(bb, 16 * s * rsb + 64 * n, rsb);
                });
                static_for<BEGIN, END>([&](auto tc) {
                    MPG_CI(t, tc);
                    if constexpr (dw_tile<TBL>(t).prod == prod && dw_leader<TBL>(t, BEGIN)) {
                        constexpr int ge = dw_group_end<TBL>(t, END);
                        const f16x8 a = dw_frag(ba, 16 * s * rsa + 64 * dw_tile<TBL>(t).m, rsa);
                        static_for<t, ge>([&](auto uc) {
                            MPG_CI(u, uc);
                            acc[u - BEGIN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfr[dw_tile<TBL>(u).n], acc[u - BEGIN], 0, 0, 0);
                        });
                    }
                });
            }
        });
    }
}

template <int BEGIN, int END, int TBL = 0>
MPG_DEV void dw_store(const f32x16* acc, float* part3, float* part2, int lane) {
    const int cc = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = BEGIN; t < END; ++t) {
        const DwTile d = dw_tile<TBL>(t);
        float* dst = d.prod == 0 ? part3 : part2;
        const int ncol = d.prod == 0 ? H2 : H1;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = 32 * d.m + 8 * (k >> 2) + 4 * hh + (k & 3);
            dst[(size_t)row * ncol + 32 * d.n + cc] = acc[t - BEGIN][k];
        }
    }
}

// A workgroup walks at most 64 blocks (the launcher sizes the grid for that); their valid-sender bits are one
// ballot taken at kernel start, so stepping to the next unmasked block is pure scalar arithmetic -- no memory
// access and no loop inside the pipelined loops (either would make the compiler drain vmcnt there).
// The blocks of a workgroup are RUNS of R consecutive senders of one (jet, receiver block), the runs strided over the
// launch: run q of workgroup g is run g + q * gridDim.x, i.e. blocks (g + q * gridDim.x) * R .. + R - 1.  A contiguous
// range would be one jet's senders, and a launch would last as long as its fullest jet (masks sorted to the end of a
// jet left whole workgroups idle at N = 150); strided run by run, every workgroup samples several jets at all positions
// of the sender axis.  Within a run the receivers' rows of dagg and a stay in registers: fetched per block they were
// 36 KB of the 60 KB a block pulls through the CU's 64 B/clk texture path -- more than the parked pieces themselves.
// R divides N (the launcher picks it), so a run never straddles two receiver blocks.
MPG_DEV int dw_block(int t, int R) { return ((int)blockIdx.x + (t / R) * (int)gridDim.x) * R + t % R; }
MPG_DEV unsigned long long dw_valid_bits(const MpgEdgeDw& p, int R, int blk0, int blk1) {   // over the slots [blk0, blk1)
    const int RB = (p.N + 31) / 32, t = blk0 + (int)(threadIdx.x & 63), x = dw_block(t, R);
    bool ok = t < blk1;
    if (ok && p.mask != nullptr) ok = p.mask[((x / p.N) / RB) * p.N + x % p.N] != 0.f;
    return __ballot(ok);
}
MPG_DEV int dw_next_valid(unsigned long long bits, int blk0, int blk, int blk1) {
    const int d = blk - blk0;
    const unsigned long long rem = d < 64 ? bits >> d : 0ull;
    return rem ? blk + __builtin_ctzll(rem) : blk1;
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load
// (vmcnt(0)), which would expose the latency of the staged pieces requested a block ahead.
MPG_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The per-workgroup loop of one consumer wave (its output tiles BEGIN..END of DW_TILES).  Instantiated per
// role: a run-time branch around the MFMA section would make the accumulators merge at every join.
template <int BEGIN, int END, int NQ, int TBL = 0>
MPG_DEV void dw_consumer(const MpgEdgeDw& p, int blk0, int blk1, unsigned long long vbits) {
    const int lane = threadIdx.x & 63;
    if constexpr (NQ > 0) lds_barrier();   // (the builders lay down the edge-scalar columns behind this one)
    f32x16 acc[END - BEGIN];
#pragma unroll
    for (int t = 0; t < END - BEGIN; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    int cur = dw_next_valid(vbits, blk0, blk0, blk1), it = 0;
    lds_barrier();  // block `cur` is in buffer 0
    while (cur < blk1) {
        if (!(MPG_DW_EXP & 1)) dw_consume<BEGIN, END, TBL>(acc, lds0 + (it & 1) * DW_BUF, lane);
        lds_barrier();
        cur = dw_next_valid(vbits, blk0, cur + 1, blk1);
        ++it;
    }
    float* part = p.part + (size_t)blockIdx.x * (H3 * H2 + H2 * H1 + H3 + H2);
    dw_store<BEGIN, END, TBL>(acc, part, part + H3 * H2, lane);
    lds_barrier();   // (the builders exchange their bias sums through LDS behind this one)
}

// keep bits (bit k = element k) of one 8-feature chunk: features 32 tile + f0 + {0..3, 8..11} of edge row `erow`
template <int DM>
MPG_DEV uint32_t dw_chunk_keep(uint32_t seed_lo, uint32_t seed_hi, uint32_t tag, uint32_t erow, int tile, int f0, uint32_t thr) {
    if constexpr (DM == 2) {
        const uint32_t w = drop_word(seed_lo, seed_hi, tag, erow, DROP_BIT_GRP + (uint32_t)tile) >> f0;
        return (w & 0xfu) | ((w >> 4) & 0xf0u);  // bits f0..f0+3 and f0+8..f0+11
    } else if constexpr (DM == 1) {
        uint32_t m = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t w = drop_word(seed_lo, seed_hi, tag, erow, (uint32_t)(8 * tile + (f0 >> 2) + 2 * u));
#pragma unroll
            for (int t = 0; t < 4; ++t) m |= (drop_keep(w, t, thr) ? 1u : 0u) << (4 * u + t);
        }
        return m;
    } else {
        return 0xffu;
    }
}

// min over the launch of the gradient-unit exponents: the launch's unit (every wave that needs it computes it itself)
MPG_DEV int dw_launch_exp(const MpgEdgeDw& p) {
    const int n = p.B * ((p.N + 31) / 32);
    int e = 0x7fffffff;
    for (int t = (int)(threadIdx.x & 63); t < n; t += 64) e = min(e, p.gexp[t]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) e = min(e, __shfl_xor(e, o, 64));
    return __builtin_amdgcn_readfirstlane(e);
}

// NQ: edge scalars (0 or MPG_EDGE_SCALARS): E1 = lrelu(a_i + c_j + sum_q es(i, j, q) wq[q]), as the forward adds it up
// NB builder waves behind NC consumer waves: 4 + 4 (edge_dw_kernel) or 6 + 6 (edge_dw12_kernel).  A receiver's pieces are dealt to
// CS = 2 NB chunk groups: thread (r, cg) owns chunks cg, cg + CS, ... of each tensor -- 3 | 3 | 2 (dZ3 | the 160-feature ones | E1)
// per block with four waves, 2 | 2 | 1 with six.
template <int DROP, int NQ, int NB = 4, int NC = 4>
MPG_DEV void dw_builder(const MpgEdgeDw& p, int R, char* smem, int blk0, int blk1, unsigned long long vbits) {
    constexpr int CS = 2 * NB, CQ = CS / 4, NZ3 = 24 / CS, N160 = (20 + CS - 1) / CS, NE1 = (12 + CS - 1) / CS;
    static_assert(24 % CS == 0 && CS % 4 == 0, "chunk groups of whole quads, dZ3's 24 chunks dealt evenly");
    const int bt = threadIdx.x - 64 * NC;     // builder thread 0 .. 64 NB - 1
    if constexpr (NQ > 0) {
        float* lwq = reinterpret_cast<float*>(smem + DW_LDS_BYTES);
        if (bt < NQ * H1) lwq[bt] = p.wq[bt];
        lds_barrier();
    }
    // receiver row r of the images and chunk group cg (chunks cg, cg + 8, cg + 16).  Four ADJACENT lanes hold the four chunk
    // groups of one receiver that make up a whole 32-feature tile: their two 16-byte reads of a dagg / a row cover one
    // contiguous 128-byte line, so a wave's load touches 16 rows x 1 line.  (With the receiver in the low lane bits every
    // lane read its own row -- 64 lines per load instruction, and the texture path, not HBM, was what the builders waited for.)
    const int r = (bt >> 2) & 31, cg = ((bt >> 7) << 2) | (bt & 3);
    const int RB = (p.N + 31) / 32;
    // chunk c = 2 frag + h holds fragment-order features 8c .. 8c+7 = registers 8s .. 8s+7 of tile (c >> 2) of
    // lane (r, h):  s = (c >> 1) & 1 and h = c & 1 are the same for all chunks of this thread
    const int cs = (cg >> 1) & 1, ch = cg & 1;
    const int f0 = 16 * cs + 4 * ch;  // features of chunk c: 32 (c >> 2) + f0 + {0..3, 8..11}

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    float db3[NZ3][8], db2[N160][8];
#pragma unroll
    for (int n = 0; n < NZ3; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) db3[n][k] = 0.f;
#pragma unroll
    for (int n = 0; n < N160; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) db2[n][k] = 0.f;
    // Everything of a block is fetched ahead, RAW: nothing may be computed from a prefetched value before the block that
    // needs it (a use right behind the load would make every iteration wait for its youngest load, i.e. drain the whole
    // prefetch queue).  The parked pieces come from HBM -- a round trip under load is longer than one block's build -- and
    // are requested TWO blocks ahead, into two register sets (block k lives in set k & 1); the rest is L2-resident and
    // requested one block ahead.  Vector memory operations complete in issue order, so within a build the one-ahead
    // requests are issued before the two-ahead ones: the next build waits for nothing younger than what it needs.
    struct Small {
        float dreg[NZ3][8], areg[NE1][8];   // dagg / a of this thread's Z3 / E1 chunks (raw)
        float dscl;                     // agg_scale * dscale * 2^eG, or 0 for a padding receiver
        uint32_t sw[NZ3];               // sign words of the Z3 chunks' lanes
        unsigned int nbw;               // this receiver's neighbour word holding the block's sender (k-NN graphs)
        float4 cv[NE1][2];              // c_j of the E1 chunks
        float esq[NQ > 0 ? NQ : 1];     // edge scalars of (this receiver, the block's sender)
    };
    struct Big {
        f16x8 eh[N160];                 // parked E2 pieces
        f16x8 zh[N160];                 // parked dZ2 pieces (in their block's gradient unit)
    };
    Small S;
    Big B0, B1;
    // gradient units: the launch's exponent and, in lane t, the exponent of slot t's (jet, receiver block)
    const int eG = dw_launch_exp(p);
    const float unitG = __builtin_bit_cast(float, (uint32_t)(eG + 127) << 23);   // 2^eG (|eG| <= 100)
    int eslot = eG;
    {
        const int t = blk0 + (int)(threadIdx.x & 63);
        if (t < blk1) eslot = p.gexp[dw_block(t, R) / p.N];
    }
    // (a value defined HERE, not by a load: the pipelined loop below reads it with v_readlane, and for a load result the
    // compiler would wait there with vmcnt(0) -- it cannot tell how old a load is across the loop's back edge -- and
    // drain every prefetch of every block)
    asm volatile("" : "+v"(eslot));
    // The workgroup's slots (at most 64) -> block, sender, jet, receiver block, ONE slot per lane: a block's indices are then
    // four v_readlane instead of the scalar divisions by N, RB and R they came from -- six division sequences per block for
    // the block, the one-ahead and the two-ahead requests, a third of everything a builder wave issued.
    int sblk, sj, sb, srb;
    {
        const int t = min(blk0 + (int)(threadIdx.x & 63), max(blk1 - 1, blk0));
        sblk = dw_block(t, R);
        const int brb = sblk / p.N;
        sj = sblk - brb * p.N;
        sb = brb / RB;
        srb = brb - sb * RB;
    }
    asm volatile("" : "+v"(sblk), "+v"(sj), "+v"(sb), "+v"(srb));
    struct Idx { int blk, j, b, rb; };
    auto idx_of = [&](int slot) {
        const int l = slot - blk0;
        return Idx{__builtin_amdgcn_readlane(sblk, l), __builtin_amdgcn_readlane(sj, l), __builtin_amdgcn_readlane(sb, l),
                   __builtin_amdgcn_readlane(srb, l)};
    };

    // piece n of the 160-feature tensors is chunk cg + CS n while that is one of their 20 chunks, E1 chunk n lies in tile (cg >> 2) + CQ n
    // while that is one of its 3 tiles; a thread without an n-th piece redoes its previous one (see chunk160 below)
    auto valid160 = [&](int n) { return cg + CS * n < 20; };
    auto validE1 = [&](int n) { return (cg >> 2) + CQ * n < 3; };
    auto tileZ3 = [&](int n) { return CQ * n + (cg >> 2); };   // 0..5
    auto e1tile = [&](int n) { return validE1(n) ? (cg >> 2) + CQ * n : (cg >> 2) + CQ * (n - 1); };  // 0..2
    // p = 1/2 (one keep BIT per element, one hashed word per (edge row, 32-feature tile)): the four adjacent lanes that hold one
    // receiver's chunk groups need the SAME five words of a block -- the dZ3 tiles 2n + (cg >> 2), n = 0..2, and the E1 tiles
    // e1tile(0), e1tile(1) -- and used to hash all five each.  Now lane q of the quad hashes word q (one instruction sequence for
    // four different words: the site and tile ride in a per-lane constant), everyone word four, and a DPP quad broadcast hands
    // them round: two hash sequences + five moves per block instead of five sequences (a tenth of what a builder wave issues; measured
    // in tools/ubench/dw_bench.hip: 115.5 -> 113.6 us at 512 jets, 70.4 -> 69.8 at 256 -- the builders are not bound by what they
    // issue alone: the sections that request memory take 1.0k of a block's 3.6k clk for ~100 instructions).
    // (the block's words: dZ3 tiles n = 0 .. NZ3 - 1, then E1 tiles n = 0 .. NE1 - 1 -- five with four builder waves, three with six;
    // lane q of a quad hashes word q, a fifth word everyone)
    uint32_t hcA = 0, hcB = 0;   // (grp + tag * 0x10001) * 0x85EBCA77 + seed_hi of drop_word, for this lane's word / the fifth
    if constexpr (DROP == 2) {
        const int ql = min(bt & 3, NZ3 + NE1 - 1);
        const uint32_t tagA = p.tag_base + (ql < NZ3 ? TAG_E2 : TAG_E0);
        const uint32_t grpA = DROP_BIT_GRP + (uint32_t)(ql < NZ3 ? tileZ3(ql) : e1tile(ql - NZ3));
        hcA = (grpA + tagA * 0x10001u) * 0x85EBCA77u + seed_hi;
        if constexpr (NZ3 + NE1 > 4)
            hcB = (DROP_BIT_GRP + (uint32_t)e1tile(4 - NZ3) + (p.tag_base + TAG_E0) * 0x10001u) * 0x85EBCA77u + seed_hi;
    }
    // Threads of chunk groups 4..7 have no third piece of the 160-feature tensors (and no second E1 chunk): they
    // redo their previous piece instead (same data to the same place), which keeps the whole build free of
    // branches -- inside a branch the compiler waits for ALL outstanding loads, i.e. for the prefetches too.
    auto chunk160 = [&](int n) { return valid160(n) ? cg + CS * n : cg + CS * (n - 1); };

    // Every global read is a raw buffer load: resource in SGPRs, block-dependent part as scalar offset, one
    // thread-constant VGPR offset per stream (plain pointers cost two VGPRs of address per load in flight).
    const int nblk = p.B * RB * p.N;
    const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageE2), 0, nblk * (NFR2 * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageZ2), 0, nblk * (NFR2 * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(p.sign3), 0, nblk * (T3 * 32 * 4), 0x00020000);
    const int ldac = p.ld_ac ? p.ld_ac : H1;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c), 0, p.B * p.N * ldac * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.B * p.N * ldac * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dagg), 0, p.B * p.N * p.ld_dagg * 4, 0x00020000);
    int vo160[N160];  // byte offset of this thread's piece n inside a 10 KiB block
#pragma unroll
    for (int n = 0; n < N160; ++n) vo160[n] = (chunk160(n) * 32 + r) * 16;
    int voS[NZ3];     // byte offset of the sign word of dZ3 chunk n (word tile >> 1 of lane (r, h)) inside a block's 768 B
#pragma unroll
    for (int n = 0; n < NZ3; ++n) voS[n] = (32 * ch + r) * 4 + (tileZ3(n) >> 1) * 256;
    int voE1[NE1];    // byte offset of E1 chunk n's first feature inside a 96-float row of a / c
#pragma unroll
    for (int n = 0; n < NE1; ++n) voE1[n] = (32 * e1tile(n) + f0) * 4;
    auto ldb4 = [&](__amdgpu_buffer_rsrc_t rs, int vo, int so) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
    };

    // dagg / a of the (jet, receiver block) of block `blk`.  Padding receivers read row 0 and get scale 0.
    auto load_d = [&](Small& P, const Idx& X) {
        const int rb = X.rb, b = X.b, ii = rb * 32 + r;
        const bool ok = ii < p.N;
        P.dscl = ok ? p.agg_scale * p.dscale * unitG : 0.f;
        const int rowD = (ok ? ii : 0) * p.ld_dagg * 4 + (32 * (cg >> 2) + f0) * 4, soD = b * p.N * p.ld_dagg * 4;
#pragma unroll
        for (int n = 0; n < NZ3; ++n) {
            const float4 u = ldb4(rD, rowD + 32 * CS * n, soD), v = ldb4(rD, rowD + 32 * CS * n + 32, soD);
            P.dreg[n][0] = u.x; P.dreg[n][1] = u.y; P.dreg[n][2] = u.z; P.dreg[n][3] = u.w;
            P.dreg[n][4] = v.x; P.dreg[n][5] = v.y; P.dreg[n][6] = v.z; P.dreg[n][7] = v.w;
        }
    };
    auto load_a = [&](Small& P, const Idx& X) {
        const int rb = X.rb, b = X.b, ii = rb * 32 + r;
        const bool ok = ii < p.N;
        const int rowA = (ok ? ii : 0) * ldac * 4, soA = b * p.N * ldac * 4;
#pragma unroll
        for (int n = 0; n < NE1; ++n) {
            const float4 u = ldb4(rA, rowA + voE1[n], soA), v = ldb4(rA, rowA + voE1[n] + 32, soA);
            P.areg[n][0] = u.x; P.areg[n][1] = u.y; P.areg[n][2] = u.z; P.areg[n][3] = u.w;
            P.areg[n][4] = v.x; P.areg[n][5] = v.y; P.areg[n][6] = v.z; P.areg[n][7] = v.w;
        }
    };
    const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(p.nbr), 0, p.nbr ? p.B * p.N * ((p.N + 31) >> 5) * 4 : 0, 0x00020000);
    auto load_nb = [&](Small& P, const Idx& X) {  // (with no graph the resource is empty: the load returns 0 and is ignored below)
        const int j = X.j, rb = X.rb, b = X.b, ii = rb * 32 + r;
        P.nbw = __builtin_amdgcn_raw_buffer_load_b32(rN, ((b * p.N + (ii < p.N ? ii : 0)) * ((p.N + 31) >> 5) + (j >> 5)) * 4, 0, 0);
    };
    // (requested ONE block ahead like the other small things, in front of the two-ahead pieces: asked for two blocks ahead,
    // behind them, the top of a build waited for them across the loop's back edge with vmcnt(0) -- 60 -> 67 us)
    auto load_sw = [&](Small& P, const Idx& X) {  // word (tile >> 1) = n of lane (r, h)
#pragma unroll
        for (int n = 0; n < NZ3; ++n) P.sw[n] = __builtin_amdgcn_raw_buffer_load_b32(rS, voS[n], X.blk * (T3 * 32 * 4), 0);
    };
    const __amdgpu_buffer_rsrc_t rES = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.es), 0, NQ > 0 ? p.B * p.N * NQ * p.N * 4 : 0, 0x00020000);
    auto load_c = [&](Small& P, const Idx& X) {
        const int j = X.j, b = X.b, so = (b * p.N + j) * ldac * 4;
#pragma unroll
        for (int n = 0; n < NE1; ++n) { P.cv[n][0] = ldb4(rC, voE1[n], so); P.cv[n][1] = ldb4(rC, voE1[n] + 32, so); }
        if constexpr (NQ > 0) {
            const int rb = X.rb, ii = rb * 32 + r;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                P.esq[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rES, (ii < p.N ? ii : 0) * 4, ((b * p.N + j) * NQ + q) * p.N * 4, 0));
        }
    };
    // parked pieces: chunk c of receiver r is element c * 32 + r of a block of 640 16-byte pieces
    auto load_e2 = [&](Big& P, int blk, int n) {
        P.eh[n] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rE, vo160[n], blk * (NFR2 * 1024), 0));
    };
    auto load_z2 = [&](Big& P, int blk, int n) {
        P.zh[n] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rZ, vo160[n], blk * (NFR2 * 1024), 0));
    };

    auto load_small = [&](Small& P, const Idx& X) { load_d(P, X); load_sw(P, X); load_nb(P, X); load_c(P, X); load_a(P, X); };
    auto load_big = [&](Big& Q, int blk) {
#pragma unroll
        for (int n = 0; n < N160; ++n) { load_z2(Q, blk, n); load_e2(Q, blk, n); }
    };

#ifdef MPG_DWSTAMP
    unsigned long long dw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dw_t = __builtin_amdgcn_s_memtime();
#endif
    // build the block of slot `slot` (small things in S, parked pieces in Q) into buffer `buf`.  Right after a piece is used
    // its successor is requested: the small things of slot `pre1` first, the parked pieces of slot `pre2` behind them
    auto build = [&](int slot, char* buf, int pre1_slot, int pre2_slot, Big& Q) {
        const bool exp_noload = (MPG_DW_EXP & 4) && p.N != 12345, exp_nowrite = (MPG_DW_EXP & 8) && p.N != 12345;
        const Idx X = idx_of(slot), X1 = idx_of((MPG_DW_EXP & 16) ? blk0 : pre1_slot);
        const int blk = X.blk, pre2 = __builtin_amdgcn_readlane(sblk, ((MPG_DW_EXP & 16) ? blk0 : pre2_slot) - blk0);
        // the next block belongs to other receivers: fetch their rows (a run never straddles two receiver blocks; two runs of
        // the same receivers would fetch the same rows)
        const bool newrun = X1.b != X.b || X1.rb != X.rb;
        const int j = X.j, rb = X.rb, b = X.b, ii = rb * 32 + r;
        const uint32_t erow = (uint32_t)((b * p.N + ii) * p.N + j);
        // the block's units: its gradient unit is 2^-e / c (c: the block's dither factor), the launch's 2^-eG
        const int de = eG - __builtin_amdgcn_readlane(eslot, slot - blk0);
        const float dth = dither_of((uint32_t)blk), rdth = __builtin_amdgcn_rcpf(dth);
        const float funit = __builtin_bit_cast(float, (uint32_t)max(de + 127, 0) << 23) * rdth;
        const _Float16 rch = (_Float16)rdth;
        // dZ3 = dagg * slope(sign bit) * keep3, in the launch's gradient unit times the block's dither factor (dZ3 of a
        // receiver is the same number for all its senders up to one of two constants: see edge_bwd2_impl.h); the bias
        // sums take it without the factor
        const float in_set = (p.nbr == nullptr || ((S.nbw >> (j & 31)) & 1u)) ? 1.f : 0.f;
        const float dscl_1 = S.dscl * in_set * dth, dscl_a = dscl_1 * p.alpha;   // (the dither factor rides in the slope constants)
        uint32_t kz3[3] = {0xffu, 0xffu, 0xffu}, ke1[2] = {0xffu, 0xffu};   // keep bits of the block's chunks (NZ3 | NE1 of them used)
        if constexpr (DROP == 2) {
            const uint32_t x0 = (erow + seed_lo) * 0x9E3779B1u;
            auto fin = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
            const uint32_t wA = fin(x0 ^ hcA), wB = fin(x0 ^ hcB);
            auto bits = [&](uint32_t w) { w >>= f0; return (w & 0xfu) | ((w >> 4) & 0xf0u); };   // bits f0..f0+3 and f0+8..f0+11
            const uint32_t q0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wA, 0x00, 0xf, 0xf, false);   // quad_perm [0,0,0,0]
            const uint32_t q1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wA, 0x55, 0xf, 0xf, false);   // [1,1,1,1]
            const uint32_t q2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wA, 0xAA, 0xf, 0xf, false);   // [2,2,2,2]
            const uint32_t q3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wA, 0xFF, 0xf, 0xf, false);   // [3,3,3,3]
            if constexpr (NZ3 == 3) { kz3[0] = bits(q0); kz3[1] = bits(q1); kz3[2] = bits(q2); ke1[0] = bits(q3); ke1[1] = bits(wB); }
            else { kz3[0] = bits(q0); kz3[1] = bits(q1); ke1[0] = bits(q2); (void)q3; (void)wB; }
        } else if constexpr (DROP == 1) {
#pragma unroll
            for (int n = 0; n < NZ3; ++n) kz3[n] = dw_chunk_keep<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, tileZ3(n), f0, p.thr);
#pragma unroll
            for (int n = 0; n < NE1; ++n) ke1[n] = dw_chunk_keep<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, e1tile(n), f0, p.thr);
        }
        DW_STAMP(0);   // block setup (index arithmetic, units, keep words) -- and whatever the barrier before it cost
#pragma unroll
        for (int n = 0; n < NZ3; ++n) {
            const int m = tileZ3(n), c = cg + CS * n;
            float v[8];
            const uint32_t keep = kz3[n];
            // sign bit of element k: bit 31 - (16 (m & 1) + 8 cs + k) of the lane's word -- shifted once so that the bit index
            // is a compile-time constant (v_bfe + v_bfi instead of shift, and, compare, select)
            const uint32_t swn = S.sw[n] << (16 * (m & 1) + 8 * cs);
            // (packed fp32 arithmetic -- v_pk_mul_f32 / v_pk_fma_f32, two elements per instruction -- was measured here: no
            // faster; a packed instruction takes the two issue turns of the scalar ones it replaces)
            static_for<0, 8>([&](auto kc) {
                MPG_CI(k, kc);
                float x = S.dreg[n][k] * sel_by_bit<31 - k>(swn, dscl_a, dscl_1);
                if (DROP && !((keep >> k) & 1u)) x = 0.f;
                v[k] = x;
                db3[n][k] = fmaf(x, rdth, db3[n][k]);
            });
            const f16x8 hh = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
            if (!exp_nowrite) *reinterpret_cast<f16x8*>(buf + DW_Z3H + r * DW_RS3 + c * 16) = hh;
            else if (hh[0] == (_Float16)123.f) db2[0][0] += (float)hh[1];
        }
        DW_STAMP(1);   // dZ3
        if (!exp_noload) { load_sw(S, X1); load_nb(S, X1); if (newrun) load_d(S, X1); }
        // E1 = keep1 * lrelu(a_i + c_j), times funit: what takes the parked dZ2 -- which goes into its image AS PARKED,
        // rounded once, by mpg_edge_bwd -- to the launch's unit multiplies the OTHER operand of its product, built in
        // fp32 anyway (chunk groups 4..7: the second chunk repeats the first)
#pragma unroll
        for (int n = 0; n < NE1; ++n) {
            const int q = e1tile(n), c = 4 * q + (cg & 3);
            const float cc[8] = {S.cv[n][0].x, S.cv[n][0].y, S.cv[n][0].z, S.cv[n][0].w, S.cv[n][1].x, S.cv[n][1].y, S.cv[n][1].z, S.cv[n][1].w};
            float v[8];
            const uint32_t keep = ke1[n];
            float wqv[NQ > 0 ? NQ : 1][8];
            if constexpr (NQ > 0) {
#pragma unroll
                for (int qq = 0; qq < NQ; ++qq) {
                    const float4* lw = reinterpret_cast<const float4*>(smem + DW_LDS_BYTES + (qq * H1 + 32 * q + f0) * 4);
                    const float4 u0 = lw[0], u1 = lw[2];
                    wqv[qq][0] = u0.x; wqv[qq][1] = u0.y; wqv[qq][2] = u0.z; wqv[qq][3] = u0.w;
                    wqv[qq][4] = u1.x; wqv[qq][5] = u1.y; wqv[qq][6] = u1.z; wqv[qq][7] = u1.w;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float z1 = S.areg[n][k] + cc[k];
                if constexpr (NQ > 0) {
#pragma unroll
                    for (int qq = 0; qq < NQ; ++qq) z1 = fmaf(S.esq[qq], wqv[qq][k], z1);
                }
                float x = lrelu(z1, p.alpha) * funit;
                if (DROP && !((keep >> k) & 1u)) x = 0.f;
                v[k] = x;
            }
            const f16x8 hh = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
            if (!exp_nowrite) *reinterpret_cast<f16x8*>(buf + DW_E1H + r * DW_RS1 + c * 16) = hh;
            else if (hh[0] == (_Float16)123.f) db2[0][0] += (float)hh[1];
        }
        DW_STAMP(2);   // requests + E1
        if (!exp_noload) { load_c(S, X1); if (newrun) load_a(S, X1); }
        // dZ2: as parked; the bias sums (fp32) in the launch's unit
#pragma unroll
        for (int n = 0; n < N160; ++n) {
            const int c = chunk160(n);
            if (!exp_nowrite) *reinterpret_cast<f16x8*>(buf + DW_Z2H + r * DW_RS2 + c * 16) = Q.zh[n];
            const float take = valid160(n) ? funit : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) db2[n][k] += take * (float)Q.zh[n][k];
            if (!exp_noload) load_z2(Q, pre2, n);
        }
        DW_STAMP(3);   // requests + dZ2
        // E2: as parked, divided by the block's dither factor (its partner dZ3 was built times that factor)
        const f16x8 rc8 = {rch, rch, rch, rch, rch, rch, rch, rch};
#pragma unroll
        for (int n = 0; n < N160; ++n) {
            const int c = chunk160(n);
            const f16x8 e = Q.eh[n] * rc8;
            if (!exp_nowrite) *reinterpret_cast<f16x8*>(buf + DW_E2H + r * DW_RS2 + c * 16) = e;
            else if (e[0] == (_Float16)123.f) db2[0][0] += (float)e[1];
            if (!exp_noload) load_e2(Q, pre2, n);
        }
        DW_STAMP(4);   // E2
    };

    // valid slots v0, v1, v2, ...: block v_k takes its parked pieces from set k & 1, which is then refilled with those of
    // v_{k+2}, and refills the small things with those of v_{k+1}.  Past the end the prefetches are clamped to the
    // range's last slot: they fetch that block again, unused.
    int cur = dw_next_valid(vbits, blk0, blk0, blk1), it = 0;
    int n1 = dw_next_valid(vbits, blk0, cur + 1, blk1), n2 = dw_next_valid(vbits, blk0, n1 + 1, blk1);
    if (cur < blk1) {
        load_small(S, idx_of(cur));
        load_big(B0, idx_of(cur).blk);
        load_big(B1, idx_of(min(n1, blk1 - 1)).blk);
        build(cur, smem, min(n1, blk1 - 1), min(n2, blk1 - 1), B0);
    }
    lds_barrier();
    while (cur < blk1) {
        {   // odd blocks: set 1
            const int n3 = dw_next_valid(vbits, blk0, n2 + 1, blk1);
            if (n1 < blk1 && !(MPG_DW_EXP & 2)) build(n1, smem + ((it + 1) & 1) * DW_BUF, min(n2, blk1 - 1), min(n3, blk1 - 1), B1);
            lds_barrier();
            DW_STAMP(5);   // waiting at the barrier (for the consumers)
            cur = n1; n1 = n2; n2 = n3; ++it;
        }
        if (!(cur < blk1)) break;
        {   // even blocks: set 0
            const int n3 = dw_next_valid(vbits, blk0, n2 + 1, blk1);
            if (n1 < blk1 && !(MPG_DW_EXP & 2)) build(n1, smem + ((it + 1) & 1) * DW_BUF, min(n2, blk1 - 1), min(n3, blk1 - 1), B0);
            lds_barrier();
            DW_STAMP(5);
            cur = n1; n1 = n2; n2 = n3; ++it;
        }
    }

#ifdef MPG_DWSTAMP
    if (blockIdx.x < 64 && (threadIdx.x & 63) == 0) {
        dw_acc[6] = (unsigned long long)it;
#pragma unroll
        for (int q = 0; q < 8; ++q) g_dw_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6) - 4) * 8 + q] = dw_acc[q];
    }
#endif
    // bias sums: add the 32 receivers of a chunk group -- 16 in this wave (lane bits 2..5), 16 in its neighbour wave
    // (through LDS: the images are dead after the loop's last barrier); fragment-order index fi = 8 c + k
    float* red = reinterpret_cast<float*>(smem);   // [wave 0 .. NB - 1][lane & 3][n][k][db3 | db2]
    constexpr int NM = NZ3 > N160 ? NZ3 : N160;
    const int bw = bt >> 6, bl = bt & 3;
#pragma unroll
    for (int n = 0; n < NM; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float x = n < NZ3 ? db3[n < NZ3 ? n : 0][k] : 0.f, y = n < N160 ? db2[n < N160 ? n : 0][k] : 0.f;
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
            if ((bt & 63) < 4) {
                red[(((bw * 4 + bl) * NM + n) * 8 + k) * 2 + 0] = x;
                red[(((bw * 4 + bl) * NM + n) * 8 + k) * 2 + 1] = y;
            }
        }
    lds_barrier();   // (the consumers take part in it, see dw_consumer)
    float* part = p.part + (size_t)blockIdx.x * (H3 * H2 + H2 * H1 + H3 + H2);
    float* pb3 = part + H3 * H2 + H2 * H1, *pb2 = pb3 + H3;
    if (r == 0) {   // lanes 0..3 of the even waves: their own 16 receivers + those of the wave behind them
#pragma unroll
        for (int n = 0; n < NM; ++n)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i0 = (((bw * 4 + bl) * NM + n) * 8 + k) * 2, i1 = ((((bw + 1) * 4 + bl) * NM + n) * 8 + k) * 2;
                const int c = cg + CS * n;
                if (n < NZ3) pb3[8 * c + k] = red[i0] + red[i1];
                if (n < N160 && c < 2 * NFR2) pb2[8 * c + k] = red[i0 + 1] + red[i1 + 1];  // (a repeated piece added zeros)
            }
    }
}

#ifdef MPG_DW8
// ---------------------------------------------------------------------------------------------------------------------
// EXPERIMENT, compiled with -DMPG_DW8 only (tools/ubench/dw_bench.hip) -- measured and NOT adopted: the same partials bit for bit,
// 87.6 against 63.5 us at B = 256 and 131.5 against 97.4 us at 512 ragged jets (p = 1/2: 94.3 / 73.4, 147.9 / 114.5).  With every
// wave of a SIMD in the same phase (one barrier per block keeps the eight waves in lock step) the second wave's instructions do
// not issue beside the first's: a build is stores to LDS (13 clk each on the shared store path), buffer requests and
// conversions, not the plain arithmetic of tools/ubench/valu_rate2.hip, and while everyone builds the matrix pipe idles.
// The UNIFORM form (no edge scalars): all eight waves build AND multiply.  The kernel above is bound by the instruction issue of
// its four builder waves (one per SIMD, ~7 clk per instruction, 378 instructions per block; with every staged load compiled
// out it is no faster) while its four consumer waves multiply for a quarter of a block's time -- and a second wave on a SIMD
// issues beside the first at the same rate (tools/ubench/valu_rate2.hip).  Here every wave takes 1 / 512 of a block's build (two
// dZ3 chunks, two pieces each of the parked dZ2 / E2, one E1 chunk: 7 pieces where a builder thread had 11) and owns five or
// six of the 45 output tiles; per block a wave builds block n + 1 into one image buffer and multiplies block n out of the other,
// one barrier per block as before.  Same images, same order of the blocks in every accumulator and every bias sum: the
// partials are those of the kernel above, bit for bit.
// tiles of wave w: waves 0..4 the row m = w of dW3 (5 tiles) + tile (5, w); waves 5..7 five tiles of dW2 each
__device__ constexpr DwTile DW8_TILES[45] = {
    {0,0,0},{0,0,1},{0,0,2},{0,0,3},{0,0,4},{0,5,0},
    {0,1,0},{0,1,1},{0,1,2},{0,1,3},{0,1,4},{0,5,1},
    {0,2,0},{0,2,1},{0,2,2},{0,2,3},{0,2,4},{0,5,2},
    {0,3,0},{0,3,1},{0,3,2},{0,3,3},{0,3,4},{0,5,3},
    {0,4,0},{0,4,1},{0,4,2},{0,4,3},{0,4,4},{0,5,4},
    {1,0,0},{1,0,1},{1,0,2},{1,1,0},{1,1,1},
    {1,1,2},{1,2,0},{1,2,1},{1,2,2},{1,3,0},
    {1,3,1},{1,3,2},{1,4,0},{1,4,1},{1,4,2}};
constexpr bool dw8_same_rows(int t, int u) { return DW8_TILES[t].prod == DW8_TILES[u].prod && DW8_TILES[t].m == DW8_TILES[u].m; }
constexpr bool dw8_leader(int t, int begin) { return t == begin || !dw8_same_rows(t, t - 1); }
constexpr int dw8_group_end(int t, int end) { int e = t + 1; while (e < end && dw8_same_rows(t, e)) ++e; return e; }
constexpr bool dw8_has_prod(int prod, int begin, int end) {
    for (int t = begin; t < end; ++t) if (DW8_TILES[t].prod == prod) return true;
    return false;
}
constexpr bool dw8_uses_n(int prod, int n, int begin, int end) {
    for (int t = begin; t < end; ++t) if (DW8_TILES[t].prod == prod && DW8_TILES[t].n == n) return true;
    return false;
}

// One block into the wave's accumulators.  A wave's tiles are one or two row groups (same product and m): their A fragments are
// held (two at most would be 8 registers; three for the dW2 waves), the B fragments pass one at a time -- 4 registers where
// holding a product's five cost 20 of the 256 this kernel has.
template <int BEGIN, int END>
MPG_DEV void dw8_consume(f32x16* acc, uint32_t buf, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int row = 8 * (g >> 1) + q, col = (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint32_t bz3 = buf + DW_Z3H + row * DW_RS3 + col, be2 = buf + DW_E2H + row * DW_RS2 + col;
        uint32_t bz2 = buf + DW_Z2H + row * DW_RS2 + col, be1 = buf + DW_E1H + row * DW_RS1 + col;
        asm volatile("" : "+v"(bz3), "+v"(be2), "+v"(bz2), "+v"(be1));
        static_for<0, 2>([&](auto pc) {
            MPG_CI(prod, pc);
            if constexpr (dw8_has_prod(prod, BEGIN, END)) {
                constexpr int rsa = prod == 0 ? DW_RS3 : DW_RS2, rsb = prod == 0 ? DW_RS2 : DW_RS1, nb = prod == 0 ? T2 : T1, na = prod == 0 ? T3 : T2;
                const uint32_t ba = prod == 0 ? bz3 : bz2, bb = prod == 0 ? be2 : be1;
                f16x8 afr[na];   // (only the rows this wave owns are ever read: the others are never defined)
                static_for<BEGIN, END>([&](auto tc) {
                    MPG_CI(t, tc);
                    if constexpr (DW8_TILES[t].prod == prod && dw8_leader(t, BEGIN)) afr[DW8_TILES[t].m] = dw_frag(ba, 16 * s * rsa + 64 * DW8_TILES[t].m, rsa);
                });
                static_for<0, nb>([&](auto nc) {
                    MPG_CI(n, nc);
                    if constexpr (dw8_uses_n(prod, n, BEGIN, END)) {
                        const f16x8 b = dw_frag(bb, 16 * s * rsb + 64 * n, rsb);
                        static_for<BEGIN, END>([&](auto tc) {
                            MPG_CI(t, tc);
                            if constexpr (DW8_TILES[t].prod == prod && DW8_TILES[t].n == n)
                                acc[t - BEGIN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[DW8_TILES[t].m], b, acc[t - BEGIN], 0, 0, 0);
                        });
                    }
                });
            }
        });
    }
}

template <int BEGIN, int END>
MPG_DEV void dw8_store(const f32x16* acc, float* part3, float* part2, int lane) {
    const int cc = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = BEGIN; t < END; ++t) {
        const DwTile d = DW8_TILES[t];
        float* dst = d.prod == 0 ? part3 : part2;
        const int ncol = d.prod == 0 ? H2 : H1;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = 32 * d.m + 8 * (k >> 2) + 4 * hh + (k & 3);
            dst[(size_t)row * ncol + 32 * d.n + cc] = acc[t - BEGIN][k];
        }
    }
}

template <int DROP, int BEGIN, int END>
MPG_DEV void dw8_wave(const MpgEdgeDw& p, int R, char* smem, int blk0, int blk1, unsigned long long vbits) {
    const int bt = threadIdx.x, lane = bt & 63;   // thread 0..511: receiver row r of the images, chunk group cg (0..15)
    const int r = (bt >> 2) & 31, cg = ((bt >> 7) << 2) | (bt & 3);
    const int RB = (p.N + 31) / 32;
    const int tq = cg >> 2;                       // the thread's tile within a group of four: chunks 4 (tq + 4n) + (cg & 3)
    const int cs = (cg >> 1) & 1, ch = cg & 1;
    const int f0 = 16 * cs + 4 * ch;              // features of a chunk of tile t: 32 t + f0 + {0..3, 8..11}
    // second pieces: dZ3 tiles tq + 4 (tq < 2), 160-feature tile 4 (tq == 0); E1 tile tq (tq < 3).  A thread without one redoes
    // another piece instead (same data to the same place): no branch in the build
    const bool z3b = tq < 2, p160b = tq == 0;
    const int m3[2] = {tq, z3b ? tq + 4 : tq}, t160[2] = {tq, p160b ? 4 : tq}, te1 = tq < 3 ? tq : 2;

    uint32_t seed_lo = 0, seed_hi = 0;
    if (DROP) { const uint64_t sd = *p.seed; seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32); }

    f32x16 acc[END - BEGIN];
#pragma unroll
    for (int t = 0; t < END - BEGIN; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float db3[2][8], db2[2][8];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) { db3[n][k] = 0.f; db2[n][k] = 0.f; }

    struct Small {
        float dreg[2][8], areg[8];      // dagg / a of this thread's Z3 / E1 chunks (raw)
        float dscl;                     // agg_scale * dscale * 2^eG, or 0 for a padding receiver
        uint32_t sw[2];                 // sign words of the Z3 chunks' lanes
        unsigned int nbw;               // this receiver's neighbour word holding the block's sender (k-NN graphs)
        float4 cv[2];                   // c_j of the E1 chunk
    };
    struct Big { f16x8 eh[2], zh[2]; };   // parked E2 / dZ2 pieces
    Small S;
    Big B0;
#if MPG_DW8_SETS == 2
    Big B1;
#else
    Big& B1 = B0;   // ONE set of parked pieces, requested one block ahead (two sets -- 16 more registers -- made the kernel spill)
#endif
    const int eG = dw_launch_exp(p);
    const float unitG = __builtin_bit_cast(float, (uint32_t)(eG + 127) << 23);   // 2^eG (|eG| <= 100)
    int eslot = eG;
    {
        const int t = blk0 + lane;
        if (t < blk1) eslot = p.gexp[dw_block(t, R) / p.N];
    }
    asm volatile("" : "+v"(eslot));
    int sblk, sj, sb, srb;
    {
        const int t = min(blk0 + lane, max(blk1 - 1, blk0));
        sblk = dw_block(t, R);
        const int brb = sblk / p.N;
        sj = sblk - brb * p.N;
        sb = brb / RB;
        srb = brb - sb * RB;
    }
    asm volatile("" : "+v"(sblk), "+v"(sj), "+v"(sb), "+v"(srb));
    struct Idx { int blk, j, b, rb; };
    auto idx_of = [&](int slot) {
        const int l = slot - blk0;
        return Idx{__builtin_amdgcn_readlane(sblk, l), __builtin_amdgcn_readlane(sj, l), __builtin_amdgcn_readlane(sb, l),
                   __builtin_amdgcn_readlane(srb, l)};
    };
    // p = 1/2: lane q of a quad of adjacent lanes hashes one of the quad's keep words -- dZ3 tiles 4n + q (quad lane 0..3 <-> the four
    // tiles tq = 0..3 do NOT sit in one quad here: a quad shares tq) -- so every thread hashes its own three words
    uint32_t hc3[2] = {0u, 0u}, hc1 = 0u;
    if constexpr (DROP == 2) {
#pragma unroll
        for (int n = 0; n < 2; ++n) hc3[n] = (DROP_BIT_GRP + (uint32_t)m3[n] + (p.tag_base + TAG_E2) * 0x10001u) * 0x85EBCA77u + seed_hi;
        hc1 = (DROP_BIT_GRP + (uint32_t)te1 + (p.tag_base + TAG_E0) * 0x10001u) * 0x85EBCA77u + seed_hi;
    }

    const int nblk = p.B * RB * p.N;
    const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageE2), 0, nblk * (NFR2 * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.stageZ2), 0, nblk * (NFR2 * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(p.sign3), 0, nblk * (T3 * 32 * 4), 0x00020000);
    const int ldac = p.ld_ac ? p.ld_ac : H1;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c), 0, p.B * p.N * ldac * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.B * p.N * ldac * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dagg), 0, p.B * p.N * p.ld_dagg * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned int*>(p.nbr), 0, p.nbr ? p.B * p.N * ((p.N + 31) >> 5) * 4 : 0, 0x00020000);
    // per-thread offsets are recomputed where they are used from OPAQUE copies of (r, cg): hoisted out of the block loop they are
    // a dozen registers the allocator spills (and every scratch access drains the prefetch queue: vmcnt(0))
    auto opq = [](int v) { asm volatile("" : "+v"(v)); return v; };
    auto vo160 = [&](int n) {   // byte offset of piece n inside a 10 KiB block: chunk c of receiver r is piece c * 32 + r
        const int rr = opq(r), cc_ = opq(cg), t = n == 0 ? (cc_ >> 2) : ((cc_ >> 2) == 0 ? 4 : (cc_ >> 2));
        return ((4 * t + (cc_ & 3)) * 32 + rr) * 16;
    };
    auto voS_ = [&]() { return (32 * (opq(cg) & 1) + opq(r)) * 4; };
    auto voE1_ = [&]() { const int cc_ = opq(cg); return (32 * min(cc_ >> 2, 2) + 16 * ((cc_ >> 1) & 1) + 4 * (cc_ & 1)) * 4; };
    auto ldb4 = [&](__amdgpu_buffer_rsrc_t rs, int vo, int so) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
    };
    auto load_d = [&](Small& P, const Idx& X) {
        const int ii = X.rb * 32 + r;
        const bool ok = ii < p.N;
        P.dscl = ok ? p.agg_scale * p.dscale * unitG : 0.f;
        const int cgo = opq(cg), f0o = 16 * ((cgo >> 1) & 1) + 4 * (cgo & 1);
        const int rowD = (ok ? ii : 0) * p.ld_dagg * 4 + f0o * 4, soD = X.b * p.N * p.ld_dagg * 4;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int tqo = opq(cg) >> 2, mo = n == 0 ? tqo : (tqo < 2 ? tqo + 4 : tqo);
            const float4 u = ldb4(rD, rowD + 128 * mo, soD), v = ldb4(rD, rowD + 128 * mo + 32, soD);
            P.dreg[n][0] = u.x; P.dreg[n][1] = u.y; P.dreg[n][2] = u.z; P.dreg[n][3] = u.w;
            P.dreg[n][4] = v.x; P.dreg[n][5] = v.y; P.dreg[n][6] = v.z; P.dreg[n][7] = v.w;
        }
    };
    auto load_a = [&](Small& P, const Idx& X) {
        const int ii = X.rb * 32 + r;
        const bool ok = ii < p.N;
        const int rowA = (ok ? ii : 0) * ldac * 4, soA = X.b * p.N * ldac * 4;
        const int voE1 = voE1_();
        const float4 u = ldb4(rA, rowA + voE1, soA), v = ldb4(rA, rowA + voE1 + 32, soA);
        P.areg[0] = u.x; P.areg[1] = u.y; P.areg[2] = u.z; P.areg[3] = u.w;
        P.areg[4] = v.x; P.areg[5] = v.y; P.areg[6] = v.z; P.areg[7] = v.w;
    };
    auto load_nb = [&](Small& P, const Idx& X) {
        const int ii = X.rb * 32 + r;
        P.nbw = __builtin_amdgcn_raw_buffer_load_b32(rN, ((X.b * p.N + (ii < p.N ? ii : 0)) * ((p.N + 31) >> 5) + (X.j >> 5)) * 4, 0, 0);
    };
    auto load_sw = [&](Small& P, const Idx& X) {   // word (tile >> 1) of lane (r, h)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int tqo = opq(cg) >> 2, mo = n == 0 ? tqo : (tqo < 2 ? tqo + 4 : tqo);
            P.sw[n] = __builtin_amdgcn_raw_buffer_load_b32(rS, voS_() + (mo >> 1) * 256, X.blk * (T3 * 32 * 4), 0);
        }
    };
    auto load_c = [&](Small& P, const Idx& X) {
        const int so = (X.b * p.N + X.j) * ldac * 4;
        const int voE1 = voE1_();
        P.cv[0] = ldb4(rC, voE1, so); P.cv[1] = ldb4(rC, voE1 + 32, so);
    };
    auto load_e2 = [&](Big& P, int blk, int n) {
        P.eh[n] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rE, vo160(n), blk * (NFR2 * 1024), 0));
    };
    auto load_z2 = [&](Big& P, int blk, int n) {
        P.zh[n] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rZ, vo160(n), blk * (NFR2 * 1024), 0));
    };
    auto load_small = [&](Small& P, const Idx& X) { load_d(P, X); load_sw(P, X); load_nb(P, X); load_c(P, X); load_a(P, X); };
    auto load_big = [&](Big& Q, int blk) {
#pragma unroll
        for (int n = 0; n < 2; ++n) { load_z2(Q, blk, n); load_e2(Q, blk, n); }
    };

    // build the block of slot `slot` into buffer `buf` (dw_builder's build, on this thread's pieces)
    auto build = [&](int slot, char* buf, int pre1_slot, int pre2_slot, Big& Q) {
        const Idx X = idx_of(slot), X1 = idx_of(pre1_slot);
        const int blk = X.blk, pre2 = MPG_DW8_SETS == 2 ? __builtin_amdgcn_readlane(sblk, pre2_slot - blk0) : X1.blk;
        const bool newrun = X1.b != X.b || X1.rb != X.rb;
        const int cgb = opq(cg), tqb = cgb >> 2, cql = cgb & 3;
        const int m3b[2] = {tqb, tqb < 2 ? tqb + 4 : tqb}, t160b[2] = {tqb, tqb == 0 ? 4 : tqb}, te1b = min(tqb, 2);
        const int j = X.j, ii = X.rb * 32 + opq(r);
        const uint32_t erow = (uint32_t)((X.b * p.N + ii) * p.N + j);
        const int de = eG - __builtin_amdgcn_readlane(eslot, slot - blk0);
        const float dth = dither_of((uint32_t)blk), rdth = __builtin_amdgcn_rcpf(dth);
        const float funit = __builtin_bit_cast(float, (uint32_t)max(de + 127, 0) << 23) * rdth;
        const _Float16 rch = (_Float16)rdth;
        const float in_set = (p.nbr == nullptr || ((S.nbw >> (j & 31)) & 1u)) ? 1.f : 0.f;
        const float dscl_1 = S.dscl * in_set * dth, dscl_a = dscl_1 * p.alpha;
        uint32_t kz3[2] = {0xffu, 0xffu}, ke1 = 0xffu;
        if constexpr (DROP == 2) {
            const uint32_t x0 = (erow + seed_lo) * 0x9E3779B1u;
            auto fin = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
            auto bits = [&](uint32_t w) { w >>= f0; return (w & 0xfu) | ((w >> 4) & 0xf0u); };   // bits f0..f0+3 and f0+8..f0+11
            kz3[0] = bits(fin(x0 ^ hc3[0]));
            kz3[1] = bits(fin(x0 ^ hc3[1]));
            ke1 = bits(fin(x0 ^ hc1));
        } else if constexpr (DROP == 1) {
#pragma unroll
            for (int n = 0; n < 2; ++n) kz3[n] = dw_chunk_keep<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E2, erow, m3[n], f0, p.thr);
            ke1 = dw_chunk_keep<DROP>(seed_lo, seed_hi, p.tag_base + TAG_E0, erow, te1, f0, p.thr);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int m = m3b[n], c = 4 * m + cql;
            float v[8];
            const uint32_t keep = kz3[n];
            const uint32_t swn = S.sw[n] << (16 * (m & 1) + 8 * cs);
            const float take = (n == 0 || z3b) ? rdth : 0.f;   // (a repeated piece adds nothing to the bias sums)
            static_for<0, 8>([&](auto kc) {
                MPG_CI(k, kc);
                float x = S.dreg[n][k] * sel_by_bit<31 - k>(swn, dscl_a, dscl_1);
                if (DROP && !((keep >> k) & 1u)) x = 0.f;
                v[k] = x;
                db3[n][k] = fmaf(x, take, db3[n][k]);
            });
            const f16x8 hh = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
            *reinterpret_cast<f16x8*>(buf + DW_Z3H + opq(r) * DW_RS3 + c * 16) = hh;
        }
        load_sw(S, X1); load_nb(S, X1); if (newrun) load_d(S, X1);
        {
            const int c = 4 * te1b + cql;
            const float cc[8] = {S.cv[0].x, S.cv[0].y, S.cv[0].z, S.cv[0].w, S.cv[1].x, S.cv[1].y, S.cv[1].z, S.cv[1].w};
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float x = lrelu(S.areg[k] + cc[k], p.alpha) * funit;
                if (DROP && !((ke1 >> k) & 1u)) x = 0.f;
                v[k] = x;
            }
            const f16x8 hh = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
            *reinterpret_cast<f16x8*>(buf + DW_E1H + opq(r) * DW_RS1 + c * 16) = hh;
        }
        load_c(S, X1); if (newrun) load_a(S, X1);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int c = 4 * t160b[n] + cql;
            *reinterpret_cast<f16x8*>(buf + DW_Z2H + opq(r) * DW_RS2 + c * 16) = Q.zh[n];
            const float take = (n == 0 || p160b) ? funit : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) db2[n][k] += take * (float)Q.zh[n][k];
            load_z2(Q, pre2, n);
        }
        const f16x8 rc8 = {rch, rch, rch, rch, rch, rch, rch, rch};
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int c = 4 * t160b[n] + cql;
            *reinterpret_cast<f16x8*>(buf + DW_E2H + opq(r) * DW_RS2 + c * 16) = Q.eh[n] * rc8;
            load_e2(Q, pre2, n);
        }
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    int cur = dw_next_valid(vbits, blk0, blk0, blk1), it = 0;
    int n1 = dw_next_valid(vbits, blk0, cur + 1, blk1), n2 = dw_next_valid(vbits, blk0, n1 + 1, blk1);
    if (cur < blk1) {
        load_small(S, idx_of(cur));
        load_big(B0, idx_of(cur).blk);
#if MPG_DW8_SETS == 2
        load_big(B1, idx_of(min(n1, blk1 - 1)).blk);
#endif
        build(cur, smem, min(n1, blk1 - 1), min(n2, blk1 - 1), B0);
    }
    lds_barrier();
#ifdef MPG_DWSTAMP
    unsigned long long dw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dw_t = __builtin_amdgcn_s_memtime();
#endif
    while (cur < blk1) {
        {   // odd blocks: parked pieces from set 1
            const int n3 = dw_next_valid(vbits, blk0, n2 + 1, blk1);
            if (n1 < blk1) build(n1, smem + ((it + 1) & 1) * DW_BUF, min(n2, blk1 - 1), min(n3, blk1 - 1), B1);
            DW_STAMP(0);
            dw8_consume<BEGIN, END>(acc, lds0 + (it & 1) * DW_BUF, lane);
            DW_STAMP(1);
            lds_barrier();
            DW_STAMP(2);
            cur = n1; n1 = n2; n2 = n3; ++it;
        }
        if (!(cur < blk1)) break;
        {   // even blocks: set 0
            const int n3 = dw_next_valid(vbits, blk0, n2 + 1, blk1);
            if (n1 < blk1) build(n1, smem + ((it + 1) & 1) * DW_BUF, min(n2, blk1 - 1), min(n3, blk1 - 1), B0);
            DW_STAMP(0);
            dw8_consume<BEGIN, END>(acc, lds0 + (it & 1) * DW_BUF, lane);
            DW_STAMP(1);
            lds_barrier();
            DW_STAMP(2);
            cur = n1; n1 = n2; n2 = n3; ++it;
        }
    }
#ifdef MPG_DWSTAMP
    if (blockIdx.x < 32 && lane == 0) {
        dw_acc[6] = (unsigned long long)it;
#pragma unroll
        for (int q = 0; q < 8; ++q) g_dw_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + q] = dw_acc[q];
    }
#endif
    float* part = p.part + (size_t)blockIdx.x * (H3 * H2 + H2 * H1 + H3 + H2);
    dw8_store<BEGIN, END>(acc, part, part + H3 * H2, lane);
    // bias sums: add the 32 receivers of a chunk group -- 16 in this wave (lane bits 2..5), 16 in its neighbour wave (through
    // LDS: the images are dead behind the loop's last barrier); fragment-order index fi = 8 c + k
    float* red = reinterpret_cast<float*>(smem);   // [wave 0..7][lane & 3][n][k][db3 | db2]
    const int bw = bt >> 6, bl = bt & 3;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float x = db3[n][k], y = db2[n][k];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
            if (lane < 4) {
                red[(((bw * 4 + bl) * 2 + n) * 8 + k) * 2 + 0] = x;
                red[(((bw * 4 + bl) * 2 + n) * 8 + k) * 2 + 1] = y;
            }
        }
    lds_barrier();
    float* pb3 = part + H3 * H2 + H2 * H1, *pb2 = pb3 + H3;
    if (r == 0) {   // lanes 0..3 of the even waves: their own 16 receivers + those of the odd neighbour
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i0 = (((bw * 4 + bl) * 2 + n) * 8 + k) * 2, i1 = ((((bw + 1) * 4 + bl) * 2 + n) * 8 + k) * 2;
                if (n == 0 || z3b) pb3[8 * (4 * m3[n] + (cg & 3)) + k] = red[i0] + red[i1];
                if (n == 0 || p160b) pb2[8 * (4 * t160[n] + (cg & 3)) + k] = red[i0 + 1] + red[i1 + 1];
            }
    }
}

template <int DROP>
__global__ __launch_bounds__(512, 1) void edge_dw8_kernel(const MpgEdgeDw p, const int R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RB = (p.N + 31) / 32;
    const int nruns = p.B * RB * p.N / R;
    const int blk0 = 0, blk1 = R * ((nruns - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);   // slots of this workgroup
    const unsigned long long vbits = dw_valid_bits(p, R, blk0, blk1);
    if (w == 0) dw8_wave<DROP, 0, 6>(p, R, smem, blk0, blk1, vbits);
    else if (w == 1) dw8_wave<DROP, 6, 12>(p, R, smem, blk0, blk1, vbits);
    else if (w == 2) dw8_wave<DROP, 12, 18>(p, R, smem, blk0, blk1, vbits);
    else if (w == 3) dw8_wave<DROP, 18, 24>(p, R, smem, blk0, blk1, vbits);
    else if (w == 4) dw8_wave<DROP, 24, 30>(p, R, smem, blk0, blk1, vbits);
    else if (w == 5) dw8_wave<DROP, 30, 35>(p, R, smem, blk0, blk1, vbits);
    else if (w == 6) dw8_wave<DROP, 35, 40>(p, R, smem, blk0, blk1, vbits);
    else dw8_wave<DROP, 40, 45>(p, R, smem, blk0, blk1, vbits);
}

#endif   // MPG_DW8

template <int DROP, int NQ>
__global__ __launch_bounds__(512, 1) void edge_dw_kernel(const MpgEdgeDw p, const int R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RB = (p.N + 31) / 32;
    const int nruns = p.B * RB * p.N / R;
    const int blk0 = 0, blk1 = R * ((nruns - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);   // slots of this workgroup
    const unsigned long long vbits = dw_valid_bits(p, R, blk0, blk1);
    if (w == 0) dw_consumer<0, 12, NQ>(p, blk0, blk1, vbits);
    else if (w == 1) dw_consumer<12, 23, NQ>(p, blk0, blk1, vbits);
    else if (w == 2) dw_consumer<23, 34, NQ>(p, blk0, blk1, vbits);
    else if (w == 3) dw_consumer<34, 45, NQ>(p, blk0, blk1, vbits);
    else dw_builder<DROP, NQ>(p, R, smem, blk0, blk1, vbits);
}

// TWELVE waves, three per SIMD at <= 168 registers: six consumers (eight output tiles each -- 128 accumulator registers -- of
// DW_TILES6) and six builders.  The builders are what bounds this kernel and a wave issues one VALU instruction per ~7 clk whatever
// shares its SIMD: half as many builder instructions per wave again, on the same four SIMDs, is a block built in
// two thirds of the time -- for twice the B-fragment reads of the consumers (each reads all of E2's five).  Same partials bit for bit.
template <int DROP>
__global__ __launch_bounds__(768, 3) void edge_dw12_kernel(const MpgEdgeDw p, const int R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RB = (p.N + 31) / 32;
    const int nruns = p.B * RB * p.N / R;
    const int blk0 = 0, blk1 = R * ((nruns - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);   // slots of this workgroup
    const unsigned long long vbits = dw_valid_bits(p, R, blk0, blk1);
    if (w == 0) dw_consumer<0, 8, 0, 1>(p, blk0, blk1, vbits);
    else if (w == 1) dw_consumer<8, 16, 0, 1>(p, blk0, blk1, vbits);
    else if (w == 2) dw_consumer<16, 24, 0, 1>(p, blk0, blk1, vbits);
    else if (w == 3) dw_consumer<24, 32, 0, 1>(p, blk0, blk1, vbits);
    else if (w == 4) dw_consumer<32, 40, 0, 1>(p, blk0, blk1, vbits);
    else if (w == 5) dw_consumer<40, 45, 0, 1>(p, blk0, blk1, vbits);
    else dw_builder<DROP, 0, 6, 6>(p, R, smem, blk0, blk1, vbits);
}

// out = scale * 2^-eG * sum over workgroup partials, feature indices mapped back from fragment order.
// 32 outputs x 8 partial-slices per block: the 256 partials of an output are read by 8 threads.
MPG_DEV void edge_dw_reduce_body(const int blk, const float* __restrict__ part, int nwg, float scale3, float scale, int accumulate,
                                 const int* __restrict__ gexp, int ngexp, float* __restrict__ dW3, float* __restrict__ dW2,
                                 float* __restrict__ db3, float* __restrict__ db2) {
    __shared__ float red[8][32];
    __shared__ int emin[4];
    {   // the launch's gradient unit (as dw_launch_exp)
        int e = 0x7fffffff;
        for (int t = threadIdx.x; t < ngexp; t += 256) e = min(e, gexp[t]);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) e = min(e, __shfl_xor(e, o, 64));
        if ((threadIdx.x & 63) == 0) emin[threadIdx.x >> 6] = e;
    }
    constexpr int PER = H3 * H2 + H2 * H1 + H3 + H2;
    const int ix = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = blk * 32 + ix;
    float s = 0.f;
    if (idx < PER) {   // (four independent partial sums: a thread's loads are in flight together; fixed summation order)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int g = sl;
        for (; g + 24 < nwg; g += 32) {
            s0 += part[(size_t)g * PER + idx]; s1 += part[(size_t)(g + 8) * PER + idx];
            s2 += part[(size_t)(g + 16) * PER + idx]; s3 += part[(size_t)(g + 24) * PER + idx];
        }
        for (; g < nwg; g += 8) s0 += part[(size_t)g * PER + idx];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][ix] = s;
    __syncthreads();
    if (sl != 0 || idx >= PER) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][ix];
    s *= __builtin_bit_cast(float, (uint32_t)(127 - min(min(emin[0], emin[1]), min(emin[2], emin[3]))) << 23);
    if (idx < H3 * H2) {
        const int r3 = idx / H2, c2 = idx % H2;
        float* d = dW3 + feat_of_fi(r3) * H2 + feat_of_fi(c2);
        *d = s * scale3 + (accumulate ? *d : 0.f);
    } else if (idx < H3 * H2 + H2 * H1) {
        const int k = idx - H3 * H2, r2 = k / H1, c1 = k % H1;
        float* d = dW2 + feat_of_fi(r2) * H1 + feat_of_fi(c1);
        *d = s * scale + (accumulate ? *d : 0.f);
    } else if (idx < H3 * H2 + H2 * H1 + H3) {
        float* d = db3 + feat_of_fi(idx - H3 * H2 - H2 * H1);
        *d = s + (accumulate ? *d : 0.f);
    } else {
        float* d = db2 + feat_of_fi(idx - H3 * H2 - H2 * H1 - H3);
        *d = s + (accumulate ? *d : 0.f);
    }
}

__global__ __launch_bounds__(256) void edge_dw_reduce(const float* __restrict__ part, int nwg, float scale3, float scale, int accumulate,
                                                      const int* __restrict__ gexp, int ngexp,
                                                      float* __restrict__ dW3, float* __restrict__ dW2,
                                                      float* __restrict__ db3, float* __restrict__ db2) {
    edge_dw_reduce_body((int)blockIdx.x, part, nwg, scale3, scale, accumulate, gexp, ngexp, dW3, dW2, db3, db2);
}
// ... and with the grouped split-K reductions of the layer's dense weight gradients riding in the same launch (blocks nb_dw ..):
// one dependent launch less in the weight-gradient tail of every layer's backward
__global__ __launch_bounds__(256) void edge_dw_reduce_group(const float* __restrict__ part, int nwg, float scale3, float scale, int accumulate,
                                                            const int* __restrict__ gexp, int ngexp,
                                                            float* __restrict__ dW3, float* __restrict__ dW2,
                                                            float* __restrict__ db3, float* __restrict__ db2,
                                                            const ReduceGroup R, const int nb_dw) {
    if ((int)blockIdx.x < nb_dw) edge_dw_reduce_body((int)blockIdx.x, part, nwg, scale3, scale, accumulate, gexp, ngexp, dW3, dW2, db3, db2);
    else splitk_reduce_group_body(R, (int)blockIdx.x - nb_dw);
}

}  // namespace

#ifdef MPG_DW_Q_UNIT   // edge_dw_q.hip: the variants with edge scalars, compiled beside this unit
int mpg_edge_dw_q(const MpgEdgeDw* p, int R, hipStream_t st) {
    dim3 grid(p->nwg), block(512);
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
    constexpr int LDSQ = DW_LDS_BYTES + MPG_EDGE_SCALARS * H1 * 4;
#define MPG_DW_Q(D)                                                                                               \
    do {                                                                                                          \
        MPG_ENSURE_LDS((edge_dw_kernel<D, MPG_EDGE_SCALARS>), LDSQ);                                              \
        hipLaunchKernelGGL((edge_dw_kernel<D, MPG_EDGE_SCALARS>), grid, block, LDSQ, st, *p, R);                  \
    } while (0)
    if (dm == 0) MPG_DW_Q(0);
    else if (dm == 1) MPG_DW_Q(1);
    else MPG_DW_Q(2);
#undef MPG_DW_Q
    return (int)hipGetLastError();
}
#else
int mpg_edge_dw_q(const MpgEdgeDw* p, int R, hipStream_t st);

extern "C" int mpg_edge_dw(const MpgEdgeDw* p, void* stream) {
    if (p->B <= 0 || p->N <= 0 || p->nwg <= 0) return -1;
    if (!p->f16) return -8;
    if (p->gexp == nullptr) return -9;
    int R = 1;   // run length: the largest divisor of N up to 6 (see dw_block)
    static const int rmax = [] { const char* e = getenv("MPG_DW_RUN"); return e != nullptr ? atoi(e) : 6; }();   // (experiments: longer runs re-read fewer rows)
    for (int d = 2; d <= rmax; ++d) if (p->N % d == 0) R = d;
    {
        const int nruns = p->B * ((p->N + 31) / 32) * p->N / R;
        if (R * ((nruns + p->nwg - 1) / p->nwg) > 64) return -5;  // a workgroup walks at most 64 blocks (one ballot of valid bits)
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(p->nwg), block(512);
    const int dm = p->thr == 0 ? 0 : (p->thr == 128 ? 2 : 1);
#ifdef MPG_DW8   // (experiment build: MPG_DW_FORM=uniform picks the uniform kernel)
    static const bool uniform = [] { const char* e = getenv("MPG_DW_FORM"); return e != nullptr && strcmp(e, "uniform") == 0; }();
#define MPG_DW_ONE(D)                                                                                             \
    do {                                                                                                          \
        if (uniform) {                                                                                            \
            MPG_ENSURE_LDS((edge_dw8_kernel<D>), DW_LDS_BYTES);                                                   \
            hipLaunchKernelGGL((edge_dw8_kernel<D>), grid, block, DW_LDS_BYTES, st, *p, R);                       \
        } else {                                                                                                  \
            MPG_ENSURE_LDS((edge_dw_kernel<D, 0>), DW_LDS_BYTES);                                                 \
            hipLaunchKernelGGL((edge_dw_kernel<D, 0>), grid, block, DW_LDS_BYTES, st, *p, R);                     \
        }                                                                                                         \
    } while (0)
#else
    // six consumers + six builders, three waves per SIMD.  (The four + four form of round 4 -- edge_dw_kernel<D, 0>, what the
    // edge-scalar variant still is -- lost its A/B by 1.5-3 % per launch and is built by the harness only: -DMPG_DW_FOUR.)
#ifdef MPG_DW_FOUR
#define MPG_DW_ONE(D)                                                                                             \
    do {                                                                                                          \
        MPG_ENSURE_LDS((edge_dw_kernel<D, 0>), DW_LDS_BYTES);                                                     \
        hipLaunchKernelGGL((edge_dw_kernel<D, 0>), grid, block, DW_LDS_BYTES, st, *p, R);                         \
    } while (0)
#else
#define MPG_DW_ONE(D)                                                                                             \
    do {                                                                                                          \
        MPG_ENSURE_LDS((edge_dw12_kernel<D>), DW_LDS_BYTES);                                                      \
        hipLaunchKernelGGL((edge_dw12_kernel<D>), grid, dim3(768), DW_LDS_BYTES, st, *p, R);                      \
    } while (0)
#endif
#endif
#ifdef MPG_SINGLE_VARIANT  // tools/ubench/dw_bench.hip: one instantiation
    MPG_DW_ONE(MPG_SINGLE_VARIANT);
#else
    if (p->es != nullptr) {
        if (p->wq == nullptr) return -3;
        if (int e = mpg_edge_dw_q(p, R, st)) return e;
    } else if (dm == 0) MPG_DW_ONE(0);
    else if (dm == 1) MPG_DW_ONE(1);
    else MPG_DW_ONE(2);
#endif
#undef MPG_DW_ONE
    if (p->defer_reduce) return (int)hipGetLastError();   // (mpg_splitk_reduce_group_dw adds the partials up)
    constexpr int PER = H3 * H2 + H2 * H1 + H3 + H2;
    // (the parked E2 carries the forward's operand scale SC_E2; db3 / db2 only the gradient unit)
    hipLaunchKernelGGL(edge_dw_reduce, dim3((PER + 31) / 32), dim3(256), 0, st, p->part, p->nwg, p->dscale / SC_E2, p->dscale, p->accumulate,
                       p->gexp, p->B * ((p->N + 31) / 32), p->dW3, p->dW2, p->db3, p->db2);
    return (int)hipGetLastError();
}

extern "C" int mpg_splitk_reduce_group_dw(const MpgReduceJob* jobs, int n, const MpgEdgeDw* p, void* stream) {
    if (p == nullptr || p->nwg <= 0 || p->gexp == nullptr) return -1;
    ReduceGroup R;
    const int nb = make_reduce_group(jobs, n, R);
    if (nb < 0) return -1;
    constexpr int PER = H3 * H2 + H2 * H1 + H3 + H2;
    const int nb_dw = (PER + 31) / 32;
    hipLaunchKernelGGL(edge_dw_reduce_group, dim3(nb_dw + nb), dim3(256), 0, (hipStream_t)stream, p->part, p->nwg, p->dscale / SC_E2, p->dscale,
                       p->accumulate, p->gexp, p->B * ((p->N + 31) / 32), p->dW3, p->dW2, p->db3, p->db2, R, nb_dw);
    return (int)hipGetLastError();
}
#endif
