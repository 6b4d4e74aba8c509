/* mpgan_amd.h -- C ABI of libmpgan_amd.so (MI355X / gfx950 hot path of MPGAN + GAPT).
 *
 * Plain pointers and sizes only: every pointer is a DEVICE pointer into caller-owned fp32
 * (or, where noted, packed-bf16 "image") memory, `stream` is a hipStream_t passed as
 * void*, and every entry point enqueues work on that stream and returns at once
 * (0 = success, otherwise a hipError_t value or a negative argument-error code).  Nothing
 * is allocated, freed or synchronised inside, so every call can be captured in a hipGraph.
 *
 * The reference (rkansal47/MPGAN) has no FFI: its hot path is Python calling ATen.  Each
 * entry point below therefore cites the reference Python lines whose ATen work it replaces
 * (paths relative to the reference root); INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 */
#ifndef MPGAN_AMD_H
#define MPGAN_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- generic fused linear (nn.Linear + LeakyReLU + Dropout and their backward) -----------
 * Replaces LinearNet.forward's per-layer addmm / leaky_relu / dropout (mpgan/model.py:77-83,
 * gapt/model.py:78-84) and their autograd backward for the node network fn, the final fnd
 * layer and GAPT's projections. */
typedef struct MpgGemm {
    const float* A;      /* A(m,k) = ak ? A[m*lda+k] : A[k*lda+m]                                   */
    const float* A2;     /* optional second K-segment: columns k >= K1 come from A2(m, k-K1) (ak=1) */
    int lda, lda2, K1;
    const float* B;      /* B(k,n) = bk ? B[n*ldb+k] : B[k*ldb+n]                                   */
    int ldb;
    float* C;            /* C[m*ldc+n]; with split-K, slice z is written at C + z*split_stride      */
    int ldc;
    int M, N, K;
    long long split_stride;
    const float* bias;   /* [N] or NULL                                                             */
    float out_scale;     /* product is multiplied by this before the bias                           */
    int act;             /* 1 = LeakyReLU(alpha)                                                    */
    float alpha;
    const uint64_t* seed;/* device pointer to the 64-bit dropout seed (NULL = no dropout anywhere)  */
    uint32_t drop_tag, drop_thr; float drop_scale;   /* forward dropout on C (thr = round(256 p))   */
    const float* gateH;  /* backward: C *= d(dropout o act)/dz evaluated from the saved output H    */
    int ldh, gate_act;
    uint32_t gate_tag, gate_thr; float gate_scale;
    const float* resid;  /* C += resid[m*ldr+n]                                                     */
    int ldr;
    int accumulate;      /* C += result instead of C = result                                       */
    int f16;             /* 1: split operands as fp16 hi/lo (forward products); 0: bf16 hi/lo (gradients);
                          * 2 (mpg_gemm_wgrad_group only, every job of the call): one-term fp16 products on 128 x 128 tiles with a
                          * power-of-two unit per 128 rows of A -- needs 16-byte aligned B, ldb % 4 == 0 and rows of B at least
                          * as long as their width rounded up to 4 (-3 otherwise)                                     */
    int ones_col;        /* 1: column N-1 of B is the constant 1 (N counts it): C[:, N-1] = row sums of A  */
} MpgGemm;

int mpg_gemm(const MpgGemm* g, int ak, int bk, int splitk, void* stream);

/* Sum S split-K partial slices [S][N][K + has_bias] into out[n*ldo + k] (and bias[n] from the extra column). */
int mpg_splitk_reduce(const float* part, int S, int N, int K, int has_bias, float* out, int ldo, float* bias,
                      void* stream);

/* Several weight-gradient GEMMs (dW = dY^T X: AK = BK = 0, split-K partials) / their split-K reductions in ONE
 * launch each: the six weight gradients of an MPLayer are each too small to fill the chip on their own. */
#define MPG_GROUP_MAX 16
typedef struct MpgReduceJob {
    const float* part; int S, N, K, has_bias; float* out; int ldo; float* bias;
    int accumulate;      /* add to out / bias instead of overwriting (gradient accumulation into .grad) */
} MpgReduceJob;
int mpg_gemm_wgrad_group(const MpgGemm* g, const int* splitk, int n, void* stream);
int mpg_splitk_reduce_group(const MpgReduceJob* jobs, int n, void* stream);

/* out = in * gate(H): backward through Dropout (and LeakyReLU when gate_act) ahead of a GEMM. */
int mpg_gate(const float* in, int ldi, const float* H, int ldh, float* out, int ldo, int M, int N,
             int gate_act, float alpha, const uint64_t* seed, uint32_t tag, uint32_t thr, float scale,
             void* stream);

/* mpg_slab_sums: out[m, 0:cols] = sum_s A[s][m][0:cols], out[m, cols:2 cols] = sum_s B[s][m][0:cols] (slab s at A + s * strideA
 * floats, rows dense; added in index order).  The partial receiver gradients da (one slab per sender chunk) and sender gradients dc
 * (one slab per receiver block) of a sender-chunked mpg_edge_bwd launch, as the [da | dc] rows the layer's dx chain and its first
 * layer's weight gradient read -- replaces the two `sum(0)` calls autograd would issue for the reference's
 * `A.view(B, N, N, -1).sum(2)` backward (mpgan/model.py:257-267).  cols % 4 == 0, 16-byte aligned pointers (-5). */
int mpg_slab_sums(const float* A, int slabsA, uint64_t strideA, const float* B, int slabsB, uint64_t strideB,
                  float* out, int M, int cols, void* stream);

/* Test helper: the {0,1} keep mask [rows, F] the kernels use for dropout site `tag`. */
int mpg_dropout_mask(float* out, uint64_t rows, int F, const uint64_t* seed, uint32_t tag, uint32_t thr,
                     void* stream);

/* ---- fused edge network of MPLayer ------------------------------------------------------------
 * Widths are the reference defaults fe = [96,160,192] (setup_training.py:471-477).
 *
 * mpg_pack_weights: W[rows,cols] (row stride ldw; transpose=1 reads W^T) * scale  ->  bf16 hi/lo
 * (f16=0) or fp16 hi/lo (f16=1)
 * fragment image of ceil(rows/32) x ceil(cols/32) x 2 fragments of 1 KiB, hi part then lo part
 * (2 * MT*QT*2 KiB in all).  Layer images: W2 = fe.net.1.weight (160x96), W3 = fe.net.2.weight
 * (192x160); the backward also takes their transposes.  When dropout is on, `scale` carries the
 * 1/(1-p) of the dropout in FRONT of that layer (inverted dropout commutes with LeakyReLU). */
int mpg_pack_weights(const float* W, int ldw, int rows, int cols, int transpose, float scale, int f16,
                     void* img, void* stream);

/* mpg_pack_many: up to MPG_PACK_MAX_JOBS images in one launch (all images of a layer after an optimizer step).
 * rows/cols are those of the PACKED matrix, which is W (transpose = 0) or W^T (transpose = 1).  row_split > 0
 * reads a stacked view of W: logical row n of the un-transposed matrix is W[n % row_split, (n / row_split) *
 * split_cols + col] -- fe.net.0.weight [96, 2F] seen as [a-half ; c-half] = [192, F]. */
#define MPG_PACK_MAX_JOBS 24
typedef struct MpgPackJob {
    const float* W; int ldw, rows, cols, transpose; float scale; int f16; void* img;
    int row_split, split_cols;
    int* status;   /* optional device word, OR-ed into: 1 = an element of an fp16 image is beyond fp16's range after scaling
                      (|W * scale| > 65504: the image holds inf and every product with it is lost), 2 = a weight is not finite */
} MpgPackJob;
int mpg_pack_many(const MpgPackJob* jobs, int njobs, void* stream);

/* mpg_chain: up to three Linear(+LeakyReLU)(+Dropout) layers -- or their input-gradient chain -- on row blocks
 * of 32 without the intermediates returning to memory as operands.  Replaces LinearNet.forward
 * (mpgan/model.py:70-85) for MPLayer.fn (:268-279) and, in its backward, the three dX = dY W products.
 *   layer l:  z = Wimg_l x + bias ;  y = z                      (act = 0)
 *                                    y = LeakyReLU(z)            (act = 1)
 *             y = dropout(y)  (drop_thr != 0: keep * drop_scale) ; y *= gate(H)  (gateH != NULL: derivative of
 *             Dropout o LeakyReLU [gate_act] of the forward layer whose OUTPUT is H, as mpg_gemm's gate) ;
 *             y += resid (resid != NULL)
 *   input  :  rows of A (sum of a_slabs slabs, K1 columns) followed by rows of A2 (K - K1 columns); in_thr != 0
 *             multiplies it by a dropout keep mask first (backward of a trailing dropout) and in_out, if given,
 *             receives that gated input.
 * Wimg_l is the mpg_pack_weights image of the [N, K] matrix applied (the transposed weight in the backward),
 * fp16 hi/lo when f16 else bf16 hi/lo.  K, N <= 256 for every layer but the last (N unbounded there).
 * wscale / ascale: exact power-of-two operand scales that keep the lo halves of fp16 pairs out of the subnormal
 * range (an fp16 hi/lo pair has its 22 bits only for |x| >= 2^-3); results are unscaled before the epilogue. */
typedef struct MpgChainLayer {
    const void* Wimg; const float* bias; int nbias;      /* bias[n] for n < nbias, 0 beyond (nbias = 0: all N) */
    int K, N, act;
    uint32_t drop_tag, drop_thr; float drop_scale;
    const float* gateH; int ldh, gate_act;
    uint32_t gate_tag, gate_thr; float gate_scale;
    const float* resid; int ldr;                         /* y += resid[m, n] (after everything else), or NULL */
    float* out; int ldo;
    float wscale;                                        /* the image holds wscale * W (a power of two; 0 = 1): z is divided by it */
} MpgChainLayer;
typedef struct MpgChain {
    const float* A; int lda, K1;
    const float* A2; int lda2;
    int a_slabs; uint64_t a_slab_stride;
    uint32_t in_tag, in_thr; float in_scale;
    float* in_out; int ld_in_out;
    int M, nlayers; float alpha; const uint64_t* seed; int f16;
    float ascale;                                        /* activations are split as ascale * x (a power of two; 0 = 1) */
    MpgChainLayer L[3];
} MpgChain;
int mpg_chain(const MpgChain* p, void* stream);

/* mpg_edge_fwd: replaces MPLayer._getA_fully_connected + self.fe(A) + mask multiply + sum/mean
 * (mpgan/model.py:241, :256-267, :284-317).  Inputs are the layer-1 node terms
 *   a[b,i,:] = W1[:, :F] x_i + b1   (receiver),   c[b,j,:] = W1[:, F:] x_j   (sender),
 * so that fe.net.0 applied to [x_i ; x_j] is a_i + c_j exactly.  Output
 *   agg[b,i,:] = agg_scale * sum_j mask[b,j] * fe([x_i ; x_j])        ([B,N,192], agg_scale = 1 | 1/N)
 * With SC > 1 the senders are split over SC workgroups and agg has a leading [SC] axis of partial
 * sums the caller adds up (mpg_edge_fwd_fn with `tickets` adds them up itself: the last workgroup to arrive for a
 * (jet, receiver block) reads the SC slabs in chunk order -- a fixed order, whoever arrives last --, leaves the
 * total in slab 0 and runs the epilogue). */
typedef struct MpgEdgeFwd {
    const float* a; const float* c;       /* [B*N, 96], row stride ld_ac (0 = 96)             */
    int ld_ac;
    const float* mask;                    /* [B*N] (1 real / 0 padded) or NULL                */
    const void* W2img; const void* W3img; /* from mpg_pack_weights                            */
    const float* b2; const float* b3;     /* fe.net.1.bias [160], fe.net.2.bias [192]         */
    float* agg;                           /* [SC, B*N, 192]                                   */
    int B, N, SC;
    float alpha, agg_scale;
    const uint64_t* seed; uint32_t tag_base, thr; float dscale;  /* dropout: thr=round(256p), dscale=1/(1-p_eff) */
    int skip_masked;                      /* skip senders with mask == 0 (exact: they add 0)   */
    int f16;                              /* images and activations are fp16 hi/lo (else bf16)  */
    unsigned int* sign3;                  /* optional [B*RB*N][3][64] per-lane sign words of Z3 for the backward (NULL = off) */
    const unsigned int* nbr;              /* optional k-nearest-neighbour graph: [B*N][ceil(N/32)] words, bit j of row (b, i) set
                                             <=> sender j is a neighbour of receiver i (mpg_knn_sets); NULL = fully connected */
    void* stageE2;                        /* with sign3: E2 = fe.net.1's output (in the forward's operand scale) parked as fp16
                                             fragments [B*RB*N blocks][10][64 lanes][8] for mpg_edge_bwd / mpg_edge_dw */
    /* Edge features and row-tiled conditioning columns of MPLayer (mpgan/model.py:247-253, :297-313: delta_r, clabels,
     * mask_fne_np) -- MPG_EDGE_SCALARS scalars per edge, each times its own column of fe.net.0.weight:
     *   Z1(i, j) = a_i + c_j + sum_q es[b][j][q][i] * wq[q][:]
     * es is [B][N senders][MPG_EDGE_SCALARS][N receivers], wq [MPG_EDGE_SCALARS][96] (unused scalars: zero columns); NULL = none.
     * The scalars themselves (4 bytes per edge and scalar, not the 2F+ features per edge of the reference's edge matrix) are
     * the caller's: a norm of a coordinate difference, a gather from a [B] table. */
    const float* es; const float* wq;
    const int* order;                     /* optional [B]: workgroup g works on jet order[g] (mpg_jet_order); NULL = in index order */
    unsigned int* tickets;                /* mpg_edge_fwd_fn with SC > 1: [B*RB] arrival counters, ZERO on entry and left zero (the
                                             workgroup of a (jet, receiver block) that arrives last adds up the SC partial slabs);
                                             NULL = the epilogue form takes whole jets only (SC = 1) */
    int two_term;                         /* product form of the two dense layers: 0 = every product as three 16-bit terms (hi*hi +
                                             hi*lo + lo*hi: fp32-level pre-activations); 1 = fe.net.2 on TWO terms -- its input, the
                                             activation E2, rounded to the one fp16 value that is parked for the backward anyway,
                                             times W3 hi + lo: 2/9 fewer MFMAs, pre-activations of fe.net.2 known to ~1e-4 of their
                                             scale (an error that is independent from edge to edge and averages out in agg and in
                                             every gradient sum).  Other values, and 1 together with edge scalars: error -8. */
} MpgEdgeFwd;
#define MPG_EDGE_SCALARS 2
int mpg_edge_fwd(const MpgEdgeFwd* p, void* stream);

/* mpg_edge_fwd_fn: mpg_edge_fwd with the node network as the EPILOGUE of every workgroup -- one launch for
 *   A = fe(cat(x_i, x_j)) ; A * mask ; sum / mean ; x = fn(cat((A, x)))       (mpgan/model.py:256-279)
 * The workgroup that has aggregated a jet's 32 receivers runs the three layers of `fn` (the chain `c`, as mpg_chain takes
 * it) on them straight from LDS; agg reaches memory only as a by-product (p->agg, for fn.net.0's weight gradient; NULL =
 * not kept).  `c` is the chain mpg_chain would be given for the same call -- K1 = 192 columns of agg (c->A is not read:
 * the rows come from the kernel) followed by the node columns c->A2 [B*N, K - 192], three layers, outputs and dropout
 * sites as there -- and the results are bit-identical to mpg_edge_fwd followed by mpg_chain.
 * Covered: SC = 1, no edge scalars, fp16 images, layer widths K <= 224 -> N0, N1 in (224, 256] -> any N2 <= 256, no
 * gates / residuals / input dropout, one dropout mode for all sites.  Anything else returns MPG_FN_NA without
 * launching: the caller then runs the two launches.
 * `c2` (or NULL): one more mpg_chain call that runs behind fn on the same rows, in the same launch -- the NEXT MPLayer's
 * layer-1 node terms a | c = [W1a ; W1c] x + [b1 ; 0] of the rows fn has just produced (one layer, K = c's last N <= 32,
 * input c2->A = c->L[2].out), so that the next layer starts with its edge launch (mpgan/model.py:511-512: the loop over
 * mp_layers). */
#define MPG_FN_NA (-100)
int mpg_edge_fwd_fn(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, void* stream);

/* mpg_knn_sets: the neighbour sets of MPLayer._getA_knn (mpgan/model.py:319-381) as bit masks for the fused edge kernels.
 * Per jet: d(i, j) = || s_j x_j - x_i + 1e-12 || over the F node features, s_j = (1 - 1e4) mask_j + 1e4 (:333-335: 1 for
 * a real sender, 1e4 for a zero-masked one); the senders of receiver i are those of rank [first, first + k) in ascending distance
 * (first = 0 with self loops, 1 without; equal distances in index order).  nbr is [B*N][ceil(N/32)] words.  Running
 * the fully-connected kernels over all N senders with these bits as a per-edge factor gives the reference's
 * gather / fe / sum over the k gathered neighbours (mean: agg_scale = 1/k); N <= 192. */
int mpg_knn_sets(const float* x, int ldx, const float* mask, int B, int N, int F, int k, int self_loops,
                 unsigned int* nbr, void* stream);

/* mpg_edge_bwd: autograd backward of the same span, data path.  Given dagg = dL/dagg and the
 * forward's sign words it produces
 *   da [SC, B*N, 96]  (partial over sender chunks),  dc [RB, B*N, 96]  (partial over receiver blocks of 32)
 * from stageE2 -- E2 = fe.net.1's output as mpg_edge_fwd parked it (required: its signs are phi'(Z2); nothing of the
 * forward is recomputed) -- and, when stageZ2 is non-NULL (weight gradients wanted), parks dZ2 = dL/d(fe.net.1's
 * pre-activation) as fp16 fragments [B*RB*N blocks][10][64 lanes][8] for mpg_edge_dw.  dZ2 is parked in units of
 * 2^-e of its (jet, receiver block) -- gradients have any magnitude, fp16 has 30 binades -- and gexp[b*RB + rb] = e
 * says which (e is chosen from max |dagg| of the block's receivers).
 * All images are fp16 (f16 must be 1, error -8): W3Timg / W2Timg the images of the transposed weights packed with the
 * forward's scales (dscale * 64, dscale * 16); W2img and b2 are not read.  The two gradient products run as two fp16
 * terms (image hi + lo times the gradient rounded to fp16 in a per-sender unit).  A sender chunk (ceil(N / SC) senders) may hold at most 160 senders (error -6: raise SC), and
 * the staging buffers must stay below 2 GiB (error -7). */
typedef struct MpgEdgeBwd {
    const float* a; const float* c; int ld_ac; const float* mask;
    const float* dagg; int ld_dagg;
    const unsigned int* sign3;            /* [B*RB*N][3][64] from mpg_edge_fwd                   */
    const void* W2img; const void* W3Timg; const void* W2Timg;
    const float* b2;
    float* da; float* dc;
    const void* stageE2;                  /* from mpg_edge_fwd (required)                        */
    void* stageZ2;                        /* optional output for mpg_edge_dw                     */
    int B, N, SC;
    float alpha, agg_scale;
    const uint64_t* seed; uint32_t tag_base, thr; float dscale;
    int f16;
    const unsigned int* nbr;              /* as MpgEdgeFwd.nbr */
    int* gexp;                            /* [B*RB] gradient-unit exponents of the parked dZ2 (required with stageE2/stageZ2) */
    const float* es; const float* wq;     /* as MpgEdgeFwd (NULL = none) */
    float* des;                           /* with es: dL/des, same layout (rows of skipped -- zero-masked -- senders are not written:
                                             clear it first) */
    float* daq;                           /* with es: [SC][B*N][MPG_EDGE_SCALARS][96] = sum_j es(i, j) * dZ1(i, j) per receiver;
                                             summed over its rows it is the gradient of wq */
    const int* order;                     /* as MpgEdgeFwd.order */
} MpgEdgeBwd;
int mpg_edge_bwd(const MpgEdgeBwd* p, void* stream);

/* mpg_edge_bwd_fn: mpg_edge_bwd with EPILOGUE chains on every workgroup's own jet -- one launch for the backward of an
 * MPLayer (mpgan/model.py:256-279) from dagg down to its input gradient, and on through the node network of the layer below:
 *   cdx  the "dx from da | dc" chain mpg_chain would be given behind this call: one layer, input A = p->da (96 columns) | A2 =
 *        p->dc, the stacked transposed W1 image, residual = the node path's dx, output dx [B*N, F <= 32];
 *   cnx  (or NULL) the node network's input-gradient chain of the NEXT-LOWER layer (three transposed layers with the gates of
 *        its forward, input = those dx rows with the trailing dropout's gate, outputs dz2, dz1, [dagg | dx]) as mpg_chain
 *        takes it.
 * A workgroup runs them on its 32 nodes when its own sender loop is done, while the fullest jets' workgroups are still in
 * theirs.  Results are bit-identical to mpg_edge_bwd followed by the mpg_chain calls.
 * Covered: a whole jet per workgroup (N <= 32, SC = 1), no edge scalars, bf16 images, one dropout mode; cnx widths K <= 32 ->
 * (224, 256] -> (224, 256] -> N <= 256.  Anything else returns MPG_FN_NA without launching. */
int mpg_edge_bwd_fn(const MpgEdgeBwd* p, const MpgChain* cdx, const MpgChain* cnx, void* stream);

/* mpg_edge_dw: weight gradients of fe.net.1 / fe.net.2 (and their biases) from the fragments parked by
 * mpg_edge_bwd:  dW3 = dscale * sum_e dZ3 E2^T [192,160], dW2 = dscale * sum_e dZ2 E1^T [160,96],
 * db3 = sum_e dZ3 [192], db2 = sum_e dZ2 [160]; E1 and dZ3 are rebuilt from a, c, dagg and the sign words.
 * ONE fp16 term per product: every operand a single fp16 value -- the parked ones as they were parked, the rebuilt ones rounded
 * once, in per-block dithered units so that the roundings are independent from edge to edge and average out over the ~1e5 edges a
 * weight gradient sums (csrc/edge_dw.hip) --, everything in ONE gradient unit 2^-min(gexp) for the launch.
 * `part` is scratch of nwg * 46,432 floats (per-workgroup partial sums); the nwg workgroups share the B*RB*N blocks
 * in runs of R consecutive senders (R = the largest divisor of N up to 6), and each may take at most 64 blocks, i.e.
 * R * ceil(B*RB*N / R / nwg) <= 64 (error -5 otherwise). */
typedef struct MpgEdgeDw {
    const float* a; const float* c; int ld_ac; const float* mask;
    const float* dagg; int ld_dagg;
    const unsigned int* sign3;
    const void* stageE2; const void* stageZ2;
    float* part; int nwg;
    float* dW3; float* dW2; float* db3; float* db2;
    int accumulate;                       /* add to dW3 / dW2 / db3 / db2 instead of overwriting */
    int B, N;
    float alpha, agg_scale;
    const uint64_t* seed; uint32_t tag_base, thr; float dscale;
    int f16;                              /* must be 1 */
    const unsigned int* nbr;              /* as MpgEdgeFwd.nbr */
    const int* gexp;                      /* [B*RB] from mpg_edge_bwd */
    const float* es; const float* wq;     /* as MpgEdgeFwd (NULL = none): E1 is rebuilt with them */
    int defer_reduce;                     /* 1: leave the per-workgroup partials in `part`; mpg_splitk_reduce_group_dw adds them up */
} MpgEdgeDw;
int mpg_edge_dw(const MpgEdgeDw* p, void* stream);
/* The reduction of an mpg_edge_dw launch that ran with defer_reduce = 1 AND the grouped split-K reduction of n (0 ... MPG_GROUP_MAX)
 * jobs of an earlier mpg_gemm_wgrad_group on the same stream, as ONE launch: the layer's edge-network and dense weight gradients
 * land in their buffers together, and the weight-gradient tail of a layer's backward is one dependent launch shorter.  Same sums
 * in the same order as mpg_edge_dw's own reduction + mpg_splitk_reduce_group. */
int mpg_splitk_reduce_group_dw(const MpgReduceJob* jobs, int n, const MpgEdgeDw* p, void* stream);

/* ---- attention core of GAPT's MAB ---------------------------------------------------------------
 * mpg_attn_fwd / mpg_attn_bwd: per (jet b, head h), d = E / H:
 *     P = softmax_s( q_h k_h^T / sqrt(d)  with keys s where ignore[b,s] != 0 at -inf ),   o_h = P v_h
 * i.e. what nn.MultiheadAttention does between its in- and out-projection as called by MAB.forward
 * (gapt/model.py:127-129).  q [B*L, ldq], k/v [B*S, ldk/ldv], o [B*L, ldo] hold the H heads side by
 * side (head h = columns h*d .. h*d+d-1); P (B*H*L*S floats, layout private to the pair of kernels) is written by fwd and read by bwd;
 * ignore [B*S] floats (1 = padded key) or NULL.  bwd takes d_o = dL/do and writes dq, dk, dv. */
typedef struct MpgAttn {
    const float* q; const float* k; const float* v; int ldq, ldk, ldv;
    const float* ignore;
    float* o; int ldo;
    float* P;
    const float* d_o;
    float* dq; float* dk; float* dv; int lddq, lddk, lddv;
    int B, L, S, H, d;
} MpgAttn;
int mpg_attn_fwd(const MpgAttn* p, void* stream);
int mpg_attn_bwd(const MpgAttn* p, void* stream);

/* ---- the per-jet pieces around the message-passing layers ----------------------------------------
 * mpg_rank_mask: MPGenerator._get_mask, mask_c branch (mpgan/model.py:689-699; GAPT_G :255-258): with
 * n_b = int(labels[b] * N), mask[b, i] = 1 for the n_b particles of jet b with the smallest first feature
 * x[b*ld_jet + i*ld_part] (rank = argsort(argsort(.)); ties by index), else 0.  mask is [B, N] contiguous; `ignore`
 * (or NULL) receives 1 - mask in the same pass: the key mask GAPT's attention blocks take (_attn_mask, gapt/model.py:194-202). */
int mpg_rank_mask(const float* x, int ld_jet, int ld_part, const float* labels, int ld_lab, int B, int N,
                  float* mask, float* ignore, void* stream);

/* mpg_jet_order: the jets of a batch by decreasing number of unmasked particles (ties by index), order[0] the fullest --
 * MpgEdgeFwd.order / MpgEdgeBwd.order.  The edge kernels take a workgroup per jet and as long as the jet has senders; when a
 * launch has more workgroups than the chip has CUs (the discriminator's real + generated batch), handing them out heaviest
 * first is the classic longest-processing-time rule: 136 -> 113 us for 512 gluon-like jets.  No effect on any result.
 * B + N + 2 + ceil(B / 64) * (N + 1) <= 16384 (error -2). */
int mpg_jet_order(const float* mask, int B, int N, int* order, void* stream);

/* mpg_gen_tail_fwd / _bwd: MPNet._final_activation (:533-538) + MPGenerator._final_mask (:741-757) on V = B*N rows:
 * out[v, 0:F] = act(y[v, 0:F]) (act 0 none, 1 tanh, 2 sigmoid), out[v, F] = mask[v] - 0.5 when mask != NULL;
 * backward dy = dout[:, 0:F] * act'(out).  Row strides ldy / ldo / ldd let the output land inside a larger batch. */
int mpg_gen_tail_fwd(const float* y, int ldy, const float* mask, float* out, int ldo, int V, int F, int act, void* stream);
int mpg_gen_tail_bwd(const float* dout, int ldd, const float* out, int ldo, float* dy, int ldy, int V, int F, int act,
                     void* stream);

/* mpg_disc_head_fwd / _bwd: MPDiscriminator._post_mp (masked sum or mean over the particles, :812-829), fnd_layer =
 * one Linear(F -> 1) followed by Dropout (LinearNet, :77-83) and the final sigmoid (:537); also GAPT_D's
 * final_fc + sigmoid (gapt/model.py:344) with N = 1.
 *   fwd:  out[b] = act( keep_b * ( pool_b( sum_i mask[b,i] y[b,i,:] ) . w + bias ) )
 *   bwd:  dy[b,i,f], and dw / db of the Linear (added to when accumulate), from
 *         loss < 0 : gout[b] = dL/dout[b] handed in by autograd
 *         loss >= 0: the loss named (0 ls, 1 og, 2 w, 3 hinge: calc_D_loss / calc_G_loss, train.py:331-395, :465-476),
 *                    jets [0, n_real) scored as real and the rest as generated (gen_step: all as real, generator
 *                    form), each term times inv_count; loss_out receives the sum of the terms.
 * aux [2B] and pooled [B,F] are scratch written by fwd and read by bwd. */
typedef struct MpgDiscHead {
    const float* y; int ldy;              /* [B, N, F], row (particle) stride ldy                  */
    const float* mask;                    /* [B, N] or NULL                                        */
    const float* w; const float* bias;    /* Linear(F -> 1): weight [F], bias [1] or NULL          */
    int B, N, F;
    int mean;                             /* pooling: 0 sum, 1 mean (mask sum + 1e-12, or N)       */
    int sigmoid;
    const uint64_t* seed; uint32_t tag, thr; float dscale;   /* dropout on the Linear's output     */
    float* out;                           /* [B]                                                   */
    float* pooled; float* aux;
    int loss, gen_step, n_real; float inv_count;
    const float* gout;
    float* terms; float* loss_out;        /* [B] scratch; scalar                                   */
    float* dy; int ld_dy;                 /* [B, N, F] or NULL                                     */
    float* dw; float* db; int accumulate;
} MpgDiscHead;
int mpg_disc_head_fwd(const MpgDiscHead* p, void* stream);
int mpg_disc_head_bwd(const MpgDiscHead* p, void* stream);
/* mpg_disc_head_loss: forward, the named loss and the whole backward of the head in one pass -- p->loss >= 0 with its
 * fields set as for mpg_disc_head_bwd: a jet's loss gradient needs nothing but its own output, so the wave that has pooled
 * a jet goes straight on to dy (one launch + the small reduction for the loss value and dw / db, instead of mpg_disc_head_fwd
 * + mpg_disc_head_bwd).  Same results as the two calls. */
int mpg_disc_head_loss(const MpgDiscHead* p, void* stream);

/* mpg_bridge_fwd / mpg_bridge_bwd: the rows between a GAPT generator's last attention block and the discriminator's first
 * one as ONE launch each way (csrc/bridge.hip) -- gen's final_fc (gapt/model.py:248, :263: Linear K -> F, no activation) and
 * final tanh (:265), then disc's input_embedding (:300-302, :336-339: Linear F -> E, LeakyReLU, dropout), which otherwise run
 * as mpg_gemm + mpg_gen_tail_fwd + mpg_gemm (and mpg_gate + mpg_gemm + mpg_gen_tail_bwd + mpg_gemm on the way back):
 *   rows >= row0 (generated):  feat = act1(W1 x + b1), written to feat ;  rows < row0 (real jets): feat is read
 *   every row:                 e = dropout(lrelu(W2 feat + b2))          (e == NULL: stage 1 alone)
 * x [M - row0, K], feat [M, F], e [M, E]; K = E = 64, F in 1..4 or 8; act1: mpg_gen_tail_fwd's codes.
 * bwd: g2 = ge through stage 2's dropout and LeakyReLU (what mpg_gate gives: the operand of W2's weight gradient), all rows;
 * for rows >= row0, when g1 or dx is given: g1 = (g2 W2 + gfeat) act1'(feat) (the operand of W1's weight gradient) and
 * dx = g1 W1.  Any of g2, g1, dx, gfeat may be NULL. */
typedef struct MpgBridge {
    const float* x; int ldx;
    const float* W1; const float* b1;       /* [F, K], [F] or NULL                                   */
    int act1;
    float* feat; int ldf;
    int M, row0, K, F, E;
    const float* W2; const float* b2;       /* [E, F], [E] or NULL                                   */
    int act2; float alpha;
    const uint64_t* seed; uint32_t tag, thr; float dscale;   /* dropout on e (site tag)              */
    float* e; int lde;
} MpgBridge;
typedef struct MpgBridgeBwd {
    const float* ge; int ldge;              /* [M, E] gradient with respect to e                     */
    const float* e; int lde;                /* stage 2's saved output                                */
    const float* feat; int ldf;
    const float* gfeat; int ldgf;           /* [M - row0, F] further gradient into feat, or NULL     */
    const float* W1; const float* W2;
    int M, row0, K, F, E, act1, act2; float alpha;
    const uint64_t* seed; uint32_t tag, thr; float dscale;
    float* g2; int ldg2;                    /* [M, E]                                                */
    float* g1; int ldg1;                    /* [M - row0, F]                                         */
    float* dx; int lddx;                    /* [M - row0, K]                                         */
} MpgBridgeBwd;
int mpg_bridge_fwd(const MpgBridge* p, void* stream);
int mpg_bridge_bwd(const MpgBridgeBwd* p, void* stream);

/* mpg_mab_fwd / mpg_mab_bwd: one launch per MAB.forward (gapt/model.py:124-139) and one for its backward, for sets of at
 * most 160 tokens (L, S <= 32: a wave or two per jet; 33 ... 160 -- --num-hits 150, setup_training.py:415 --: a workgroup per
 * jet, a wave per tile of 32 tokens, key tiles walked with a running maximum / sum, projections made once per tile and
 * shared through LDS), E in {32, 64}, heads of 16 features, with or without the
 * two layer norms (csrc/mab.hip; anything else runs block by block through mpg_gemm / mpg_attn_* / mpg_gate):
 *   q = x Wq' + bq, k|v = y Wkv' + bkv ; o = softmax(q k' / 4 + key mask) v per head ; za = x + o Wo' + bo ;
 *   z = dropout(za, site tag) ; u = z Wf' + bf ; out = dropout(z + dropout(act(u), site tag+1), site tag+2)
 * x [B*L, E] queries, y [B*S, E] keys/values (y == x: self-attention), ignore [B*S] floats (1 = padded key) or NULL.
 * Win / Wo / Wf: fp16 hi|lo images (mpg_pack_weights, scale wscale) of in_proj_weight [3E, E], out_proj.weight and
 * ff.net.0.weight [E, E]; activations are split as ascale * value.  fwd writes out and, when given, save_o (attention
 * output before the out-projection) and save_z [B*L, E] -- the two activations the weight gradients and the backward need.
 * bwd recomputes q, k, v, P and u from x, y and save_z, and writes the input gradients dx (dy when y != x) and the three
 * pre-activation gradients whose products with (x | y, save_o, save_z) are the weight gradients:
 *   dq [B*L, :E] / dk, dv [B*S] (row strides lddq / lddkv), dza, du [B*L, E].
 * WinT / WoT / WfT: bf16 images of the transposed weights ([E, 3E], [E, E], [E, E], scale 1). */
typedef struct MpgMab {
    const float* x; int ldx;
    const float* y; int ldy;
    const float* ignore;
    const void* Win; const float* bin;
    const void* Wo; const float* bo;
    const void* Wf; const float* bf;
    int B, L, S, E, H;
    float alpha; int ff_act;
    const uint64_t* seed; uint32_t tag; uint32_t thr_mab; float sc_mab; uint32_t thr_ff; float sc_ff;
    float wscale, ascale;
    float* out; int ldo;
    float* save_o; float* save_z;
    const void* WinT; const void* WoT; const void* WfT;
    const float* dout; int lddout;
    float* dx; int lddx; float* dy; int lddy;
    float* dq; int lddq; float* dk; float* dv; int lddkv;
    float* dza; float* du;
    /* layer_norm=True (gapt/model.py:118-120, :131-136): nn.LayerNorm(E) behind each residual, or all NULL.  za -> norm1 -> dropout;
     * z + ff(z) -> norm2 -> dropout.  fwd keeps za (before norm1) in save_za for the backward; bwd writes, per row, the
     * gradient with respect to each norm's OUTPUT (dn1, dn2: their column sums are the norms' bias gradients) and that times
     * the normalised input (gn1, gn2: column sums = weight gradients).  One wave per jet (the two-wave kernels step aside). */
    const float* ln1_w; const float* ln1_b; const float* ln2_w; const float* ln2_b; float ln_eps;
    float* save_za;
    float* dn1; float* gn1; float* dn2; float* gn2;
} MpgMab;
int mpg_mab_fwd(const MpgMab* p, void* stream);
int mpg_mab_bwd(const MpgMab* p, void* stream);
/* mpg_mab_chain_fwd: up to MPG_MAB_CHAIN_MAX self-attention blocks applied one after the other -- the loop over the SABs of
 * GAPT_G / GAPT_D (gapt/model.py:261-262, :341-342) -- in ONE launch: blk[b] is the argument block mpg_mab_fwd would be given
 * for block b (y == x; blk[b].x == blk[b-1].out; one shape, one key mask).  A wave keeps its jet's rows in registers from
 * block to block; each block still writes its out / save_o / save_z.  Same results as the n single launches. */
#define MPG_MAB_CHAIN_MAX 4
typedef struct MpgMabChain { MpgMab blk[MPG_MAB_CHAIN_MAX]; int n; } MpgMabChain;
int mpg_mab_chain_fwd(const MpgMabChain* c, void* stream);

/* mpg_layernorm_fwd / _bwd: nn.LayerNorm(E) over the rows of x [M, E] -- MAB.norm1 / norm2 of GAPT with layer_norm
 * (gapt/model.py:118-120, :131-136).  fwd writes y and stats [M][mean, rstd]; bwd writes dx and, through `part`
 * (scratch of nwaves * 2 * E floats, nwaves a multiple of 4 = waves of the launch), dw = sum_rows g * xhat and
 * db = sum_rows g in a fixed summation order (added to when accumulate). */
int mpg_layernorm_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, float* stats, int M,
                      int E, float eps, void* stream);
int mpg_layernorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* w, const float* stats, float* dx,
                      int lddx, float* part, int nwaves, float* dw, float* db, int accumulate, int M, int E, void* stream);

/* mpg_batchnorm_*: nn.BatchNorm1d(F) over the rows of x [M, F] -- LinearNet's optional normalisation behind each
 * LeakyReLU (mpgan/model.py:58-60, :80-81).  stats writes the batch mean and BIASED variance (two passes; `part` is scratch
 * of nchunk * F floats); apply normalises with whatever mean / var it is given (batch statistics in training, running
 * statistics in eval) and the affine w, b (NULL = 1, 0); bwd (training statistics) writes dx and dw = sum g xhat,
 * db = sum g (added to when accumulate) with `part` of 2 * nchunk * F and `sums` of 2 * F floats as scratch. */
int mpg_batchnorm_stats(const float* x, int ldx, int M, int F, float* part, int nchunk, float* mean, float* var, void* stream);
int mpg_batchnorm_apply(const float* x, int ldx, const float* mean, const float* var, const float* w, const float* b, float eps,
                        float* y, int ldy, int M, int F, void* stream);
int mpg_batchnorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* mean, const float* var, const float* w, float eps,
                      float* part, int nchunk, float* sums, float* dx, int lddx, float* dw, float* db, int accumulate, int M, int F,
                      void* stream);

/* ---- optimisers --------------------------------------------------------------------------------
 * One launch over one flat buffer of n parameters; `gscale` multiplies the gradient first (1/world after a
 * summing all-reduce).  They replace torch.optim.*.step() as the reference builds them (setup_training.py:1511-1523):
 * mpg_rmsprop  (--optimizer rmsprop, the default): v = alpha v + (1-alpha) g^2; p -= lr g / (sqrt(v) + eps)
 * mpg_adam     (--optimizer adam; weight_decay 5e-4 there): torch.optim.Adam with L2 weight decay; `step` is a
 *              device float holding the number of steps taken so far (advanced by the call)
 * mpg_adadelta (--optimizer adadelta): torch.optim.Adadelta (rho 0.9, eps 1e-6 by default).
 * zero_grad != 0: g is cleared behind its last use -- optimizer.zero_grad() of the NEXT train_D / train_G (train.py:419,
 * :494) without a launch of its own.
 * counter != NULL: *counter += counter_add by the same launch -- the device-resident dropout / noise seed moved on to the
 * next iteration's value by the iteration's last launch (no kernel may be reading it on another stream); NULL: nothing. */
int mpg_rmsprop(float* p, float* g, float* v, uint64_t n, float lr, float alpha, float eps, float gscale, int zero_grad,
                uint64_t* counter, uint64_t counter_add, void* stream);
int mpg_adam(float* p, float* g, float* m, float* v, float* step, uint64_t n, float lr, float beta1,
             float beta2, float eps, float weight_decay, float gscale, int zero_grad, uint64_t* counter, uint64_t counter_add,
             void* stream);
int mpg_adadelta(float* p, float* g, float* v, float* u, uint64_t n, float lr, float rho, float eps,
                 float gscale, int zero_grad, uint64_t* counter, uint64_t counter_add, void* stream);

/* mpg_normal: out[i] = mean + std * z_i with z ~ N(0, 1) -- the generator's input noise (get_gen_noise, train.py:100-141:
 * torch.randn * sd) from a counter-based stream keyed by the device-resident 64-bit `seed` (the dropout seed, advanced
 * once per iteration) and a site `tag`: a captured hipGraph draws fresh values on every replay, and torch's generator
 * (whose graph-safe state costs two fill launches in front of every replay) is not involved. */
int mpg_normal(float* out, uint64_t n, const uint64_t* seed, uint32_t tag, float mean, float std, void* stream);
/* mpg_normal_rank_mask: mpg_normal over a [B, N, L] noise tensor and mpg_rank_mask of its first feature (out[:, :, 0]) in one
 * launch -- same values, same masks; mask / ignore [B, N] (ignore = 1 - mask, or NULL), labels read at stride ld_lab.  N L even. */
int mpg_normal_rank_mask(float* out, int B, int N, int L, const uint64_t* seed, uint32_t tag, float mean, float std,
                         const float* labels, int ld_lab, float* mask, float* ignore, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPGAN_AMD_H */
