// The data-gradient kernel with its epilogue chains (mpg_edge_bwd_fn, see edge_bwd_fn.hip),
// dropout mode 1, parking dZ2 for the weight-gradient kernel.
#include "edge_bwd1_impl.h"

int mpg_edge_bwd_fn_d1w1(const MpgEdgeBwd* p, const MpgChain* cdx, const MpgChain* cnx, int epi, hipStream_t st) {
    return b1_launch_fn<1, true>(p, cdx, cnx, epi, st);
}
