// The data-gradient kernel with its epilogue chains (mpg_edge_bwd_fn, see edge_bwd_fn.hip),
// dropout mode 2, data path only.
#include "edge_bwd1_impl.h"

int mpg_edge_bwd_fn_d2w0(const MpgEdgeBwd* p, const MpgChain* cdx, const MpgChain* cnx, int epi, hipStream_t st) {
    return b1_launch_fn<2, false>(p, cdx, cnx, epi, st);
}
