// The data-gradient kernel with the node network's input-gradient chain as its prologue (mpg_edge_bwd_fn, see edge_bwd_fn.hip),
// dropout mode 2, parking dZ2 for the weight-gradient kernel.
#include "edge_bwd2_impl.h"

int mpg_edge_bwd_fn_d2w1(const MpgEdgeBwd* p, const MpgChain* c, bool sl, hipStream_t st) { return b2_launch_fn<2, true>(p, c, sl, st); }
