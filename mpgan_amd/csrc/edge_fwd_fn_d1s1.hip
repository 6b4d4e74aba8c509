// The fused edge forward + node network (mpg_edge_fwd_fn, see edge_fwd_fn.hip), dropout mode 1, with the backward's by-products.
#include "edge_fwd1_impl.h"

int mpg_edge_fwd_fn_d1s1(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, bool sl, hipStream_t st) {
    return f1_launch_fn<1, true>(p, c, c2, sl, st);
}
