// The fused edge forward + node network (mpg_edge_fwd_fn, see edge_fwd_fn.hip), dropout mode 1, without the backward's by-products.
#include "edge_fwd1_impl.h"

int mpg_edge_fwd_fn_d1s0(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, bool sl, hipStream_t st) {
    return f1_launch_fn<1, false>(p, c, c2, sl, st);
}
