// The fused edge forward + node network (mpg_edge_fwd_fn, see edge_fwd_fn.hip), dropout mode 0, with the backward's by-products.
#include "edge_fwd1_impl.h"

int mpg_edge_fwd_fn_d0s1(const MpgEdgeFwd* p, const MpgChain* c, const MpgChain* c2, bool sl, hipStream_t st) {
    if (p->SC > 1 && !fwd_fn_eight_waves()) return MPG_FN_NA;   // (sender chunks: the eight-wave form only)
    return fwd_fn_eight_waves() ? f1_launch_fn<0, true>(p, c, c2, sl, st) : f2_launch_fn<0, true>(p, c, c2, sl, st);
}
