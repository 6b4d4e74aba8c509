// The data-gradient kernel with edge scalars (MpgEdgeBwd.es), dropout mode 1 (see edge_bwd2.hip).
#include "edge_bwd2_impl.h"

int mpg_edge_bwd_q1(const MpgEdgeBwd* p, hipStream_t st) { return b2_launch<1, MPG_EDGE_SCALARS>(p, st); }
