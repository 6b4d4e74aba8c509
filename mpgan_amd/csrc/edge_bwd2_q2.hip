// The data-gradient kernel with edge scalars (MpgEdgeBwd.es), dropout mode 2 (see edge_bwd2.hip).
#include "edge_bwd2_impl.h"

int mpg_edge_bwd_q2(const MpgEdgeBwd* p, hipStream_t st) { return b2_launch<2, MPG_EDGE_SCALARS>(p, st); }
