// The forward edge kernel with edge scalars (MpgEdgeFwd.es: delta_r, row-tiled conditioning columns), dropout mode 2 (see edge.hip).
#include "edge_fwd2_impl.h"

int mpg_edge_fwd_q2(const MpgEdgeFwd* p, hipStream_t st) { return p->two_term ? -8 : f2_launch<2, MPG_EDGE_SCALARS>(p, st); }   // (three-term products only)
