// The forward edge kernel with edge scalars (MpgEdgeFwd.es: delta_r, row-tiled conditioning columns), dropout mode 1 (see edge.hip).
#include "edge_fwd2_impl.h"

int mpg_edge_fwd_q1(const MpgEdgeFwd* p, hipStream_t st) { return p->two_term ? -8 : f2_launch<1, MPG_EDGE_SCALARS>(p, st); }   // (three-term products only)
