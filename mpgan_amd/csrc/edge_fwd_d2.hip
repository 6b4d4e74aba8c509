// The one-bit-dropout (p = 1/2) variants of the forward edge kernel (see edge.hip).
#include "edge_fwd1_impl.h"

int mpg_edge_fwd_d2(const MpgEdgeFwd* p, hipStream_t st) { return f1_launch<2>(p, st); }
