// The byte-threshold-dropout variants of the forward edge kernel (see edge.hip).
#include "edge_fwd2_impl.h"

int mpg_edge_fwd_d1(const MpgEdgeFwd* p, hipStream_t st) { return f2_launch<1>(p, st); }
