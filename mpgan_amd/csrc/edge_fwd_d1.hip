// The byte-threshold-dropout variants of the forward edge kernel (see edge.hip).
#include "edge_fwd1_impl.h"

int mpg_edge_fwd_d1(const MpgEdgeFwd* p, hipStream_t st) { return f1_launch<1>(p, st); }
