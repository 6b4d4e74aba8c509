// The byte-threshold-dropout variants of the data-gradient kernel (see edge_bwd2.hip).
#include "edge_bwd1_impl.h"

int mpg_edge_bwd_d1(const MpgEdgeBwd* p, hipStream_t st) { return b1_launch<1>(p, st); }
