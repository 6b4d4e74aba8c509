// The weight-gradient kernel of the fused edge network with edge scalars (MpgEdgeDw.es): the same source as edge_dw.hip,
// instantiated for MPG_EDGE_SCALARS columns, compiled beside it.
#define MPG_DW_Q_UNIT
#include "edge_dw.hip"
