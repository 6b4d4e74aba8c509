#pragma once
// Grouped one-term fp16 weight gradients (wgrad2.hip): an experiment outside the product library.
#include <hip/hip_runtime.h>
#include "../../include/mpgan_amd.h"

// dW[n, k] = out_scale * sum_m dy[m, n] x[m, k]  (n < N, k < K, m < M), written as split partials part[z][n][ldp] (ldp = K + hb;
// column K = the column sums of dy when hb)
struct W2Job {
    const float* dy; const float* x; float* part; long long split_stride;
    int ldy, ldx, ldp, N, K, M; float out_scale; int hb;
    int dy_vec;   // dy rows are 16-byte aligned float4 groups (else element loads)
};
struct W2Group { W2Job j[MPG_GROUP_MAX]; int splitk[MPG_GROUP_MAX]; int wg0[MPG_GROUP_MAX + 1]; int n; };

int mpg_wgrad2_launch(const W2Group* G, hipStream_t st);
