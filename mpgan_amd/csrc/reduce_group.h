// The grouped split-K reduction (mpg_splitk_reduce_group) as a device function + its host-side table, shared by gemm.hip and by
// edge_dw.hip, whose per-workgroup reduction can carry a layer's grouped reductions in the same launch (mpg_splitk_reduce_group_dw).
#pragma once
#include "../../include/mpgan_amd.h"

namespace {

struct ReduceGroup { MpgReduceJob j[MPG_GROUP_MAX]; int blk0[MPG_GROUP_MAX + 1]; int n; };

// block `blk` (of R.blk0[R.n]) of 256 threads
__device__ __forceinline__ void splitk_reduce_group_body(const ReduceGroup& R, const int blk) {
    int q = 0;
    while (blk >= R.blk0[q + 1]) ++q;
    const MpgReduceJob& J = R.j[q];
    const int idx = (blk - R.blk0[q]) * 256 + (int)threadIdx.x;
    const int ldp = J.K + J.has_bias;
    if (idx >= J.N * ldp) return;
    const int n = idx / ldp, k = idx % ldp;
    // (four independent partial sums: the loads of a thread are then in flight together; the summation order is fixed)
    const size_t zs = (size_t)J.N * ldp;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 4 <= J.S; z += 4) {
        s0 += J.part[(size_t)z * zs + idx]; s1 += J.part[(size_t)(z + 1) * zs + idx];
        s2 += J.part[(size_t)(z + 2) * zs + idx]; s3 += J.part[(size_t)(z + 3) * zs + idx];
    }
    for (; z < J.S; ++z) s0 += J.part[(size_t)z * zs + idx];
    const float s = (s0 + s1) + (s2 + s3);
    if (k < J.K) { float* d = J.out + (size_t)n * J.ldo + k; *d = s + (J.accumulate ? *d : 0.f); }
    else if (J.bias != nullptr) J.bias[n] = s + (J.accumulate ? J.bias[n] : 0.f);
}

// the launch table of n jobs; returns the number of blocks (0: nothing to do), -1 for a bad n
inline int make_reduce_group(const MpgReduceJob* jobs, int n, ReduceGroup& R) {
    if (n < 0 || n > MPG_GROUP_MAX) return -1;
    R.n = n;
    R.blk0[0] = 0;
    for (int q = 0; q < n; ++q) {
        R.j[q] = jobs[q];
        const int tot = jobs[q].N * (jobs[q].K + jobs[q].has_bias);
        R.blk0[q + 1] = R.blk0[q] + (tot + 255) / 256;
    }
    return R.blk0[n];
}

}  // namespace
