#!/bin/bash
# build first: a profiled python must never compile (see tools/final_profiles.sh)
python3 $GRAFT_REPO_ROOT/__graft_entry__.py || exit 1
# HBM traffic counters of the edge kernels (tools/kbwd.py workload), one counter per pass as the guide prescribes.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$c
  rm -rf $out
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/kbwd.py > $out.log 2>&1 || exit 1
  for k in edge_fwd_kernel edge_bwd_kernel edge_dw_kernel chain_kernel gemm_group_kernel; do
    echo "== $c $k"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out $k
  done
done
