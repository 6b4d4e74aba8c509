#!/bin/bash
# build first: a profiled python must never compile (see tools/final_profiles.sh)
python3 $GRAFT_REPO_ROOT/__graft_entry__.py || exit 1
# SQ counters of the edge kernels (tools/kbwd.py workload): MFMA busy cycles, waits, VALU activity.
cd /tmp && export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/final/pmc_sq
rm -rf $o
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $o -o p -- python3 $GRAFT_REPO_ROOT/tools/kbwd.py > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
for k in "edge_fwd_kernel<0" "edge_fwd_kernel<2" "edge_bwd_kernel<0, true" "edge_bwd_kernel<2, true" "edge_dw_kernel<0" "edge_dw_kernel<2"; do
  echo "== $k"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $o "$k"
done
