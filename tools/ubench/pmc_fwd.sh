#!/bin/bash
# SQ counters of fwd_bench binaries: tools/ubench/pmc_fwd.sh <binary index> ...   (rocprofv3 --pmc, program directly after --)
cd /tmp && export TMPDIR=/tmp
for i in "$@"; do
  o=$GRAFT_REPO_ROOT/gpurun_out/pmc_fwd_$i
  rm -rf $o
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $o -o p -- $GRAFT_REPO_ROOT/tools/ubench/fwd_bench_$i 256 r > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
  echo "== fwd_bench_$i"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $o "edge_fwd"
  o2=$GRAFT_REPO_ROOT/gpurun_out/pmc_fwd_lds_$i
  rm -rf $o2
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $o2 -o p -- $GRAFT_REPO_ROOT/tools/ubench/fwd_bench_$i 256 r > $o2.log 2>&1 || { tail -5 $o2.log; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $o2 "edge_fwd"
done
