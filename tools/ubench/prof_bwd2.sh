#!/bin/bash
# SQ counters of both data-gradient kernels under the A/B harness (two counter passes); results in gpurun_out/$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-bwd2prof}
BIN=${2:-$R/tools/ubench/bwd2_bench_0}
ARGS=${3:-"256 1 1"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  o=$OUT/pass$i
  rm -rf $o
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $o -o p -- $BIN $ARGS > $o.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then tail -5 $o.log; exit 1; fi
  for k in "edge_bwd_kernel"; do
    echo "== pass $i $k"; python3 $R/tools/pmc_summary.py $o "$k"
  done
  i=$((i+1))
done
