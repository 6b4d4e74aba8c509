#!/bin/bash
# build bwd2_bench for the dropout modes given (default "0 2"): tools/ubench/run_bwd2.sh [modes] [extra flags] [suffix]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for d in ${1:-0 2}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -DMPG_SINGLE_VARIANT=$d $2 tools/ubench/bwd2_bench.hip mpgan_amd/csrc/edge.hip -o tools/ubench/bwd2_bench_$d$3 2>&1 | grep -E "error" -A5 | head -20 &
done
wait
ls -la tools/ubench/bwd2_bench_*
