#!/bin/bash
# build variants of dw_bench locally: tools/ubench/run_dw.sh "<flags1>" "<flags2>" ... ; binaries dw_bench_0, _1, ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include $f tools/ubench/dw_bench.hip -o tools/ubench/dw_bench_$i 2>&1 | grep -E "error" -A5 | head -10 &
  i=$((i+1))
done
wait
