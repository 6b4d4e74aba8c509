#!/bin/bash
# build variants of w2_bench locally: tools/ubench/run_w2.sh "<flags1>" "<flags2>" ... ; binaries w2_bench_0, _1, ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include $f tools/ubench/w2_bench.hip -o tools/ubench/w2_bench_$i 2>&1 | grep -E "error" -A5 | head -10 &
  i=$((i+1))
done
wait
