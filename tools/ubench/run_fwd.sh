#!/bin/bash
# build variants of fwd_bench locally: tools/ubench/run_fwd.sh "<flags1>" "<flags2>" ... ; binaries fwd_bench_0, _1, ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include $f tools/ubench/fwd_bench.hip -o tools/ubench/fwd_bench_$i 2>&1 | grep -E "error|scratch" | head -5 &
  i=$((i+1))
done
wait
