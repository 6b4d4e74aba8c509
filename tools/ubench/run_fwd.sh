#!/bin/bash
# build variants of fwd_bench locally: tools/ubench/run_fwd.sh "<flags1>" "<flags2>" ... ; binaries fwd_bench_0, _1, ...
cd /root/repo
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DMPG_SINGLE_VARIANT=0 $f tools/ubench/fwd_bench.hip -o tools/ubench/fwd_bench_$i 2>&1 | grep -E "error|scratch" | head -5 &
  i=$((i+1))
done
wait
