#!/bin/bash
# build variants of bwd_bench locally: tools/ubench/run_bwd.sh "<flags1>" "<flags2>" ... ; binaries bwd_bench_0, _1, ...
cd /root/repo
i=0
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -DMPG_SINGLE_VARIANT=0 $f tools/ubench/bwd_bench.hip mpgan_amd/csrc/edge.hip -o tools/ubench/bwd_bench_$i 2>&1 | grep -E "error|spill" -A5 | head -10 &
  i=$((i+1))
  if (( i % 4 == 0 )); then wait; fi
done
wait
