#!/bin/bash
# run every built bwd_bench_<i> (tools/ubench/run_bwd.sh) at B=256, with weight-gradient staging, full and ragged jets
cd $GRAFT_REPO_ROOT
for b in tools/ubench/bwd_bench_*; do
  echo "== $b"
  timeout -k 5 60 $b 256 1 || exit 1
  timeout -k 5 60 $b 256 1 r || exit 1
  timeout -k 5 60 $b 256 0 r || exit 1
done
