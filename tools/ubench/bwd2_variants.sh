#!/bin/bash
# run every built bwd2_bench_<mode> over full / ragged / scattered masks, with and without parking, N = 30 and 150
cd $GRAFT_REPO_ROOT
for b in tools/ubench/bwd2_bench_*; do
  echo "== $b"
  for args in "256 1 0" "256 1 1" "256 1 2" "256 0 1" "512 1 1" "16 1 1 150 3" "3 1 2 33 1" "2 0 2 5 1"; do
    timeout -k 5 60 $b $args || { echo "rc=$? for $args"; exit 1; }
  done
done
