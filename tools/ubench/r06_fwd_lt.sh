#!/bin/bash
# Round 6 go/no-go: the eight-wave forward with layer 3 (LT=1) / layers 3 and 2 (LT=3) on two terms, against the three-term form
# (LT=0); p = 0 (v0) and p = 1/2 (v2); full jets and ragged (gluon-like) jets; *_old = before the scratch fixes of this round.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for rep in 1 2; do
for b in v0lt0_old v0lt0 v0lt1 v0lt3 v2lt0_old v2lt0 v2lt1 v2lt3; do
  timeout -k 5 60 tools/ubench/fwd_bench_$b 256 || exit 1
  timeout -k 5 60 tools/ubench/fwd_bench_$b 256 r || exit 1
done
done
for b in v0lt0s v0lt1s v0lt3s v2lt0s v2lt1s v2lt3s; do
  timeout -k 5 60 tools/ubench/fwd_bench_$b 256 r || exit 1
done
