#!/bin/bash
# Round 6: what bounds mpg_edge_dw (the twelve-wave kernel)?  The harness with parts of the kernel compiled out (-DMPG_DW_EXP bits:
# 1 consumers idle (no transposing reads, no MFMAs), 2 builders idle, 4 no staged loads (nothing of the parked fragments, rows or sign
# words is fetched), 8 no LDS writes, 16 every prefetch redirected to one cached block) -- results are then wrong; only the time counts.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for rep in 1 2; do
for v in 0 2; do for e in 0 4 16 8 12 1 2; do
  echo "== v$v EXP=$e"
  timeout -k 5 60 tools/ubench/dw_bench_v${v}e${e} 256 | tail -1 || exit 1
  timeout -k 5 60 tools/ubench/dw_bench_v${v}e${e} 512 r | tail -1 || exit 1
done; done; done
