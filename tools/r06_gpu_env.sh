#!/bin/bash
# same-box matrix of the scheduling switches (environment only), current tree
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06_env; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-secondary --no-cpu-baseline --no-roofline > $O/bench_$name.json 2> $O/bench_$name.log || { tail -3 $O/bench_$name.log; return 1; }
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name: %.0f jets/s %.4f ms' % (d['value'], d['ms_per_step']))"
}
for rep in 1 2 3; do
  run base_$rep X=1 || exit 1
  run wgrad_side0_$rep MPG_WGRAD_SIDE=0 || exit 1
  run gen_ahead0_$rep MPG_GEN_AHEAD=0 || exit 1
  run gen_ahead_early_$rep MPG_GEN_AHEAD_LATE=0 || exit 1
  run dw_reduce_own_$rep MPG_DW_REDUCE_GROUPED=0 || exit 1
  run parts0_$rep MPG_PARTS=0 || exit 1
done
