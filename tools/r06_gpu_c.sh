#!/bin/bash
# round 6, third GPU call: full suite (default form), experiments on the weight-gradient tail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06_c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.txt 2>&1; rc=$?
tail -8 $O/pytest.txt; cp gpurun_out/parity_bars.txt $O/parity_bars.txt 2>/dev/null
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
echo "pytest rc=$rc"
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-secondary --no-cpu-baseline --no-roofline > $O/bench_$name.json 2> $O/bench_$name.log || return 1
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name:', round(d['value']), 'jets/s', round(d['ms_per_step'],4), 'ms')"
}
for rep in 1 2; do
  run base_$rep X=1 || exit 1
  run skipwgrad_$rep MPG_EXP_SKIP=wgrad || exit 1
  run wt256_$rep MPG_WGRAD_TARGET=256 || exit 1
  run wt1024_$rep MPG_WGRAD_TARGET=1024 || exit 1
  run wt2048_$rep MPG_WGRAD_TARGET=2048 || exit 1
done
exit $rc
