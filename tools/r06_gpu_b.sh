#!/bin/bash
# round 6, second GPU call: the -m gpu suite in both product forms of the forward, same-box A/B of the step, precision tables
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06_b; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.txt 2>&1; rc=$?
tail -15 $O/pytest.txt; cp gpurun_out/parity_bars.txt $O/parity_bars.txt 2>/dev/null
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
echo "pytest (three-term) rc=$rc"
MPG_FWD_TWO_TERM=1 timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_two_term.txt 2>&1; rc2=$?
tail -30 $O/pytest_two_term.txt; cp gpurun_out/parity_bars.txt $O/parity_bars_two_term.txt 2>/dev/null
if [ $rc2 -ge 124 ]; then echo "pytest killed rc=$rc2"; exit $rc2; fi
echo "pytest (two-term) rc=$rc2"
for rep in 1 2; do
  for tt in 0 1; do
    MPG_FWD_TWO_TERM=$tt timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_tt${tt}_$rep.json 2> $O/bench_tt${tt}_$rep.log || exit 1
    python - <<PY
import json; d=json.load(open("$O/bench_tt${tt}_$rep.json")); print("two_term=$tt rep $rep:", round(d["value"]), "jets/s", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), d["kernels"]["mpg_edge_fwd_fn"])
PY
  done
done
for tt in 0 1; do
  MPG_FWD_TWO_TERM=$tt timeout -k 10 600 python tests/probe_precision.py > $O/probe_tt$tt.txt 2>&1 || { tail -5 $O/probe_tt$tt.txt; exit 1; }
done
tail -40 $O/probe_tt1.txt
exit 0
