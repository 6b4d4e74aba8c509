#!/bin/bash
# round 6, first GPU call: forward go/no-go harness, the whole -m gpu suite (parity evidence), the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06_a; mkdir -p $O
tools/ubench/r06_fwd_lt.sh > $O/fwd_lt.txt 2>&1 || { echo "fwd harness failed"; tail -5 $O/fwd_lt.txt; exit 1; }
echo "harness done"; tail -6 $O/fwd_lt.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.txt 2>&1; rc=$?
tail -25 $O/pytest.txt; cp gpurun_out/parity_bars.txt $O/ 2>/dev/null
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
echo "pytest rc=$rc"
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.log; rc2=$?
tail -3 $O/bench.log; head -c 1500 $O/bench.json
exit $(( rc + rc2 ))
