#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06_f; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest.txt 2>&1; rc=$?
tail -8 $O/pytest.txt; cp gpurun_out/parity_bars.txt $O/parity_bars.txt 2>/dev/null
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.log || { tail -5 $O/bench.log; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_f/bench.json"))
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
for k,v in d["secondary"].items(): print(k, round(v["value"]), round(v["ms_per_step"],4), v.get("roofline",{}).get("frac"))
PY
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()"
