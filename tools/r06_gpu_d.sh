#!/bin/bash
# round 6: the full evidence set (tools/final_profiles.sh) + the three-segment overhead on one GPU + soak
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/final_profiles.sh || { echo "final_profiles failed"; exit 1; }
cd $R
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/final/bench_one_graph_$rep.json 2>/dev/null || exit 1
  MPG_SPLIT_GRAPHS=1 timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/final/bench_three_segments_$rep.json 2>/dev/null || exit 1
done
python - <<'PY'
import json
for n in ("one_graph_1","three_segments_1","one_graph_2","three_segments_2"):
    d=json.load(open(f"gpurun_out/final/bench_{n}.json")); print(n, round(d["value"]), "jets/s", round(d["ms_per_step"],4), "ms", d["config"]["graphs_per_step"], "graphs")
PY
(timeout -k 10 400 python tools/soak.py) > gpurun_out/final/soak.txt 2>&1; tail -3 gpurun_out/final/soak.txt
