#!/bin/bash
# build first: a profiled python must never compile (see tools/final_profiles.sh)
python3 $GRAFT_REPO_ROOT/__graft_entry__.py || exit 1
# Per-kernel times of the default bench (one rank) under rocprofv3; run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/bprof
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > $out.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $out 40
