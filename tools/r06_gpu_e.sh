#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06_e; mkdir -p $O
python3 $R/__graft_entry__.py || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/gp_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gp_stats -o b -- python3 $R/bench.py --gp 10 --loss w --steps 6 --warmup 3 --no-graphs --no-secondary --no-cpu-baseline --no-roofline > $O/gp_under_rocprof.log 2>&1 || { tail -5 $O/gp_under_rocprof.log; exit 1; }
python3 $R/tools/prof_summary.py $O/gp_stats 40 > $O/gp_stats_summary.txt
head -45 $O/gp_stats_summary.txt
