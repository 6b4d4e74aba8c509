#!/bin/bash
# same-box A/B of two library builds: mpgan_amd/lib (default) against mpgan_amd/lib_alt (MPG_LIBDIR); usage: r06_gpu_ab.sh <label> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; L=${1:-alt}; O=gpurun_out/r06_ab_$L; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.log || { tail -3 $O/bench_$name.log; return 1; }
  python -c "import json; d=json.load(open('$O/bench_$name.json')); k=d['kernels']; print('$name: %.0f jets/s %.4f ms  fwd_fn %.1f us  bwd_fn %.1f us  chain %.1f us' % (d['value'], d['ms_per_step'], k['mpg_edge_fwd_fn']['avg_ms']*1e3, k['mpg_edge_bwd_fn']['avg_ms']*1e3, k['mpg_chain']['avg_ms']*1e3))"
}
for rep in $(seq 1 ${2:-3}); do
  run base_$rep X=1 || exit 1
  run ${L}_$rep MPG_LIBDIR=$R/mpgan_amd/lib_alt MPG_LIB_STALE_OK=1 || exit 1
done
