#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06_w128; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.log || { tail -3 $O/bench_$name.log; return 1; }
  python -c "import json; d=json.load(open('$O/bench_$name.json')); k=d['kernels']; print('$name: %.0f jets/s %.4f ms  wgrad %.1f us  reduce %.1f us' % (d['value'], d['ms_per_step'], k['mpg_gemm_wgrad_group']['avg_ms']*1e3, k['mpg_splitk_reduce_group_dw']['avg_ms']*1e3))"
}
for rep in 1 2; do
  run old_$rep MPG_LIBDIR=$R/mpgan_amd/lib_alt MPG_LIB_STALE_OK=1 MPG_WGRAD_TARGET_BIG=2048 || exit 1
  for t in 64 128 256 512 1024; do run big${t}_$rep MPG_WGRAD_TARGET_BIG=$t || exit 1; done
done
