#!/bin/bash
# Round-end evidence: rocprofv3 kernel stats of the default bench, HBM traffic counters and SQ counters of the edge
# kernels, the MFMA issue/power micro-benchmarks.  Run on the GPU box from the repo root; results in gpurun_out/final
# (tools/make_profiles.py <tag> then writes the committed summaries under profiles/).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
# build BEFORE any profiled process: a profiled python must never start hipcc (the profiler's preload initialises the
# GPU in every child; hipcc then execs clang -- a GPU-initialised exec, forbidden on this pool)
python3 $R/__graft_entry__.py || exit 1
python3 -c "import sys; sys.path.insert(0, '$R'); from mpgan_amd import _lib; print(_lib.source_digest())" > $R/gpurun_out/final/source_digest.txt || exit 1
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/final/bench_stats
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/final/bench_under_rocprof.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $out 40 > $R/gpurun_out/final/bench_stats_summary.txt
out=$R/gpurun_out/final/gapt_stats
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b -- python3 $R/bench.py --model gapt --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/final/gapt_under_rocprof.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $out 40 > $R/gpurun_out/final/gapt_stats_summary.txt
echo "stats done"
KERNELS=("edge_fwd1_fn_kernel<0" "edge_fwd1_fn_kernel<2" "edge_bwd1_fn_kernel<0, true" "edge_bwd1_fn_kernel<0, false" "edge_bwd1_fn_kernel<2, true" "edge_bwd1_fn_kernel<2, false" "edge_dw12_kernel<0" "edge_dw12_kernel<2" chain2_kernel gemm_group_kernel)
for c in FETCH_SIZE WRITE_SIZE; do
  o=$R/gpurun_out/final/pmc_$c
  rm -rf $o
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o -o p -- python3 $R/tools/kbwd.py > $o.log 2>&1 || exit 1
  for k in "${KERNELS[@]}"; do
    echo "== $c $k"; python3 $R/tools/pmc_summary.py $o "$k"
  done
done > $R/gpurun_out/final/pmc_summary.txt 2>&1
echo "pmc done"
o=$R/gpurun_out/final/pmc_sq
rm -rf $o
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $o -o p -- python3 $R/tools/kbwd.py > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
for k in "edge_fwd1_fn_kernel<0" "edge_fwd1_fn_kernel<2" "edge_bwd1_fn_kernel<0, true" "edge_bwd1_fn_kernel<0, false" "edge_bwd1_fn_kernel<2, true" "edge_bwd1_fn_kernel<2, false" "edge_dw12_kernel<0" "edge_dw12_kernel<2"; do
  echo "== $k"; python3 $R/tools/pmc_summary.py $o "$k"
done > $R/gpurun_out/final/pmc_sq_summary.txt 2>&1
# the one-launch attention blocks of GAPT: SQ counters and HBM traffic from a short bench run (no timing legs)
for c in SQ FETCH_SIZE WRITE_SIZE; do
  o=$R/gpurun_out/final/pmc_gapt_$c
  rm -rf $o
  if [ $c = SQ ]; then ctrs="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"; else ctrs=$c; fi
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $o -o p -- python3 $R/bench.py --model gapt --steps 3 --warmup 2 --no-graphs --no-roofline --no-cpu-baseline > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
  for k in "mab_chain_fwd2_kernel<4" "mab_chain_fwd2_kernel<8" "mab_fwd2_kernel<true, 4" "mab_fwd2_kernel<true, 8" "mab_bwd2_kernel<false" "mab_bwd2_kernel<true" "mab_bwd_kernel<2, false" "mab_bwd_kernel<2, true" bridge_fwd_kernel bridge_bwd_kernel; do
    echo "== $c $k"; python3 $R/tools/pmc_summary.py $o "$k"
  done
done > $R/gpurun_out/final/pmc_gapt_summary.txt 2>&1
echo "sq done"
# HBM traffic of the SECONDARY workloads' kernels (bench.py's `secondary` rooflines): GAPT B = 512 comes from the passes above;
# MPGAN N = 150, B = 16 (sender chunks: the plain eight-wave edge kernels) from a short bench run of its own
for c in FETCH_SIZE WRITE_SIZE; do
  o=$R/gpurun_out/final/pmc_n150_$c
  rm -rf $o
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o -o p -- python3 $R/bench.py --particles 150 --batch 16 --steps 3 --warmup 2 --no-graphs --no-roofline --no-cpu-baseline --no-secondary > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
  for k in edge_fwd edge_bwd edge_dw1 chain disc_head; do
    echo "== mpgan_n150_b16 $c $k"; python3 $R/tools/pmc_summary.py $o "$k"
  done
  for k in mab_bwd mab_chain_fwd mab_fwd bridge; do
    echo "== gapt_n30_b512 $c $k"; python3 $R/tools/pmc_summary.py $R/gpurun_out/final/pmc_gapt_$c "$k"
  done
done > $R/gpurun_out/final/pmc_secondary_summary.txt 2>&1
echo "secondary pmc done"
# timelines of one replayed iteration of each workload (tools/timeline.py), from kernel traces of short runs of their own
cd /tmp
for w in "mpgan_n30:" "mpgan_n150:--particles 150 --batch 16" "gapt_n30:--model gapt" "gapt_n150:--model gapt --particles 150 --batch 64"; do
  name=${w%%:*}; args=${w#*:}
  o=$R/gpurun_out/final/trace_$name
  rm -rf $o
  rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 $R/bench.py $args --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-secondary > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
  python3 $R/tools/timeline.py $o > $R/gpurun_out/final/timeline_$name.txt 2>&1
done
echo "timelines done"
cd $R
(timeout -k 10 600 python3 tools/determinism.py) > gpurun_out/final/determinism.txt 2>&1 || { tail -5 gpurun_out/final/determinism.txt; exit 1; }
echo "determinism done"
(timeout -k 10 120 tools/ubench/mfma_model; timeout -k 10 120 tools/ubench/mfma_model2; timeout -k 10 120 tools/ubench/mfma_power) > gpurun_out/final/ubench.txt 2>&1
echo "ubench done"
