#!/usr/bin/env python3
"""Turn gpurun_out/final/* (tools/final_profiles.sh, run on the GPU box) into the committed evidence under profiles/."""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_f"
shutil.copy(os.path.join(F, "bench_stats", "b_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
log = open(os.path.join(F, "bench_under_rocprof.log")).read()
line = [l for l in log.splitlines() if l.startswith('{"metric"')][-1]
bench = json.loads(line)
summ = open(os.path.join(F, "bench_stats_summary.txt")).read()
summ = re.sub(r"/\S*/gpurun_out/", "gpurun_out/", summ)
# launch-weighted rocprof average of the roofline kernel, to set beside bench.py's own event timing
rk = bench.get("roofline", {}).get("kernel", "")
tot_ns = calls = 0
for l in open(os.path.join(P, f"{tag}_bench_kernel_stats.csv")).read().splitlines()[1:]:
    m = re.match(r'"(.*)",(\d+),(\d+),', l)
    if m and rk and rk in m.group(1):
        calls += int(m.group(2)); tot_ns += int(m.group(3))
agree = ""
if calls:
    k = bench["kernels"]["mpg_" + rk.replace("_kernel", "")]
    agree = (f"# Agreement of the two clocks for {rk}: bench.py HIP events avg {k['avg_ms'] * 1e3:.1f} us over "
             f"{k['launches_per_step']} launches/step; rocprof launch-weighted avg {tot_ns / calls / 1e3:.1f} us over {calls} launches.\n")
open(os.path.join(P, f"{tag}_bench_kernel_stats.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary\n"
    "# (MI355X, 1 GPU, MPGAN N=30, B=256, gluon-like multiplicity, D dropout 0.5 -- ONE workload: no secondary legs; warm-up, capture warm-up, timed and\n"
    "# the 4 eager roofline iterations all land in the trace).  Full table: " + f"{tag}_bench_kernel_stats.csv.\n"
    "# The bench line printed by this very run:\n# " + line + "\n#\n" + agree + summ)

# ---- GAPT kernel stats of the same round
gl = os.path.join(F, "gapt_under_rocprof.log")
if os.path.isfile(gl):
    gline = [l for l in open(gl).read().splitlines() if l.startswith('{"metric"')][-1]
    gs = re.sub(r"/\S*/gpurun_out/", "gpurun_out/", open(os.path.join(F, "gapt_stats_summary.txt")).read())
    open(os.path.join(P, f"{tag}_bench_gapt_kernel_stats.txt"), "w").write(
        "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --model gapt --steps 20 --warmup 5 --no-cpu-baseline\n"
        "# (MI355X, 1 GPU, GAPT N=30, B=512, D dropout 0.5).  The bench line printed by this very run:\n# " + gline + "\n#\n" + gs)

# ---- HBM traffic
pm = open(os.path.join(F, "pmc_summary.txt")).read()
vals = {}
cur = None
for l in pm.splitlines():
    m = re.match(r"== (\w+) (.*)", l)
    if m:
        cur = (m.group(1), m.group(2)); continue
    m = re.match(r"(\w+)\s+n=\s*(\d+) avg=\s*([\d.]+)", l)
    if m and cur:
        vals[cur] = float(m.group(3))
def kb(c, k): return vals.get((c, k), 0.0)
def tot(k): return (2 * kb("FETCH_SIZE", k) + kb("WRITE_SIZE", k)) * 1024
rows = [("edge_fwd1_fn_kernel<0", "forward, no dropout"), ("edge_fwd1_fn_kernel<2", "forward, p = 1/2"),
        ("edge_bwd1_fn_kernel<0, true", "backward + staging"), ("edge_bwd1_fn_kernel<0, false", "backward, data path only"),
        ("edge_bwd1_fn_kernel<2, true", "backward + staging, p = 1/2"), ("edge_bwd1_fn_kernel<2, false", "backward, data path, p = 1/2"),
        ("edge_dw12_kernel<0", "weight gradients"), ("edge_dw12_kernel<2", "weight gradients, p = 1/2"),
        ("chain2_kernel", "chained node layers (all uses)"), ("gemm_group_kernel", "grouped dense weight gradients")]
txt = ("# HBM traffic of the fused kernels: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one counter per pass, as\n"
       "# MI355X_MICROARCH.md prescribes); workload tools/kbwd.py = MPLayer forward+backward at B=256, N=30 (one launch = 256 jets,\n"
       "# 230,400 edges).  Counters are KiB per dispatch, averaged over the kernel's dispatches.  gfx950 correction: FETCH_SIZE\n"
       "# reports half of the bytes of wide coalesced reads, so HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE.\n#\n"
       "#   kernel                                             2 x FETCH + WRITE = MB per launch\n")
for k, what in rows:
    txt += f"#   {k:34s} {what:32s} 2 x {kb('FETCH_SIZE', k) / 1024:7.1f} + {kb('WRITE_SIZE', k) / 1024:7.1f} = {tot(k) / 2**20:7.1f}\n"
# achieved HBM rate = bytes per launch / rocprof average duration of the B=256 launches in the bench trace
dur = {}
for l in open(os.path.join(P, f"{tag}_bench_kernel_stats.csv")).read().splitlines()[1:]:
    m = re.match(r'"(.*)",(\d+),(\d+),([\d.]+),', l)
    if m:
        dur[m.group(1)] = float(m.group(4)) * 1e-9
def rate(sub, key):
    d = [v for k, v in dur.items() if sub in k]
    return tot(key) / d[0] / 1e12 if d else float("nan")
txt += ("#\n# Achieved HBM rate at B=256 (bytes above / rocprof average duration in " + f"{tag}_bench_kernel_stats.csv" + "; HBM3E peak 8 TB/s):\n"
        f"#   edge_fwd1_fn_kernel<0, true>   {rate('edge_fwd1_fn_kernel<0, true', 'edge_fwd1_fn_kernel<0'):5.2f} TB/s  (compute bound: see the SQ counters)\n"
        f"#   edge_bwd1_fn_kernel<0, true>   {rate('edge_bwd1_fn_kernel<0, true', 'edge_bwd1_fn_kernel<0, true'):5.2f} TB/s\n"
        f"#   edge_dw12_kernel<0>        {rate('edge_dw12_kernel<0', 'edge_dw12_kernel<0'):5.2f} TB/s\n")
txt += ("#\n# forward: a|c in, agg + sign words out (algorithmic 17.7 MB) + 10 KiB of fp16 E2 fragments per unmasked (jet, sender) block,\n"
        "# parked for the backward (which reads them for the LeakyReLU gate instead of recomputing the layer) and for mpg_edge_dw;\n"
        "# backward + staging: 10 KiB of fp16 dZ2 fragments per block; mpg_edge_dw reads both back and writes 256 per-workgroup\n"
        "# partial sums.\n#\n" + pm)
open(os.path.join(P, f"{tag}_pmc_hbm_traffic.txt"), "w").write(txt)
w, wo = tot("edge_bwd1_fn_kernel<2, true"), tot("edge_bwd1_fn_kernel<2, false")
w0 = tot("edge_bwd1_fn_kernel<0, true")
traffic = {
    "note": f"bytes per launch from profiles/{tag}_pmc_hbm_traffic.txt (2 x FETCH_SIZE + WRITE_SIZE), B=256, N=30, the launches with their epilogue chains (node network / dx). edge_bwd_fn_kernel: "
            "launch-weighted mean over the 6 launches of one default bench step (2 at 2B = 512 jets with staging [D, p = 1/2], "
            "2 at B = 256 data path only [D in the G step], 2 at B = 256 with staging [G, no dropout]); edge_fwd_fn_kernel: mean over "
            "its 8 launches (2 at B without by-products, 2 at 512 jets and 4 at B with sign words + parked E2)",
    # forward: 2 launches at B without by-products for a backward (the generator in the D step: a|c in, agg out = 13.3 MB
    # algorithmic, not in the PMC workload), 2 at 2B and 4 at B with sign words and parked E2
    "edge_fwd_fn_kernel": {"bytes_per_launch": int((2 * 13.3e6 + 2 * 2 * tot("edge_fwd1_fn_kernel<2") + 4 * tot("edge_fwd1_fn_kernel<0")) / 8)},
    "edge_bwd_fn_kernel": {"bytes_per_launch": int((2 * 2 * w + 2 * wo + 2 * w0) / 6)},
    "edge_dw_kernel": {"bytes_per_launch": int(tot("edge_dw12_kernel<2"))},
}
# ---- the secondary workloads' kernels (bench.py: secondary.*.roofline.traffic), launch-weighted over every dispatch whose name
#      contains the entry point's stem (mab_bwd = mab_bwd_kernel + mab_bwd2_kernel, edge_fwd = the eight-wave edge_fwd1_kernel ...)
sp = os.path.join(F, "pmc_secondary_summary.txt")
if os.path.isfile(sp):
    sv, cur = {}, None
    for l in open(sp).read().splitlines():
        m = re.match(r"== (\w+) (\w+) (.*)", l)
        if m:
            cur = m.groups(); continue
        m = re.match(r"(\w+)\s+n=\s*(\d+) avg=\s*([\d.]+)", l)
        if m and cur:
            sv[cur] = float(m.group(3))
    sec = {}
    for wl, stems in (("gapt_n30_b512", ("mab_bwd", "mab_chain_fwd", "mab_fwd", "bridge")), ("mpgan_n150_b16", ("edge_fwd", "edge_bwd", "edge_dw1", "chain", "disc_head"))):
        for st in stems:
            f, w_ = sv.get((wl, "FETCH_SIZE", st)), sv.get((wl, "WRITE_SIZE", st))
            if f is not None and w_ is not None:
                kn = "edge_dw_kernel" if st.startswith("edge_dw") else (st if st.endswith("_kernel") else st + "_kernel")
                sec.setdefault(wl, {})[kn] = {"bytes_per_launch": int((2 * f + w_) * 1024)}
    traffic["secondary"] = sec
    traffic["note"] += ("; secondary: per launch, launch-weighted over every dispatch of the entry point's kernels in a short bench.py run "
                        f"of that workload (gpurun_out/final/pmc_secondary_summary.txt -> profiles/{tag}_pmc_hbm_traffic_secondary.txt)")
    shutil.copy(sp, os.path.join(P, f"{tag}_pmc_hbm_traffic_secondary.txt"))
# what the counters belong to: the digest of the kernel sources they were collected on (tools/final_profiles.sh writes it on the
# box) -- bench.py reports roofline.traffic only while the tree it runs on still has that digest
dg = os.path.join(F, "source_digest.txt")
traffic["source"] = {"profile": f"profiles/{tag}_pmc_hbm_traffic.txt",
                     "source_digest": open(dg).read().strip() if os.path.isfile(dg) else None}
json.dump(traffic, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
# ---- SQ counters (tools/final_profiles.sh: pmc_sq_summary.txt)
sq_path = os.path.join(F, "pmc_sq_summary.txt")
if os.path.isfile(sq_path):
    sq = open(sq_path).read()
    cur, tab = None, {}
    for l in sq.splitlines():
        m = re.match(r"== (.*)", l)
        if m:
            cur = m.group(1); tab[cur] = {}; continue
        m = re.match(r"(\w+)\s+n=\s*(\d+) avg=\s*([\d.]+)", l)
        if m and cur:
            tab[cur][m.group(1)] = float(m.group(3))
    txt = ("# SQ counters of the fused edge kernels (rocprofv3 --kernel-trace --pmc ..., workload tools/kbwd.py, B=256, N=30),\n"
           "# per dispatch.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over\n"
           "# waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (= 32 x number of 32x32x16 MFMAs) summed over SIMDs.  Two waves per SIMD in the\n"
           "# eight-wave kernels (three in edge_dw12_kernel), so  MFMA share of wave time = MFMA_BUSY / (4 x WAVE_CYCLES / waves per SIMD);  by DEVICE time (last column):\n"
           "# MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).\n#\n"
           "#   kernel                               MFMA busy / wave time   wave parked (WAIT_ANY)   issue stalled (WAIT_INST_ANY)   VALU active   VALU instr per MFMA   MFMA busy / device time\n")
    for k, v in tab.items():
        if not v.get("SQ_WAVE_CYCLES"):
            continue
        wps = 3.0 if "edge_dw12" in k else 2.0   # (eight waves per workgroup in the edge kernels: two per SIMD; twelve in mpg_edge_dw: three)
        wc = v["SQ_WAVE_CYCLES"]
        nm = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 32.0
        txt += (f"#   {k:38s} {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * wc / wps):8.2f} {v.get('SQ_WAIT_ANY', 0) / wc:22.2f} "
                f"{v.get('SQ_WAIT_INST_ANY', 0) / wc:24.2f} {v.get('SQ_ACTIVE_INST_VALU', 0) / wc:22.2f} "
                f"{(v.get('SQ_INSTS_VALU', 0) / nm if nm else 0):16.1f} "
                f"{(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (v['GRBM_GUI_ACTIVE'] / 8 * 1024) if v.get('GRBM_GUI_ACTIVE') else 0):22.2f}\n")
    open(os.path.join(P, f"{tag}_pmc_sq_counters.txt"), "w").write(txt + "#\n" + sq)

gp = os.path.join(F, "pmc_gapt_summary.txt")
if os.path.isfile(gp):
    open(os.path.join(P, f"{tag}_pmc_gapt_mab_counters.txt"), "w").write(
        "# One-launch attention blocks (csrc/mab.hip) under rocprofv3 --kernel-trace --pmc, one pass per counter set, workload\n"
        "# bench.py --model gapt --steps 3 --warmup 2 --no-graphs (B = 512: launches of 512 and 1,024 jets), per dispatch.\n"
        "# SQ_* units as in the edge-kernel table (quad-cycles summed over waves; MFMA_BUSY in cycles summed over SIMDs);\n"
        "# HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB).  One wave per jet: MFMA busy / (4 x WAVE_CYCLES) is the share of a\n"
        "# wave's lifetime the matrix pipe works -- these kernels are a latency chain of VALU work between small MFMA groups.\n#\n"
        + open(gp).read())

ub = open(os.path.join(F, "ubench.txt")).read()
open(os.path.join(P, f"{tag}_ubench_mfma.txt"), "w").write(
    "# tools/ubench/mfma_model, mfma_model2, mfma_power on MI355X (the machine model the fused kernels are scheduled against).\n"
    "# mfma_model : one wave per SIMD; a 32x32x16 f16 MFMA is 32 clk; up to ~6 independent VALU ops issued behind it are\n"
    "#              (almost) free, beyond that each costs ~4.6 clk; LDS b128 latency 61 clk.\n"
    "# mfma_model2: same with the accumulator in AGPRs; v_accvgpr_read is free up to 3 per MFMA, 6 per MFMA cost +20 clk.\n"
    "# mfma_power : dependent MFMA chains on all 1024 SIMDs: with RANDOM operands the clock drops to ~1.4 GHz (22.7 ns per\n"
    "#              MFMA = 1.48 PFLOP/s dense f16 for the whole chip) against 14.9 ns = 2.26 PFLOP/s with constant operands.\n#\n" + ub)
print("profiles written:", sorted(os.listdir(P)))

# timelines of one replayed iteration per workload, and the determinism tool's output
for name, out in (("mpgan_n30", "iteration_timeline"), ("mpgan_n150", "iteration_timeline_n150"), ("gapt_n30", "iteration_timeline_gapt"),
                  ("gapt_n150", "iteration_timeline_gapt_n150")):
    src = os.path.join(F, f"timeline_{name}.txt")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{out}.txt"))
src = os.path.join(F, "determinism.txt")
if os.path.exists(src):
    shutil.copy(src, os.path.join(P, f"{tag}_determinism.txt"))
