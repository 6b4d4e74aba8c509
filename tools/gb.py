import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],3), "fwd", round(k["mpg_mab_fwd"]["avg_ms"]*1e3,1), "bwd", round(k["mpg_mab_bwd"]["avg_ms"]*1e3,1))
