#!/bin/bash
# evaluator leg of bench.py against library variants (kernel-development helper): bash tools/exp_eval.sh <variant|base>...
mkdir -p gpurun_out/ee
for v in "$@"; do
  for wl in gowalla ml10m; do
    if [ $v = base ]; then unset MACR_HIP_LIB; else export MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_hip_$v.so; fi
    python bench.py --steps 20 --warmup 5 --regions 1 --no-e2e --no-cpu-baseline --workload $wl > gpurun_out/ee/${v}_$wl.json 2> gpurun_out/ee/${v}_$wl.err
    python - $v $wl <<'PY'
import json,sys
v,wl=sys.argv[1:]
try:
    d=json.load(open("gpurun_out/ee/%s_%s.json"%(v,wl))); r=d["roofline_eval"]
    print(v, wl, "eval_ms", round(d["eval_ms_per_pass"],4), "frac", round(r["frac"],3), "stream_frac", round(r["stream"]["frac"],3), {k:round(x,1) for k,x in r["kernels_us"].items()}, d["eval_metrics"])
except Exception as e: print(v, wl, "ERR", e)
PY
  done
done
