#!/bin/bash
# training-step timing of library variants (kernel-development helper): bash tools/exp_step.sh <variant|base>...
mkdir -p gpurun_out/es
for v in "$@"; do
  if [ $v = base ]; then unset MACR_HIP_LIB; else export MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_hip_$v.so; fi
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-eval --no-e2e > gpurun_out/es/$v.json 2> gpurun_out/es/$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.load(open("gpurun_out/es/%s.json"%v)); print(v, round(d["ms_per_step"]*1e3,2), round(d["timed_regions"]["min_ms_per_step"]*1e3,2), {k:round(x["avg_us"],2) for k,x in d["kernels"].items()})
except Exception as e: print(v,"ERR",e)
PY
done
