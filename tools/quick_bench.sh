#!/bin/bash
# quick look at the training step: python bench.py variants -> one line each (kernel-development helper)
mkdir -p gpurun_out/qb
for v in "" "--presorted" "--no-defer" "--workload ml10m" "--train normalbce" "--workload addressa"; do
  n=$(echo "x$v" | tr -d ' -')
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-eval $v > gpurun_out/qb/$n.json 2> gpurun_out/qb/$n.err
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open("gpurun_out/qb/%s.json"%n)); print(n, round(d["ms_per_step"]*1e3,2), {k:round(v["avg_us"],2) for k,v in d["kernels"].items()})
except Exception as e: print(n,"ERR",e)
PY
done
