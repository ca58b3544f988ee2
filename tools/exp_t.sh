#!/bin/bash
mkdir -p gpurun_out/mf
for v in base bxb1 bxb4; do
  if [ $v = base ]; then unset MACR_HIP_LIB; else export MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_hip_$v.so; fi
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-eval --no-e2e > gpurun_out/mf/$v.json 2> gpurun_out/mf/$v.err
  python bench.py --workload ml10m --steps 200 --warmup 40 --no-cpu-baseline --no-eval --no-e2e > gpurun_out/mf/${v}_ml10m.json 2>> gpurun_out/mf/$v.err
done
unset MACR_HIP_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/mf/*.json')):
    try:
        d=json.load(open(f))
        print(f, round(d['ms_per_step']*1e3,2), {k:round(v['avg_us'],2) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
