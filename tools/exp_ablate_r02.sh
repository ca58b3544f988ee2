mkdir -p gpurun_out/exp1
for v in base noatomic nopart; do
  if [ $v = base ]; then unset MACR_HIP_LIB; else export MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_hip_$v.so; fi
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-eval > gpurun_out/exp1/$v.json 2> gpurun_out/exp1/$v.err
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-eval --unsorted > gpurun_out/exp1/${v}_unsorted.json 2>> gpurun_out/exp1/$v.err
done
unset MACR_HIP_LIB
python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-eval --no-defer > gpurun_out/exp1/base_nodefer.json 2>> gpurun_out/exp1/base.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/exp1/*.json')):
    try:
        d=json.load(open(f))
        print(f, round(d['ms_per_step']*1e3,2), {k:round(v['avg_us'],2) for k,v in d['kernels'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
