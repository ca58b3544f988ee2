P='import json,sys
o=json.loads(sys.stdin.readline()); print(sys.argv[1], round(o["value"]/1e6,2),"M/s", round(o["ms_per_step"]*1e3,1),"us/step", {k:round(v["avg_us"],1) for k,v in o["kernels"].items()})'
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
for v in base 8T; do
  if [ $v = base ]; then unset MACR_HIP_LIB; else export MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_hip_$v.so; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-eval 2>/dev/null | python -c "$P" "gowalla_$v"
  python bench.py --workload ml10m --steps 100 --warmup 10 --no-cpu-baseline --no-eval 2>/dev/null | python -c "$P" "ml10m_$v"
done
