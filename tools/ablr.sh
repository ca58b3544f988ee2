P='import json,sys
o=json.loads(sys.stdin.readline()); print(sys.argv[1], round(o["value"]/1e6,2),"M/s", round(o["ms_per_step"]*1e3,1),"us/step", {k:round(v["avg_us"],1) for k,v in o["kernels"].items()})'
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q -k "mf_ or lgcn or full" 2>&1 | tail -3
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-eval 2>/dev/null | python -c "$P" gowalla
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-eval 2>/dev/null | python -c "$P" gowalla
python bench.py --workload ml10m --steps 100 --warmup 10 --no-cpu-baseline --no-eval 2>/dev/null | python -c "$P" ml10m
