P='import json,sys
o=json.loads(sys.stdin.readline()); print(sys.argv[1], round(o["value"]/1e6,2),"M/s", round(o["ms_per_step"]*1e3,1),"us/step", "eval Musers/s", round(o["eval_users_per_s"]/1e6,2), {k:round(v) for k,v in o["roofline_eval"]["kernels_us"].items()})'
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P" gowalla
python bench.py --workload ml10m --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P" ml10m
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
