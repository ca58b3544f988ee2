P='import json,sys
o=json.loads(sys.stdin.readline()); print(sys.argv[1], "eval Musers/s", round(o["eval_users_per_s"]/1e6,2), {k:round(v) for k,v in o["roofline_eval"]["kernels_us"].items()})'
python -m pytest tests/test_gpu_ops.py tests/test_gpu_product.py -m gpu -x -q 2>&1 | tail -3
python bench.py --workload addressa --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" addressa
python bench.py --workload addressa --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" addressa
