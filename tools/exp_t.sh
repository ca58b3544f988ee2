timeout 1200 python -m pytest tests/test_gpu_product.py -x -q 2>&1 | tail -4
python bench.py --workload config4 --steps 256 --warmup 70 --regions 2 > gpurun_out/final/bench_config4_1gpu.json 2> gpurun_out/final/bench_config4_1gpu.err
python -c "
import json; d=json.load(open('gpurun_out/final/bench_config4_1gpu.json'))
print(d['ms_per_step'], d['value'], d['eval_users_per_s'], d['eval_ms_per_pass'], d['eval_info'], d['eval_fast_stats'], d['last_losses'])"
