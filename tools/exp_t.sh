# config4 on one GPU, steady state: regions of 256 steps after 70 warm-up steps (every period has run its full lag)
for K in 1 8 16 32 63; do
  echo "== K=$K"
  S=256; [ $K = 1 ] && S=40
  timeout 900 python bench.py --workload config4 --steps $S --warmup 70 --regions 2 --no-cpu-baseline --c4-lazy $K --no-eval 2>gpurun_out/c4_K$K.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'], 4), 'value', round(d['value']), 'K', d['lazy_adam_period'])
print({k: round(v['event_us'], 1) for k, v in d['kernels'].items()})
"
done
