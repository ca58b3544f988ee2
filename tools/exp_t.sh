#!/bin/bash
# The evaluator part of the default bench in a few lines: users/s of the default (bf16 candidate filter) and of the fp32
# filter, per-kernel times of a sampled and of a seeded ranking.    bash tools/exp_t.sh
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(r['eval_users_per_s']/1e6,2), 'M users/s', round(r['eval_ms_per_pass'],3), 'ms per evaluation,', round(r['eval_ms_unseeded'],3), 'unseeded;  fp32 filter:', round(r['roofline_eval']['eval_users_per_s']/1e6,2), 'M users/s')
for name, k in (('sampled', r['roofline_eval_bf16']['kernels_us']), ('seeded', r['roofline_eval_bf16']['seeded']['kernels_us'])):
    print(name, {a: round(b, 1) for a, b in k.items()})
print('modes', [(m['seeded'], m['query_blocks_relisted']) for m in r['eval_modes']])"
