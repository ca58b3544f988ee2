#!/bin/bash
# scratch: eval kernels of the bench under a filter
f=${1:-bf16}
MACR_EVAL_FILTER=$f python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f abl=$MACR_ABL', round(r['eval_users_per_s']/1e6,2), 'M users/s', r['eval_ms_per_pass'], r['eval_ms_unseeded'])
k=r['roofline_eval']['kernels_us']; print({a:round(b,1) for a,b in k.items() if 'stream' in a or 'select' in a})
k=r['roofline_eval']['seeded']['kernels_us']; print({a:round(b,1) for a,b in k.items() if 'stream' in a or 'select' in a})"
