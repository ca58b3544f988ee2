#!/bin/bash
# how stale may the evaluator's threshold seeds be?  eval time vs training steps between two evaluations
mkdir -p gpurun_out/sd
for wl in gowalla ml10m; do
  for n in 0 20 200 2000 20000; do
    python bench.py --steps 20 --warmup 5 --regions 1 --no-e2e --no-cpu-baseline --workload $wl --eval-train-steps $n --eval-reps 3 > gpurun_out/sd/${wl}_$n.json 2> gpurun_out/sd/${wl}_$n.err
    python - $wl $n <<'PY'
import json,sys
wl,n=sys.argv[1:]
try:
    d=json.load(open("gpurun_out/sd/%s_%s.json"%(wl,n)))
    print(wl, "train steps between evals", n, "eval_ms", round(d["eval_ms_per_pass"],4), "unseeded_ms", round(d["eval_ms_unseeded"],4), d["eval_metrics"])
except Exception as e: print(wl, n, "ERR", e)
PY
  done
done
