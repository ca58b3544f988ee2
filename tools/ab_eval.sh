#!/bin/bash
# A/B of the evaluator leg on one box: tools/ab_eval.sh OUT "ENV1" "ENV2" ...  (each ENV a string of VAR=value settings; "-" = none)
# Prints users/s and the per-kernel microseconds of one reduced-precision-filter evaluation for every setting, two runs each, interleaved.
out=$1; shift
: > "$out"
for rep in 1 2; do
  for e in "$@"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    line=$(env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-config4 ${AB_ARGS} 2>/dev/null | tail -1)
    python - "$e" "$line" >> "$out" <<'PY'
import json, sys
e, line = sys.argv[1], sys.argv[2]
try:
    j = json.loads(line)
    r = j.get("roofline_eval_f16") or j["roofline_eval_bf16"]
    k = r["kernels_us"]
    sk = (r.get("seeded") or {}).get("kernels_us", {})
    print("%-40s %.2f M users/s  eval %.1f us of kernels sampled, %.1f seeded (listing %.1f, prep+tau_seed %.1f, select %.1f)  %s" % (
        e, 1e-6 * j["eval_users_per_s"], sum(k.values()), sum(sk.values()), sk.get("score_stream_b", 0), sk.get("bf16_prep+tau_seed", 0),
        sk.get("select_b", 0), "  ".join("%s %.1f" % (n, v) for n, v in k.items())))
except Exception as ex:
    print(e, "FAILED", ex, line[:300])
PY
  done
done
cat "$out"
