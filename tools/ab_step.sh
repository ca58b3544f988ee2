#!/bin/bash
# A/B of the MF step on one box: tools/ab_step.sh OUT "ENV1" "ENV2" ...  (each ENV a string of VAR=value settings; "-" = none)
# Prints ms_per_step and the per-kernel averages of every setting, three runs each, interleaved.
out=$1; shift
: > "$out"
for rep in 1 2 3; do
  for e in "$@"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    line=$(env $envs python bench.py --no-cpu-baseline --no-eval --no-e2e ${AB_ARGS} 2>/dev/null | tail -1)
    python - "$e" "$line" >> "$out" <<'PY'
import json, sys
e, line = sys.argv[1], sys.argv[2]
try:
    j = json.loads(line)
    k = j.get("kernels", {})
    print("%-28s step %.2f us  %s" % (e, 1e3 * j["ms_per_step"], "  ".join("%s %.2f" % (n, v.get("avg_us", v.get("event_us", 0))) for n, v in k.items())))
except Exception as ex:
    print(e, "FAILED", ex, line[:200])
PY
  done
done
cat "$out"
