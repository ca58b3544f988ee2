#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run: true per-kernel durations (kernel-development helper)
# usage: bash tools/prof_quick.sh <tag> [bench args...]
tag=$1; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/pq_$tag; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-eval "$@" > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls=%6s avg_us=%8.2f total_ms=%8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
find $OUT -type f -size +4M -delete
