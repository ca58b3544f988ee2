#!/bin/bash
# rocprofv3 kernel-trace stats of any command: bash tools/prof_cmd.sh <tag> <command...>   (kernel-development helper)
tag=$1; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/pc_$tag; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- "$@" > $OUT/stdout.txt 2> $OUT/err.txt
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-70s calls=%6s avg_us=%9.2f min_us=%9.2f max_us=%9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
find $OUT -type f -size +4M -delete
