#!/bin/bash
# kernel-by-kernel timeline of the last graph replay of tools/eval_timeline.py (seeded and sampled evaluations)
ROOT=$PWD; OUT=$ROOT/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
WL=${1:-gowalla}; FL=${2:-mf}
for seeds in 1 0; do
  MACR_EVAL_SEEDS=$seeds python $ROOT/tools/eval_timeline.py $WL 8 $FL > $OUT/plain_$seeds.txt 2>&1
  MACR_EVAL_SEEDS=$seeds rocprofv3 --output-format csv --kernel-trace -d $OUT/s$seeds -o tl -- python $ROOT/tools/eval_timeline.py $WL 8 $FL > $OUT/run_$seeds.txt 2> $OUT/err_$seeds.txt
done
cd $ROOT
python - <<'PY'
import csv, glob
for seeds in (1, 0):
    print(open("gpurun_out/timeline/plain_%d.txt" % seeds).read().strip().splitlines()[-1], "(without rocprof)")
    p = glob.glob("gpurun_out/timeline/s%d/**/*kernel_trace.csv" % seeds, recursive=True)
    if not p:
        print("no trace"); continue
    rows = sorted(csv.DictReader(open(p[0])), key=lambda r: int(r["Start_Timestamp"]))
    # the last evaluation = from the last k_topk_ws_init (or the two sigmoids before it) to the end
    last = max(i for i, r in enumerate(rows) if "ws_init" in r["Kernel_Name"])
    first = last
    while first > 0 and "branch_sigmoid" in rows[first - 1]["Kernel_Name"]:
        first -= 1
    ev = rows[first:]
    t0 = int(ev[0]["Start_Timestamp"]); prev_end = t0; busy = 0
    print("seeds=%d: %d launches" % (seeds, len(ev)))
    for r in ev:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ", "").replace("macr::", "")
        name = name[:name.index("(")] if "(" in name else name
        print("  %8.1f  %7.1f us  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:60]))
        busy += e - s; prev_end = e
    print("  total %.1f us, kernels %.1f us, gaps %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))
PY
find $OUT -type f -size +8M -delete
