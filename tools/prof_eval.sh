#!/bin/bash
# PMC passes over one evaluation (tools/eval_marks.py): instruction mix and wait cycles per kernel.
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof_eval; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/p1 -o ev -- python $ROOT/tools/eval_marks.py > /dev/null 2> $OUT/p1.err
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/p2 -o ev -- python $ROOT/tools/eval_marks.py > /dev/null 2> $OUT/p2.err
cd $ROOT
python - <<'PY'
import csv, collections, glob
for p in sorted(glob.glob("gpurun_out/prof_eval/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "score_stream" in n or "k_select" in n:
            key = ("stream" if "stream" in n else "select")
            agg[(key, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    # last 7 dispatches = the timed evaluation
    keys = sorted(agg, key=lambda k: int(k[1]))[-6:]
    for k in keys:
        print(k, {c: round(v / 1e6, 2) for c, v in agg[k].items()})
PY
tail -n 3 $OUT/p1.err
find $OUT -type f -size +8M -delete
