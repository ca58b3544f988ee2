#!/bin/bash
# PMC passes over tools/listing_bench variants: effective clock, MFMA pipe utilisation, instruction mix
ROOT=$PWD; OUT=$ROOT/gpurun_out/listing_pmc; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for b in "$@"; do
  rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace -d $OUT/$b.1 -o p -- $ROOT/tools/listing_bench_$b 15424 40981 170 3 > /dev/null 2> $OUT/$b.err1
  rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace -d $OUT/$b.2 -o p -- $ROOT/tools/listing_bench_$b 15424 40981 170 3 > /dev/null 2> $OUT/$b.err2
done
cd $ROOT
python - "$@" <<'PY'
import csv, glob, sys, collections
for b in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in glob.glob("gpurun_out/listing_pmc/%s.*/**/*counter_collection.csv" % b, recursive=True):
        for r in csv.DictReader(open(p)):
            n = r["Kernel_Name"]
            if "k_list" in n or "k_score_stream_b" in n:
                key = n[:n.index("(")].replace("void ", "")[:40]
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for p in glob.glob("gpurun_out/listing_pmc/%s.1/**/*kernel_trace.csv" % b, recursive=True):
        for r in csv.DictReader(open(p)):
            n = r["Kernel_Name"]
            if "k_list" in n or "k_score_stream_b" in n:
                dur[n[:n.index("(")].replace("void ", "")[:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, c in agg.items():
        m = {x: sum(v) / len(v) for x, v in c.items()}
        d = sum(dur[k]) / max(1, len(dur[k])) / 1e3
        gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
        print("%s %-40s %.1f us (under pmc)  clock %.2f GHz  MFMA pipe busy %.0f %% of SIMD cycles; per MFMA: VALU %.1f SALU %.1f LDS %.1f; VALU-active %.0f %% of wave cycles, waiting %.0f %%, LDS wait %.0f %%" % (
            b, k, d, gui / d / 1e3 if d else 0, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui) if gui else 0,
            m.get("SQ_INSTS_VALU", 0) / max(1, m.get("SQ_INSTS_MFMA", 1)), m.get("SQ_INSTS_SALU", 0) / max(1, m.get("SQ_INSTS_MFMA", 1)),
            m.get("SQ_INSTS_LDS", 0) / max(1, m.get("SQ_INSTS_MFMA", 1)),
            100 * m.get("SQ_ACTIVE_INST_VALU", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)) , 100 * m.get("SQ_WAIT_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)),
            100 * m.get("SQ_WAIT_INST_LDS", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1))))
        print("      raw:", {x: round(v / 1e6, 2) for x, v in m.items()})
PY
find $OUT -type f -size +4M -delete
