#!/bin/bash
# PMC of the stream SpMM kernel (tools/spmm_bench): separate passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/spmm_bench
O=$GRAFT_REPO_ROOT/gpurun_out/spmm_pmc
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "Name:\s*[A-Za-z0-9_]*" | sed 's/Name:\s*//' | sort -u > $O/counters.txt
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- $B 31668 38048 43.3 64 2 5 > /dev/null 2>$O/$n.err; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run p3 SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run p4 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
run p5 GRBM_GUI_ACTIVE FETCH_SIZE
run p6 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
for n in ("p1","p2","p3","p4","p5","p6"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % n, recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "spmm_stream" not in r.get("Kernel_Name", ""): continue
            k = (r["Counter_Name"])
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (c, v) in sorted(agg.items()):
        print(n, k, "launches", c, "avg_per_launch", v / c)
PY
