#!/bin/bash
# PMC passes over the dense SpMM layer (tools/spmm_lab.py): where do the wave cycles go?
mkdir -p gpurun_out/spmm; OUT=$PWD/gpurun_out/spmm; ROOT=$PWD
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|Counter_Name)|SQ_|TA_|TCP_|TD_" | head -400 > $OUT/counters.txt
wc -l $OUT/counters.txt
run() { # tag counters...
  tag=$1; shift
  (cd /tmp && rocprofv3 --output-format csv --pmc "$@" --kernel-trace -d $OUT/pmc_$tag -o x -- python $ROOT/tools/spmm_lab.py yelp2018 10 > /dev/null 2>> $OUT/err_pmc.txt)
  python - $tag <<'PY'
import csv,glob,sys,collections
tag=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/spmm/pmc_%s/**/*counter_collection.csv'%tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmm_row' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(tag, {k: round(sum(v)/len(v),1) for k,v in agg.items()})
PY
  rm -rf $OUT/pmc_$tag
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA >> $OUT/pmc2.txt
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS >> $OUT/pmc2.txt
run sq3 SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG >> $OUT/pmc2.txt
run ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum >> $OUT/pmc2.txt
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum >> $OUT/pmc2.txt
run grbm GRBM_GUI_ACTIVE GRBM_COUNT >> $OUT/pmc2.txt
cat $OUT/pmc2.txt; tail -5 $OUT/err_pmc.txt
