#!/bin/bash
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/spmm; mkdir -p $OUT; ROOT=$PWD; : > $OUT/nb.txt
for v in nb1 nb2 base; do
  if [ $v = base ]; then unset MACR_HIP_LIB; else export MACR_HIP_LIB=$ROOT/macr_amd/csrc/_abl/libmacr_hip_$v.so; fi
  echo "== $v" >> $OUT/nb.txt
  python tools/spmm_lab.py yelp2018 30 >> $OUT/nb.txt 2>/dev/null
  python tools/bench_lgcn.py 2>/dev/null | cut -c1-330 >> $OUT/nb.txt
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/f$v -o x -- python $ROOT/tools/spmm_lab.py yelp2018 10 > /dev/null 2>&1)
  python - $v <<'PY' >> gpurun_out/spmm/nb.txt
import csv,glob,collections,sys
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/spmm/f%s/**/*counter_collection.csv'%sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmm_row' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v),1) for k,v in agg.items()})
PY
  rm -rf $OUT/f$v
done
cat $OUT/nb.txt
