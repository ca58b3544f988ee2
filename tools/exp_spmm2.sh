#!/bin/bash
mkdir -p gpurun_out/spmm; OUT=$PWD/gpurun_out/spmm; : > $OUT/times2.txt
for v in nogather ng_nopiece ng_nostore ng_all nopiece; do
echo "== $v" >> $OUT/times2.txt
MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_hip_$v.so python tools/spmm_lab.py yelp2018 30 >> $OUT/times2.txt 2>> $OUT/err2.txt
done
cat $OUT/times2.txt; tail -3 $OUT/err2.txt
