#!/bin/bash
mkdir -p gpurun_out/spmm; OUT=$PWD/gpurun_out/spmm; : > $OUT/times2.txt
python tools/bench_lgcn.py >> $OUT/times2.txt 2>> $OUT/err2.txt
cat $OUT/times2.txt; tail -3 $OUT/err2.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "lgcn or lightgcn or propag or spmm or LightGCN or embed_size" 2>&1 | tail -5
