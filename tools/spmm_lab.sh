#!/bin/bash
# SpMM lab on the GPU box: tools/spmm_bench.hip (which #includes the product's spmm_kernels.hip) built with the given -D
# variants, each run on the Yelp2018-shape graph; one JSON line per variant -> gpurun_out/spmm_lab.txt
#   bash tools/spmm_lab.sh "name1:-DFLAG1 -DFLAG2" "name2:" ...        (environment variables pass through)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/spmm_lab.txt
mkdir -p $ROOT/gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  bin=$ROOT/tools/spmm_bench_$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-result $flags $ROOT/tools/spmm_bench.hip -o $bin 2> $ROOT/gpurun_out/spmm_lab_$name.err || { echo "$name: build failed" >> $OUT; tail -5 $ROOT/gpurun_out/spmm_lab_$name.err; continue; }
  echo "== $name ($flags) ${LAB_ENV}" >> $OUT
  env $LAB_ENV timeout 120 $bin ${LAB_ARGS:-31668 38048 43.3 64 2 30} >> $OUT 2>> $ROOT/gpurun_out/spmm_lab_$name.err
  tail -3 $ROOT/gpurun_out/spmm_lab_$name.err >> $OUT
done
cat $OUT
