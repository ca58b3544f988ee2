#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench, then two PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -x
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --eval-reps 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --eval-reps 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --eval-reps 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
cd $ROOT
find $OUT -type f -size +8M -delete; find $OUT -type f | head -50; tail -5 $OUT/*.err
du -sh $OUT
