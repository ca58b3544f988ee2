#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of a command, then PMC passes in runs of their own
# (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots, MI355X_MICROARCH.md "rocprofv3 PMC slots";
# counters are never combined with sys/hip/hsa tracing).
#   bash tools/profile.sh <tag> <command...>     -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq}
# The default bench:  bash tools/profile.sh bench python $PWD/bench.py --steps 200 --warmup 20 --no-cpu-baseline
set -x
tag=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$tag
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o bench -- "$@" > $OUT/stdout_trace.txt 2> $OUT/trace.err
timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o bench -- "$@" > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o bench -- "$@" > /dev/null 2> $OUT/pmc_write.err
timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_sq -o bench -- "$@" > /dev/null 2> $OUT/pmc_sq.err
# optional pass: the VALU instruction classes (names differ between rocprofv3 builds: a refused counter only loses this pass)
timeout 600 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/pmc_valu -o bench -- "$@" > /dev/null 2> $OUT/pmc_valu.err
cd $ROOT
# per-kernel averages instead of the raw per-dispatch tables (the merge back is capped at 64 MiB)
python tools/summarize_profile.py condense $OUT
find $OUT -type f -size +6M -delete; du -sh $OUT; tail -2 $OUT/*.err
