#!/bin/bash
# Ablations of the fp16 listing pass k_score_stream_h (eval_kernels.hip -DMACR_ABL_H_NOEPI / _NOMFMA / _NOBARRIER): builds
# -DMACR_DEV_FAST variants of the library (d = 64, kinds NORMAL / RUBI_BOTH: ~40 s each) under macr_amd/csrc/_abl and prints the
# listing pass's time under each, sampled and seeded.   tools/abl_stream_h.sh NONE NOEPI NOEPI+NOMFMA NOEPI+NOBARRIER
# (profiles/r06_eval_f16_filter.txt holds the round-6 numbers)
C=/root/repo/macr_amd/csrc; L=$C/_abl; mkdir -p $L
for v in "$@"; do
  defs=""; [ "$v" != "NONE" ] && for a in ${v//+/ }; do if [ "$a" = "TPS2" ]; then defs="$defs -DMACR_H_TPS=2"; else defs="$defs -DMACR_ABL_H_$a"; fi; done
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -DMACR_DEV_FAST $defs -c $C/eval_kernels.hip -o $L/eval_h_$v.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmacr_hip_h_$v.so $C/_obj/capi_common.o $C/_obj/train_kernels.o $C/_obj/spmm_kernels.o $C/_obj/sample_kernels.o $L/eval_h_$v.o ) &
done; wait
args=()
for v in "$@"; do args+=("MACR_EVAL_FILTER=f16 MACR_HIP_LIB=$L/libmacr_hip_h_$v.so"); done
bash tools/ab_eval.sh gpurun_out/ab_eval_h_abl.txt "${args[@]}" > /dev/null 2>&1
grep -o "libmacr_hip_h_[A-Za-z0-9+]*.so *[0-9.]* M users/s\|FAILED.*\|seeded (listing [0-9.]*\|score_stream_b [0-9.]*" gpurun_out/ab_eval_h_abl.txt | paste - - -
