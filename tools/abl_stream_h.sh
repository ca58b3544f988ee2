#!/bin/bash
# ablations of k_score_stream_h (eval_kernels.hip -DMACR_ABL_H_*; -DMACR_DEV_FAST builds under macr_amd/csrc/_abl): listing-pass time of each
# arguments: variant names, optionally "VARIANT:ENV=VALUE"
L=/root/repo/macr_amd/csrc/_abl
args=()
for v in "$@"; do n=${v%%:*}; e=""; [ "$n" != "$v" ] && e="${v#*:}"; args+=("MACR_EVAL_FILTER=f16 MACR_HIP_LIB=$L/libmacr_hip_h_$n.so $e"); done
bash tools/ab_eval.sh gpurun_out/ab_eval_h_abl.txt "${args[@]}" > /dev/null 2>&1
grep -o "libmacr_hip_h_[A-Za-z]*.so [A-Z_=0-9]*\|FAILED.*\|seeded (listing [0-9.]*\|score_stream_b [0-9.]*" gpurun_out/ab_eval_h_abl.txt | paste - - -
