bash tools/ab_step.sh gpurun_out/r05_ab_lazy_mf2.txt "MACR_LAZY_ADAM_MF=1" "MACR_LAZY_ADAM_MF=2" "MACR_LAZY_ADAM_MF=3" "MACR_LAZY_ADAM_MF=4"
