bash tools/ab_step.sh gpurun_out/r05_ab_tpart.txt "-" "MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_TPART.so" "MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_NOPART.so"
