P='import json,sys
o=json.loads(sys.stdin.readline()); print(sys.argv[1], round(o["value"]/1e6,2),"M/s", round(o["ms_per_step"]*1e3,1),"us/step", {k:round(v["avg_us"],1) for k,v in o["kernels"].items()}, "eval Musers/s", round(o["eval_users_per_s"]/1e6,2), {k:round(v) for k,v in o["roofline_eval"]["kernels_us"].items()})'
for v in base noatomic nopart noadmit nomask nomfma nomfma_noadmit; do
  if [ $v = base ]; then unset MACR_HIP_LIB; else export MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_hip_$v.so; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P" $v
done
