bash tools/final_bench.sh > gpurun_out/final_bench.log 2>&1
bash tools/round_profiles.sh > gpurun_out/round_profiles.log 2>&1
python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
