# A/B on one box: the first backward layer of the LightGCN step row-sparse (MACR_LGCN_BWD1_DENSE=0) or dense (1; default at d = 64)
for i in 1 2 3; do for f in 0 1; do
MACR_LGCN_BWD1_DENSE=$f python bench.py --workload yelp2018 --no-cpu-baseline --no-eval > gpurun_out/lg_${f}_$i.json 2>/dev/null
python - <<P
import json
d = json.load(open("gpurun_out/lg_${f}_$i.json"))
print("bwd1_dense=$f run $i: step %.2f us, %.4g interactions/s" % (d["ms_per_step"] * 1e3, d["value"]), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()}, "losses", d.get("last_losses"))
P
done; done
