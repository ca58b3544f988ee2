# A/B of the evaluator's launch folds on one box: MACR_EVAL_FOLD = 0 (separate launches) | p (prologue only) | m (metrics+means only) | 1 (both)
for i in 1 2; do
for f in 0 p m 1; do
MACR_BENCH_DEBUG=1 MACR_EVAL_FOLD=$f python bench.py > gpurun_out/bench_fold${f}_$i.json 2> gpurun_out/bench_fold${f}_$i.err
grep "eval times" gpurun_out/bench_fold${f}_$i.err | sed "s/^/fold=$f /"
done
done
python - <<'P'
import json
for i in (1, 2):
    for f in "0pm1":
        d = json.load(open("gpurun_out/bench_fold%s_%d.json" % (f, i)))
        e = d["eval"]
        print("fold=%s run %d: step %.2f us; bf16 eval mean %.4f median %.4f min %.4f ms, device %.1f us, unseeded %.4f; f32 %.4f" % (
            f, i, d["ms_per_step"] * 1e3, e["bf16"]["ms_per_eval"], e["bf16"]["ms_per_eval_median"], e["bf16"]["ms_per_eval_min"],
            e["bf16"]["device_us_per_eval"], e["bf16"]["ms_unseeded"], e["f32"]["ms_per_eval"]))
P
