# A/B on one box: the host learns a replayed evaluation's results by watching the pinned result words (MACR_EVAL_POLL=1) or by
# synchronising the stream (0)
for i in 1 2 3; do
for f in 0 1; do
MACR_BENCH_DEBUG=1 MACR_EVAL_POLL=$f python bench.py --no-cpu-baseline > gpurun_out/bench_poll${f}_$i.json 2> gpurun_out/bench_poll${f}_$i.err
grep "eval times" gpurun_out/bench_poll${f}_$i.err | sed "s/^/poll=$f /"
done
done
python - <<'P'
import json
for i in (1, 2, 3):
    for f in "01":
        d = json.load(open("gpurun_out/bench_poll%s_%d.json" % (f, i)))
        e = d["eval"]
        print("poll=%s run %d: bf16 eval mean %.4f median %.4f min %.4f ms, device %.1f us, host gap %.1f us, unseeded %.4f; f32 mean %.4f median %.4f; users/s %.4g; metrics %s" % (
            f, i, e["bf16"]["ms_per_eval"], e["bf16"]["ms_per_eval_median"], e["bf16"]["ms_per_eval_min"],
            e["bf16"]["device_us_per_eval"], e["bf16"]["host_gap_us"], e["bf16"]["ms_unseeded"], e["f32"]["ms_per_eval"], e["f32"]["ms_per_eval_median"], d["eval_users_per_s"], d["eval_metrics"]))
P
