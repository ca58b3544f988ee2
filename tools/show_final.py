"""One line per gpurun_out/final/bench_*.json (tools/final_bench.sh)."""
import glob, json
g = lambda x, k: (x.get(k) if x else None)
rd = lambda x: None if x is None else round(x, 4)
for f in sorted(glob.glob('gpurun_out/final/bench_*.json')):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    if "value" not in d:
        print(f.split('/')[-1], d); continue
    r = d.get("roofline_eval") or {}
    print(f.split('/')[-1], "value %.4g" % d["value"], "ms/step %.4g" % d["ms_per_step"], "eval users/s %.4g" % (d.get("eval_users_per_s") or 0),
          "eval ms", rd(d.get("eval_ms_per_pass")), "unseeded", rd(d.get("eval_ms_unseeded")), "roofline", rd(g(d.get("roofline"), "frac")),
          "step", rd(g(d.get("roofline_step"), "frac")), "eval frac", rd(r.get("frac")), "stream", rd(g(r.get("stream"), "frac")),
          "seeded:", {k: (rd(v) if isinstance(v, float) else v) for k, v in (r.get("seeded") or {}).items() if k in ("seeded", "query_blocks_relisted", "frac", "avg_us")})
    if d.get("eval_modes"):
        print("    ", "".join("S" if m["seeded"] else "-" for m in d["eval_modes"]), [m["query_blocks_relisted"] for m in d["eval_modes"]],
              "e2e", g(d.get("end_to_end"), "interactions_per_s"))
    if r: print("    ", {k: round(v, 1) for k, v in r["kernels_us"].items()})
