#!/usr/bin/env python3
"""c sweep of the tuners (tune.py:545-578) on Gowalla / ML-10M shapes: 12 values of c through the shared-listing-pass
kernel (groups of four) vs twelve single-c evaluations (one graph replay each).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator

dev = torch.device("cuda")
out = []
for wl in (sys.argv[1:] or ["gowalla", "ml10m"]):
    cfg = synth.WORKLOADS[wl]
    d = cfg["d"]
    gen = torch.Generator(device=dev).manual_seed(3)
    P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
    w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    users, mask, gt = synth.eval_problem(cfg, seed=777)
    uid = torch.from_numpy(users).to(dev)
    ev = Evaluator(mask, gt, cfg["n_items"], dev)
    cs = [float(c) for c in np.linspace(20.0, 42.0, 12)]
    def single():
        return [ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, c) for c in cs]
    def sweep():
        return ev.test_mf_sweep(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, cs)
    res = {}
    for name, fn in (("single", single), ("sweep", sweep)):
        a = fn(); fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 5 / len(cs) * 1e3
        res[name + "_hit"] = [float(x["hit_ratio"][0]) for x in r]
    assert res["single_hit"] == res["sweep_hit"]
    out.append({"workload": wl, "values_of_c": len(cs), "ms_per_c_single": res["single"], "ms_per_c_sweep": res["sweep"],
                "speedup": res["single"] / res["sweep"]})
print(json.dumps(out))
