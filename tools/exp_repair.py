#!/usr/bin/env python3
"""Cost of a repair round at the Gowalla shape under both candidate filters: thresholds seeded from a ranking of a
DIFFERENT item table (stale everywhere), and from the same table with one query block's seeds scrambled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from macr_amd import ops, synth

dev = torch.device("cuda")
cfg = synth.WORKLOADS["gowalla"]; d = cfg["d"]
gen = torch.Generator(device=dev).manual_seed(1)
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
Q2 = synth.xavier_table(cfg["n_items"], d, gen, dev)
users, mask_lists, gt = synth.eval_problem(cfg, seed=777)
uid = torch.from_numpy(users).to(dev); U = len(users)
mask = ops.CSR.from_lists(mask_lists, dev)
stats = torch.zeros(2, dtype=torch.int32, device=dev)
for filt in ("f32", "bf16"):
    ops.set_eval_filter(filt)
    seeds = torch.full((U, ops.SEED_WIDTH), -1, dtype=torch.int32, device=dev)
    ops.score_topk(ops.SCORE_NORMAL, P, uid, Q, 20, mask=mask, seed_out=seeds)
    good = seeds.clone()
    for name, sd, tab in (("fresh seeds", good.clone(), Q), ("one block scrambled", None, Q), ("all stale", good.clone(), Q2)):
        if sd is None:
            sd = good.clone(); sd[:256] = torch.randint(0, cfg["n_items"], (256, ops.SEED_WIDTH), device=dev, dtype=torch.int32)
        for rep in range(2):
            s_in = sd.clone()
            ops.timing_begin()
            ops.score_topk(ops.SCORE_NORMAL, P, uid, tab, 20, mask=mask, seed=s_in, seed_out=s_in, stats=stats)
            marks = {}
            for n, ms in ops.timing_end():
                marks[n] = marks.get(n, 0.0) + ms * 1e3
        print(filt, name, "relisted", stats.tolist(), "total %.0f us" % sum(marks.values()), {k: round(v) for k, v in marks.items()})
ops.set_eval_filter("env")
