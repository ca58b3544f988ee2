#!/usr/bin/env python3
"""Steady-state evaluations as the product runs them (graph replays), for `rocprofv3 --kernel-trace`:
tools/eval_timeline.sh prints the last replay launch by launch (start offset, duration, gap to the previous kernel)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator

wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flavour = sys.argv[3] if len(sys.argv) > 3 else "mf"
cfg = synth.WORKLOADS[wl]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
d = cfg["d"]
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
users, mask_lists, gt_lists = synth.eval_problem(cfg, seed=777)
ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
uid = torch.from_numpy(users).to(dev)
test = ev.test_mf if flavour == "mf" else ev.test_lgcn
for rep in range(reps):
    test(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, cfg["c"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(10):
    test(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, cfg["c"])
torch.cuda.synchronize()
print("wall per evaluation: %.1f us (%s, seeds %s, filter %s)" % ((time.perf_counter() - t0) * 1e5, wl,
      os.environ.get("MACR_EVAL_SEEDS", "1"), ev.filter))
