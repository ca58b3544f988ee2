#!/usr/bin/env python3
"""Per-launch HIP-event intervals of one full-catalogue evaluation (kernel-development helper)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator

wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
cfg = synth.WORKLOADS[wl]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
d = cfg["d"]
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
users, mask_lists, gt_lists = synth.eval_problem(cfg, seed=777)
ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
uid = torch.from_numpy(users).to(dev)
ev.use_graph = False          # per-launch events need the launches themselves, not the graph replay
for rep in range(3):
    ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, cfg["c"])
torch.cuda.synchronize()
ops.timing_begin()
ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, cfg["c"])
for name, ms in ops.timing_end():
    print("%-16s %8.1f us" % (name, ms * 1e3))
