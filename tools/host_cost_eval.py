#!/usr/bin/env python3
"""Host-side cost of one Evaluator.test_mf call at the Gowalla shape (graph replay): wall time per call against the
device time of the replay (events around it), and a cProfile of 200 calls."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator

dev = torch.device("cuda")
cfg = synth.WORKLOADS["gowalla"]
gen = torch.Generator(device=dev).manual_seed(1)
d = cfg["d"]
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
users, mask_lists, gt_lists = synth.eval_problem(cfg, seed=777)
ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
uid = torch.from_numpy(users).to(dev)
run = lambda: ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 40.0)
for _ in range(5):
    run()
torch.cuda.synchronize()
n = 200
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(n):
    run()
e1.record(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e6
print("filter %s: wall %.1f us per evaluation (events %.1f us), seeded %s" % (ev.filter, wall, e0.elapsed_time(e1) / n * 1e3, ev._last_seeded))
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    run()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(18)
