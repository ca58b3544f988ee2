"""Evaluator on one GPU's share of BASELINE configs[4] (10 M x 1 M, d=128, item-sharded over 8 GPUs): 125 000 items,
20 000 of the query users.  Prints one JSON line."""
import json, os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator
dev = torch.device("cuda", 0)
cfg = dict(n_users=200000, n_items=125000, d=128, n_train=200000*20, n_test_users=20000, test_per_user=10)
gen = torch.Generator(device=dev).manual_seed(1)
d = cfg["d"]
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
users, mask_lists, gt_lists = synth.eval_problem(cfg, seed=777)
ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
uid = torch.from_numpy(users).to(dev)
for rep in range(2):
    ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 40.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for rep in range(3):
    r = ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 40.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
fl = 2.0 * len(users) * cfg["n_items"] * d
ev.use_graph = False          # per-kernel events need the launches themselves
ops.timing_begin(); ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 40.0)
marks = {}
for n, ms in ops.timing_end():
    marks[n] = marks.get(n, 0.0) + ms * 1e3
stream = marks.get("score_stream_b", marks.get("score_stream", float("nan")))      # bf16 candidate filter (default) / fp32
print(json.dumps({"workload": "configs[4] shard: %d query users x %d items, d=%d, K=20, c=40" % (len(users), cfg["n_items"], d),
                  "eval_ms": dt * 1e3, "users_per_s": len(users) / dt, "flops": fl, "achieved_tflops": fl / dt / 1e12,
                  "frac_of_fp32_mfma_peak": fl / dt / 157.3e12, "listing_pass_us": stream,
                  "filter": ev.filter, "listing_pass_frac_of_fp32_mfma_peak": fl / (stream * 1e-6) / 157.3e12, "kernels_us": {k: round(v, 1) for k, v in marks.items()}}))
