import ctypes, os, sys, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from macr_amd import ops, synth, _lib
from macr_amd.evaluator import Evaluator
cfg = synth.WORKLOADS["gowalla"]; dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(1)
P = synth.xavier_table(cfg["n_users"], 64, gen, dev); Q = synth.xavier_table(cfg["n_items"], 64, gen, dev)
if len(sys.argv) > 1 and sys.argv[1] == "trained":     # spread scores like a trained model
    P = P * 30; Q = Q * 30 + torch.randn(cfg["n_items"], 1, device=dev, generator=gen) * 0.3
w = synth.xavier_table(64, 1, gen, dev).reshape(-1); wu = synth.xavier_table(64, 1, gen, dev).reshape(-1)
users, mask, gt = synth.eval_problem(cfg, 777)
ev = Evaluator(mask, gt, cfg["n_items"], dev); uid = torch.from_numpy(users).to(dev)
for kind in (ops.SCORE_NORMAL, ops.SCORE_RUBI_BOTH):
    ev.rank(kind, P, uid, Q, 20, w, wu, 40.0); torch.cuda.synchronize()
    L = _lib.lib()
    if hasattr(L, "macr_dbg_counters"):
        out = (ctypes.c_ulonglong * 8)(); L.macr_dbg_counters(out)
        ev.rank(kind, P, uid, Q, 20, w, wu, 40.0); torch.cuda.synchronize(); L.macr_dbg_counters(out)
        print("kind", kind, "tile-rounds(per wave)", out[3], "with candidates", out[0], "appended keys", out[1], "compactions", out[2],
              "-> per user: appends %.1f compactions %.1f" % (out[1] / len(users), out[2] / len(users)))
    ops.timing_begin(); ev.rank(kind, P, uid, Q, 20, w, wu, 40.0); print("kind", kind, [(n, round(ms * 1e3)) for n, ms in ops.timing_end()])
