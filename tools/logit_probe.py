#!/usr/bin/env python3
"""Where the (B,B) kernel's logits sit in the synthetic bench model as it trains: per 64-column wave tile, is every column
inside the window of the 4-transcendental form (train_kernels.hip, `fast`)?  Prints, after 0 / 20 / 220 / 2000 / 6000 steps,
the quantiles of p, n (the columns' logits; x = p*a, y = n*b with a, b in (0,1)) and the share of tiles inside
  current window   p >= -20, -20 <= n <= 3
  wider window     p >= -20, -60 <= n <= 3
  python tools/logit_probe.py [--workload gowalla|ml10m]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from macr_amd import ops, synth
ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="gowalla"); a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]; B, d = cfg["batch"], cfg["d"]; dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(12345)
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
state = ops.MFState(P, Q, w, wu, ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B), B)
batches = synth.train_batches(256, cfg["n_users"], cfg["n_items"], B, gen, dev)
loss = torch.zeros((256, 3), device=dev)
done = 0
for upto in (0, 220, 400, 800, 2000, 6000):
    for s in range(done, upto):
        k = s % 256
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k, 0], batches[k, 1], batches[k, 2], loss[k], defer=True)
    ops.timing_begin()
    for s in range(upto, upto + 6):
        k = s % 256
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k, 0], batches[k, 1], batches[k, 2], loss[k], defer=True)
    marks = ops.timing_end(max_n=64)
    bx = [ms for name, ms in marks if name == "bxb+adam"]
    state.flush(); done = upto + 6
    u, i, j = (batches[done % 256, r].long() for r in range(3))
    p = (state.P[u] * state.Q[i]).sum(1); n = (state.P[u] * state.Q[j]).sum(1)
    q = lambda t: [round(float(x), 2) for x in torch.quantile(t, torch.tensor([0, .01, .5, .99, 1.0], device=dev))]
    def tiles(ok): return float(ok.view(-1, 64).all(1).float().mean())
    sg = torch.sigmoid
    ssu = sg(state.P[u] @ state.wu); a_ = sg(state.Q[i] @ state.w) * ssu; b_ = sg(state.Q[j] @ state.w) * ssu
    R = 256 if B >= 4096 else 64
    pad = (-B) % R
    amax = torch.nn.functional.pad(a_, (0, pad)).view(-1, R).max(1).values; bmax = torch.nn.functional.pad(b_, (0, pad)).view(-1, R).max(1).values
    okrc = ((p[None, :] * amax[:, None] >= -20) & (n[None, :] * bmax[:, None] <= 3) & (n[None, :] * bmax[:, None] >= -60))
    pc = (-B) % 64
    share = float(torch.nn.functional.pad(okrc, (0, pc), value=True).view(okrc.shape[0], -1, 64).all(2).float().mean())
    print("   a %s b %s  bmax per row block %s  wave tiles in the row-bounded window %.3f" % (q(a_), q(b_), q(bmax), share))
    cur = (p >= -20) & (n >= -20) & (n <= 3); wide = (p >= -20) & (n >= -60) & (n <= 3)
    print("bxb+adam %.1f us  steps %5d  p %s  n %s  cols in window %.3f / %.3f  tiles %.3f / %.3f  (n>3: %.4f, n<-20: %.4f, p<-20: %.4f)" % (
        1e3 * float(np.median(bx)) if bx else -1, done, q(p), q(n), float(cur.float().mean()), float(wide.float().mean()), tiles(cur), tiles(wide),
        float((n > 3).float().mean()), float((n < -20).float().mean()), float((p < -20).float().mean())), flush=True)
