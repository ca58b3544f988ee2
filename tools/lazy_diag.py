"""diagnostic: where do lazy and dense Adam first differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_lazy_adam import _models, _batch
from macr_amd import ops
d = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_users, n_items, B = 5003, 1201, 1500
(dense, lazy), rs = _models(n_users, n_items, d, [1, 5], ops.LOSS_RUBIBCEBOTH)
allu = torch.arange(n_users, dtype=torch.int32, device="cuda")
alli = torch.arange(n_items, dtype=torch.int32, device="cuda")
for step in range(1, 13):
    u, i, j = _batch(rs, n_users, n_items, B)
    pre = [t.clone() for t in (dense.P, dense.mP, dense.vP)]
    pre_lazy = lazy.backend.lazy_rows(lazy, "P", allu)
    l0, l1 = dense.step(u, i, j), lazy.step(u, i, j)
    cur = lazy.backend.lazy_rows(lazy, "P", allu)
    curq = lazy.backend.lazy_rows(lazy, "Q", alli)
    badp = (cur != dense.P).any(dim=1)
    badq = (curq != dense.Q).any(dim=1)
    tu = torch.zeros(n_users, dtype=torch.bool, device="cuda"); tu[u.long()] = True
    ti = torch.zeros(n_items, dtype=torch.bool, device="cuda"); ti[i.long()] = True; ti[j.long()] = True
    print("step", step, "loss equal", bool(torch.equal(l0, l1)), "bad P rows", int(badp.sum()), "of which touched now", int((badp & tu).sum()),
          "bad Q rows", int(badq.sum()), "touched now", int((badq & ti).sum()))
    if int(badp.sum()):
        r = int(badp.nonzero()[0])
        e = int((cur[r] != dense.P[r]).nonzero()[0])
        b1, b2 = torch.tensor(0.9, dtype=torch.float32), torch.tensor(0.999, dtype=torch.float32)
        print("  elem", e, "dense pre th/m/v %r %r %r" % (pre[0][r, e].item(), pre[1][r, e].item(), pre[2][r, e].item()),
              "post %r %r %r" % (dense.P[r, e].item(), dense.mP[r, e].item(), dense.vP[r, e].item()),
              "m*b1 %r v*b2 %r" % ((pre[1][r, e].cpu() * b1).item(), (pre[2][r, e].cpu() * b2).item()),
              "lazy pre-step virtual th %r, now %r; stored th/m/v %r %r %r" % (pre_lazy[r, e].item(), cur[r, e].item(), lazy._P[r, e].item(), lazy._mP[r, e].item(), lazy._vP[r, e].item()),
              "gP row abs max dense %r lazy %r flags %r %r" % (dense.gP[r].abs().max().item(), lazy.gP[r].abs().max().item(), int(dense.tP[r]), int(lazy.tP[r])))
        print("  row", r, "stamp", int(lazy.stP[r]), "touched", bool(tu[r]), "diff", (cur[r] - dense.P[r]).abs().max().item(),
              "m equal", bool(torch.equal(lazy._mP[r], dense.mP[r])), "v equal", bool(torch.equal(lazy._vP[r], dense.vP[r])))
    if int(badq.sum()):
        r = int(badq.nonzero()[0])
        print("  Q row", r, "stamp", int(lazy.stQ[r]), "touched", bool(ti[r]), "diff", (curq[r] - dense.Q[r]).abs().max().item(), "refs", int((i == r).sum() + (j == r).sum()))
