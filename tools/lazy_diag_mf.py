"""diagnostic: where do the lazy and the dense deferred MF sequences first differ?"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_lazy_adam import _mf_states, _MF_NAMES
from macr_amd import ops, _lib

def snapshot(s):
    c = ops.MFState(s.P.clone(), s.Q.clone(), s.w.clone(), s.wu.clone(), s.hyper, s.batch_cap, lazy_period=s.lazy_period)
    for n in ("mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu", "gP", "gQ", "tP", "tQ", "adam_pow"):
        setattr(c, n, getattr(s, n).clone())
    c.ws = s.ws.clone()
    c.pending_B, c.pending_kind = s.pending_B, s.pending_kind
    if s._seq_lazy is not None:
        c._lazy_bufs = tuple(t.clone() for t in s._lazy_bufs)
        st, sp, sq = c._lazy_bufs
        c._seq_lazy = _lib.LazyAdam(ops._ptr(st), ops._ptr(sp), ops._ptr(sq), s._seq_lazy.period)
    c.flush()
    return c

d = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kind = ops.LOSS_RUBIBCEBOTH
n_users, n_items, B = 2000, 1500, 128
periods = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,1,2").split(",")]
states, rs = _mf_states(n_users, n_items, d, B, periods)
print("periods", periods)
for step in range(2):
    u = rs.choice(n_users, B, replace=False).astype(np.int32)
    ij = rs.choice(n_items, 2 * B, replace=False).astype(np.int32)
    b = [torch.from_numpy(a).cuda() for a in (u, ij[:B], ij[B:])]
    losses = [s.step(kind, *b, defer=True).clone() for s in states]
    a_ws, c_ws = states[0].ws.view(torch.float32), states[2].ws.view(torch.float32)
    neq = (a_ws != c_ws) & ~(torch.isnan(a_ws) & torch.isnan(c_ws))
    if int(neq.sum()):
        idx = neq.nonzero().flatten()
        print("   ws regions (x64 floats):", sorted(set((idx // 64).tolist())))
        print("   ws differs at float offsets", idx[:8].tolist(), "count", int(neq.sum()), "of", a_ws.numel(), "vals", a_ws[idx[:3]].tolist(), c_ws[idx[:3]].tolist())
    for nm in ("gP", "gQ"):
        x, y = getattr(states[0], nm), getattr(states[2], nm)
        if not torch.equal(x, y):
            print("   %s differs rows" % nm, (x != y).any(dim=1).nonzero().flatten()[:6].tolist(), float((x - y).abs().max()), float(x.abs().max()))
    snaps = [snapshot(s) for s in states]
    msg = []
    for name in _MF_NAMES:
        a, b2, c = (getattr(x, name) for x in snaps)
        if not torch.equal(a, b2): msg.append("dense/dense %s" % name)
        if not torch.equal(a, c):
            bad = (a != c)
            rows = bad.any(dim=1).nonzero().flatten().tolist() if a.dim() == 2 else bad.nonzero().flatten().tolist()
            msg.append("dense/lazy %s rows %s" % (name, rows[:6]))
            if name in ("P", "Q") and rows:
                r = rows[0]
                inb = (u == r).any() if name == "P" else (ij == r).any()
                stamps = states[2]._lazy_bufs[1 if name == "P" else 2]
                msg.append("  row %d in batch %s stamp(before flush) %d maxdiff %g" % (r, bool(inb), int(stamps[r]), float((a[r] - c[r]).abs().max())))
    print("step", step + 1, "loss dd", bool(torch.equal(losses[0], losses[1])), "dl", bool(torch.equal(losses[0], losses[2])), "; ".join(msg))
