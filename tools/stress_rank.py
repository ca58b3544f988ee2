import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from macr_amd import ops
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for seed in range(60):
    rs = np.random.RandomState(9000 + seed)
    U = int(rs.choice([700, 2100, 5000])); N = int(rs.choice([12000, 33000, 60000])); d = int(rs.choice([32, 64, 128, 256]))
    K = int(rs.choice([10, 20, 32])); kind = int(rs.choice([0, 1])); W = int(rs.choice([1, 2, 4, 8]))
    P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32); Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    trend = rs.choice([0.0, 1.0, -1.0])
    Q[:, 0] += trend * np.linspace(1.5, -1.5, N).astype(np.float32); P[:, 0] = np.abs(P[:, 0])
    w = (rs.standard_normal(d) * 0.3).astype(np.float32); wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    lens = rs.poisson(30, U); lens[0] = min(N - 3, 20000)
    mask = [sorted(rs.choice(N, size=int(l), replace=False).tolist()) for l in lens]
    mptr, midx = oracle.csr_from_lists(mask)
    sig_i = ops.branch_sigmoid(dev(Q), dev(w)); sig_u = ops.branch_sigmoid(dev(P), dev(wu))
    sel = np.arange(0, U, 37)
    sub = oracle.csr_from_lists([mask[q] for q in sel])
    wv, wi, wc = oracle.score_topk(kind, P[sel], Q, K, sig_u.cpu().numpy()[sel], sig_i.cpu().numpy(), 40.0, sub)
    m = ops.CSR(dev(mptr), dev(midx))
    vs, is_ = [], []
    for r in range(W):
        lo, hi = N * r // W, N * (r + 1) // W
        v, i = ops.score_topk(kind, dev(P), None, dev(Q[lo:hi]), K, sig_u, sig_i[lo:hi].contiguous(), 40.0, m, lo)
        mv, mi, _ = ops.topk_merge(v, i); vs.append(mv); is_.append(mi)
    gv, gi, gc = ops.topk_merge(torch.stack(vs), torch.stack(is_))
    ok = np.array_equal(gi.cpu().numpy()[sel], wi) and np.array_equal(gv.cpu().numpy()[sel].view(np.uint32), wv.view(np.uint32))
    bad += not ok
    print(seed, U, N, d, K, kind, W, trend, "OK" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
