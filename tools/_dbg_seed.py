import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from macr_amd import ops
from oracle import oracle
from test_gpu_ops import random_mask, dev
rs = np.random.RandomState(31)
U, N, d, K, S = 900, 6000, 64, 20, ops.SEED_WIDTH
P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32)
Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
Q2 = (Q + rs.standard_normal((N, d)).astype(np.float32) * 0.05).astype(np.float32)
w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
mask = random_mask(rs, U, N, 30, heavy=(5,))
mcsr = ops.CSR.from_lists(mask, "cuda")
sig_i = ops.branch_sigmoid(dev(Q2), dev(w)); sig_u = ops.branch_sigmoid(dev(P), dev(wu))
so = ops.branch_sigmoid(dev(Q), dev(w))
prev = torch.full((U, S), -7, dtype=torch.int32, device="cuda")
v0, i0 = ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q), K, sig_u, so, 30.0, mcsr, seed_out=prev)
worst = prev.clone(); worst[:, K - 2:] = worst[:, K - 2:K - 1]
print(worst[0].tolist(), worst[1].tolist())
stats = torch.zeros(2, dtype=torch.int32, device="cuda")
out = worst.clone()
v, ix = ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q2), K, sig_u, sig_i, 30.0, mcsr, seed=out, seed_out=out, stats=stats)
print(stats.tolist())
