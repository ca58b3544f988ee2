import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from macr_amd import ops
import oracle
rs = np.random.RandomState(0)
U, K, N = 15424, 20, 40981
rank = torch.from_numpy(np.stack([rs.permutation(N)[:K] for _ in range(U)]).astype(np.int32)).cuda()
gt_lists = [sorted(rs.choice(N, size=rs.randint(1, 30), replace=False).tolist()) for _ in range(U)]
gptr, gidx = oracle.csr_from_lists(gt_lists)
gt = ops.CSR(torch.from_numpy(gptr).cuda(), torch.from_numpy(gidx).cuda())
cnt = torch.full((U,), K, dtype=torch.int32).cuda()
host = torch.zeros((4, 1), dtype=torch.float64).pin_memory()
devo = torch.zeros((4, 1), dtype=torch.float64).cuda()
def t(f, n=200):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("mean->dev  %.1f us" % t(lambda: ops.metrics_mf_mean(rank, gt, [20], out=devo)))
print("mean->host %.1f us" % t(lambda: ops.metrics_mf_mean(rank, gt, [20], out=host)))
print("metrics    %.1f us" % t(lambda: ops.metrics_mf(rank, cnt, gt, [20])))
m = ops.metrics_mf(rank, cnt, gt, [20])
print("colmean    %.1f us" % t(lambda: ops.colmean(m)))
