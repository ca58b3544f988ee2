#!/usr/bin/env python3
"""Kernel-development helper: the dense LightGCN propagation alone (2 layers) at a workload's shapes, per-kernel
HIP-event times.  The plan / access-policy experiment knobs are read from the environment by the library
(MACR_SPMM_PLAN, MACR_SPMM_POL, MACR_SPMM_HOT_KB).  usage: spmm_lab.py [workload] [reps] [check]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from macr_amd import ops, synth

name = sys.argv[1] if len(sys.argv) > 1 else "yelp2018"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = synth.WORKLOADS[name]
dev = torch.device("cuda"); n_u, n_i, d, L = cfg["n_users"], cfg["n_items"], 64, 2
zipf = os.environ.get("LAB_ZIPF", "1") != "0"          # LAB_ZIPF=0: uniform item popularity (a graph without hub rows)
cache = "/tmp/spmm_lab_%s_%d.npz" % (name, zipf)
if os.path.exists(cache):
    A = sp.load_npz(cache)
else:
    lists = synth.interaction_lists(n_u, n_i, cfg["n_train"] / n_u, seed=9, zipf=zipf)
    rows = np.repeat(np.arange(n_u), [len(l) for l in lists]); cols = np.concatenate(lists)
    R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_u, n_i))
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5).astype(np.float32)
    dinv[np.isinf(dinv)] = 0
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32); A.sort_indices()
    sp.save_npz(cache, A, compressed=False)
N, nnz = A.shape[0], A.nnz
gen = torch.Generator(device=dev).manual_seed(1)
T = synth.xavier_table(N, d, gen, dev)
adj = ops.CSR.from_scipy(A, dev)
out = torch.empty_like(T)
for _ in range(5):
    ops.lgcn_propagate(adj, T, L, out=out)
torch.cuda.synchronize()
ops.timing_begin()
for _ in range(reps):
    ops.lgcn_propagate(adj, T, L, out=out)
agg = {}
for nm, ms in ops.timing_end(4096):
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += ms
res = {"plan": os.environ.get("MACR_SPMM_PLAN", "0"), "pol": os.environ.get("MACR_SPMM_POL", "0"),
       "hot_kb": os.environ.get("MACR_SPMM_HOT_KB", ""), "N": N, "nnz": nnz,
       "kernels_us": {k: round(1e3 * v[1] / v[0], 2) for k, v in agg.items()}}
if len(sys.argv) > 3:
    Td = T.double().cpu().numpy()
    E1 = A @ Td; E2 = A @ E1
    ref = (Td + E1 + E2) / 3
    res["max_err"] = float(np.abs(out.double().cpu().numpy() - ref).max())
print(json.dumps(res))
