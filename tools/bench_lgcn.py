#!/usr/bin/env python3
"""Diagnostic bench of the LightGCN path at Yelp2018 shapes (BASELINE.json configs[3]): per-kernel HIP-event
times of one training step and of the propagation alone, with the SpMM roofline (SURVEY.md 8d:
nnz*8 + (N+1)*4 + 2*N*d*4 compulsory bytes per layer)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from macr_amd import ops, synth

cfg = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "yelp2018"]
dev = torch.device("cuda"); n_u, n_i, d, B, L = cfg["n_users"], cfg["n_items"], 64, cfg["batch"], 2
lists = synth.interaction_lists(n_u, n_i, cfg["n_train"] / n_u, seed=9)
rows = np.repeat(np.arange(n_u), [len(l) for l in lists]); cols = np.concatenate(lists)
R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_u, n_i))
A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
deg = np.asarray(A.sum(1)).ravel()
with np.errstate(divide="ignore"):
    dinv = np.power(deg, -0.5).astype(np.float32)
dinv[np.isinf(dinv)] = 0
A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32); A.sort_indices()
N, nnz = A.shape[0], A.nnz
gen = torch.Generator(device=dev).manual_seed(1)
T = synth.xavier_table(N, d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
adj = ops.CSR.from_scipy(A, dev)
state = ops.LGCNState(T, n_u, n_i, w, wu, adj, L, ops.make_hyper(1e-3, 1e-5, cfg["alpha"], cfg["beta"], B), B)
batches = synth.train_batches(32, n_u, n_i, B, gen, dev)
for k in range(5):
    state.step(ops.LOSS_RUBIBCEBOTH, batches[k, 0], batches[k, 1], batches[k, 2])
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
for k in range(n):
    state.step(ops.LOSS_RUBIBCEBOTH, batches[k % 32, 0], batches[k % 32, 1], batches[k % 32, 2])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
ops.timing_begin()
for k in range(10):
    state.step(ops.LOSS_RUBIBCEBOTH, batches[k, 0], batches[k, 1], batches[k, 2])
agg = {}
for name, ms in ops.timing_end(512):
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
bytes_layer = nnz * 8 + (N + 1) * 4 + 2 * N * d * 4
dense = "spmm_stream" if "spmm_stream" in agg else "spmm_csr"
out = {"N": N, "nnz": nnz, "us_per_step": dt * 1e6, "interactions_per_s": B / dt,
       "kernels_us": {k: round(1e3 * v[1] / v[0], 1) for k, v in agg.items()},
       "launches_per_step": {k: v[0] / 10 for k, v in agg.items()},
       "spmm_algorithmic_MB": bytes_layer / 1e6,
       "spmm_GBps": bytes_layer / (1e-3 * agg[dense][1] / agg[dense][0]) / 1e9}
print(json.dumps(out))
