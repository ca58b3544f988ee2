#!/usr/bin/env python3
"""Where the CPU checker's training step spends its time on THIS box's host cores (Gowalla shape), phase by phase, next to
the torch-CPU graph port; OMP_NUM_THREADS sweeps show what the thread count does to it.  Diagnostic for bench.py's
cpu_baseline leg."""
import os, sys, time, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle

rs = np.random.RandomState(0)
nu, ni, d, B = 29858, 40981, 64, 4096
P = (rs.standard_normal((nu, d)) * 0.05).astype(np.float32); Q = (rs.standard_normal((ni, d)) * 0.05).astype(np.float32)
w = (rs.standard_normal(d) * 0.3).astype(np.float32); wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
u = rs.choice(nu, B, replace=False).astype(np.int32); i = rs.randint(0, ni, B).astype(np.int32); j = rs.randint(0, ni, B).astype(np.int32)


def t(fn, n=3):
    fn(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e3 * (time.perf_counter() - t0) / n


out = {"cpus": os.cpu_count(), "omp": os.environ.get("OMP_NUM_THREADS")}
eu, ei, ej = P[u], Q[i], Q[j]
out["pair_loss_grad_ms"] = t(lambda: oracle.pair_loss_grad(oracle.LOSS_RUBIBCEBOTH, eu, ei, ej, w, wu, 1e-2, 1e-3))
out["l2_reg_ms"] = t(lambda: oracle.l2_reg(eu, ei, ej, 1e-5, B))
L = oracle.lib()
m, v, g = np.zeros_like(P), np.zeros_like(P), np.zeros_like(P)
out["adam_dense_P_ms"] = t(lambda: L.orc_adam_dense(P, m, v, g.ctypes.data_as(ctypes.c_void_p), P.size, 1e-3, 0.9, 0.999, 1e-8))
st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
out["mf_train_step_ms"] = t(lambda: oracle.mf_train_step(oracle.LOSS_RUBIBCEBOTH, u, i, j, P, Q, w, wu, st, 1e-3, 1e-5, 1e-2, 1e-3, B))
if "--torch" in sys.argv:
    import torch
    from oracle import torch_port as tp
    port = tp.MFPort(P, Q, w, wu, 1e-3, 1e-5, 1e-2, 1e-3, B)
    out["torch_threads"] = torch.get_num_threads()
    out["torch_port_step_ms"] = t(lambda: port.train_step(1, u, i, j))
print(json.dumps(out))
