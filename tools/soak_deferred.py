import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macr_amd import ops, synth
cfg = synth.WORKLOADS["addressa"]; dev = torch.device("cuda", 0)
B, d = cfg["batch"], cfg["d"]
def mk():
    gen = torch.Generator(device=dev).manual_seed(7)
    P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
    w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    return ops.MFState(P, Q, w, wu, ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B), B), gen
a, gen = mk(); b, _ = mk()
batches = synth.train_batches(64, cfg["n_users"], cfg["n_items"], B, gen, dev, zipf=True, sort_by_pos=True)
la = torch.zeros((3000, 3), device=dev); lb = torch.zeros((3000, 3), device=dev)
for s in range(3000):
    k = s % 64
    a.step(ops.LOSS_RUBIBCEBOTH, batches[k, 0], batches[k, 1], batches[k, 2], la[s], defer=True)
    b.step(ops.LOSS_RUBIBCEBOTH, batches[k, 0], batches[k, 1], batches[k, 2], lb[s])
a.flush(); torch.cuda.synchronize()
la, lb = la.cpu().numpy(), lb.cpu().numpy()
rel = np.abs(la[:, 0] - lb[:, 0]) / np.abs(lb[:, 0])
print("loss first/last", lb[0, 0], lb[-1, 0], "finite", np.isfinite(la).all(), "max rel diff deferred vs complete", rel.max(), "at", rel.argmax())
print("param max abs diff P", float((a.P - b.P).abs().max()), "Q", float((a.Q - b.Q).abs().max()), "scale", float(b.P.abs().max()))
