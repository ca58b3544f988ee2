"""Host time per MF training step (enqueue only) against the step's GPU time: how far the Python loop is from being
the bottleneck.  python tools/host_cost.py"""
import sys, time, torch
sys.path.insert(0, ".")
from macr_amd import ops, synth
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
cfg = synth.WORKLOADS[wl]; B, d = cfg["batch"], cfg["d"]
gen = torch.Generator(device=dev).manual_seed(1)
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
state = ops.MFState(P, Q, w, wu, ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, B), B)
batches = synth.train_batches(64, cfg["n_users"], cfg["n_items"], B, gen, dev)
loss = torch.zeros((64, 3), dtype=torch.float32, device=dev)
def run(n):
    for k in range(n):
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k % 64, 0], batches[k % 64, 1], batches[k % 64, 2], loss[k % 64], defer=True)
run(50); state.flush(); torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter(); run(n); t1 = time.perf_counter()
state.flush(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.1f us per step; wall %.1f us per step" % (1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
