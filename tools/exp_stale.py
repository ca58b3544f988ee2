"""Evaluator with STALE threshold seeds: per-kernel times and macr_score_topk stats of a seeded ranking after n training
steps (kernel-development helper).  python tools/exp_stale.py [workload] [steps...]"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator

wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
steps_list = [int(x) for x in sys.argv[2:]] or [20, 200, 2000]
dev = torch.device("cuda", 0)
cfg = synth.WORKLOADS[wl]
B, d = cfg["batch"], cfg["d"]
gen = torch.Generator(device=dev).manual_seed(12345)
P = synth.xavier_table(cfg["n_users"], d, gen, dev); Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
state = ops.MFState(P, Q, w, wu, ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B), B)
batches = synth.train_batches(256, cfg["n_users"], cfg["n_items"], B, gen, dev, zipf=True, sort_by_pos=False)
loss = torch.zeros(3, dtype=torch.float32, device=dev)
users, mask_lists, gt_lists = synth.eval_problem(cfg, seed=777)
ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
ev.use_graph = False
uid = torch.from_numpy(users).to(dev)
k = 0
def train(n):
    global k
    for _ in range(n):
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k % 256, 0], batches[k % 256, 1], batches[k % 256, 2], loss, defer=True); k += 1
    state.flush()
def evaluate(force_seeded):
    ev._seed_skip = 0 if force_seeded else 1
    ops.timing_begin()
    ev.test_mf(ops.SCORE_RUBI_BOTH, state.P, uid, state.Q, [20], state.w, state.wu, cfg["c"])
    marks = ops.timing_end()
    torch.cuda.synchronize()
    return {n: round(1e3 * t, 1) for n, t in marks}, ev._stats.tolist()
train(25); evaluate(False)
for n in steps_list:
    train(n)
    print(wl, "after", n, "steps, seeded  :", *evaluate(True))
    print(wl, "same tables,   unseeded:", *evaluate(False))
# random seeds: the stalest case
for key, sd in ev._seeds.items():
    sd.copy_(torch.randint(0, cfg["n_items"], sd.shape, device=dev, dtype=torch.int32))
print(wl, "random seeds:", *evaluate(True))
