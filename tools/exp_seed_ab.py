"""A/B of the evaluator's threshold modes under identical conditions (graph replays, n training steps between two
evaluations): seeded / sampled with the seed bookkeeping / use_seeds = False.  python tools/exp_seed_ab.py [workload] [steps]"""
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator

wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
n_between = int(sys.argv[2]) if len(sys.argv) > 2 else 20
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
uid = torch.from_numpy(users).to(dev)
k = 0
def train(n):
    global k
    for _ in range(n):
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k % 256, 0], batches[k % 256, 1], batches[k % 256, 2], loss, defer=True); k += 1
    state.flush()
def evaluate(mode):
    ev.use_seeds = mode != "off"
    ev._seed_skip, ev._seed_backoff = (1 if mode == "sampled" else 0), 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.test_mf(ops.SCORE_RUBI_BOTH, state.P, uid, state.Q, [20], state.w, state.wu, cfg["c"])
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), ev._stats.tolist()
train(300)
for m in ("sampled", "seeded", "off", "seeded", "sampled", "off"):
    evaluate(m)
res = {"seeded": [], "sampled": [], "off": []}
rel = []
for r in range(30):
    for m in ("seeded", "sampled", "off"):
        train(n_between)
        t, st = evaluate(m)
        res[m].append(t)
        if m == "seeded": rel.append(st[0])
print(wl, "steps between", n_between, {m: (round(float(np.median(v)), 4), round(float(np.min(v)), 4)) for m, v in res.items()}, "relisted (seeded runs):", sorted(set(rel)))
