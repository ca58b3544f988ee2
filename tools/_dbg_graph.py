import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from macr_amd import ops, synth
from macr_amd.evaluator import Evaluator
wl = "gowalla"
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
uid = torch.from_numpy(users).to(dev)
k = 0
def train(n):
    global k
    for _ in range(n):
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k % 256, 0], batches[k % 256, 1], batches[k % 256, 2], loss, defer=True); k += 1
    state.flush()
train(45)
evs = {}
for name, graph in (("graph", True), ("direct", False)):
    ev = evs[name] = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev); ev.use_graph = graph; ev.use_seeds = False
    ev._stats_peek = True
def run(name):
    ev = evs[name]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ev.test_mf(ops.SCORE_RUBI_BOTH, state.P, uid, state.Q, [20], state.w, state.wu, cfg["c"])
    torch.cuda.synchronize(); t = 1e3 * (time.perf_counter() - t0)
    return round(t, 3), ev._stats.tolist(), float(r["recall"][0])
for n in (0, 200, 2000, 2000):
    train(n)
    for name in ("graph", "direct", "graph", "direct"):
        print("after", k, "steps", name, *run(name))
print("---- policy evaluator")
ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
evs["pol"] = ev
for i in range(8):
    train(20)
    t, st, rec = run("pol")
    print("eval", i, "t", t, "stats", st, "last_seeded", ev._last_seeded, "skip", ev._seed_skip, "backoff", ev._seed_backoff)
ev.use_graph = False
for i in range(4):
    train(20)
    ops.timing_begin()
    t, st, rec = run("pol")
    m = ops.timing_end()
    print("direct", i, "t", t, "stats", st, "last_seeded", ev._last_seeded, {n: round(1e3 * x) for n, x in m if x > 0.02})
