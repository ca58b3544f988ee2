import sys, torch, numpy as np
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
ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev); ev.use_graph = False
uid = torch.from_numpy(users).to(dev)
k = 0
def train(n):
    global k
    for _ in range(n):
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k % 256, 0], batches[k % 256, 1], batches[k % 256, 2], loss, defer=True); k += 1
    state.flush()
def scores():
    si = torch.sigmoid(state.Q @ state.w); su = torch.sigmoid(state.P[uid.long()] @ state.wu)
    return ((state.P[uid.long()] @ state.Q.T) - cfg["c"]) * si[None, :] * su[:, None]
train(25)
ev._seed_skip = 1
ev.test_mf(ops.SCORE_RUBI_BOTH, state.P, uid, state.Q, [20], state.w, state.wu, cfg["c"])
seed = list(ev._seeds.values())[0].clone().long()
for n in (20, 200):
    train(n)
    S = scores()
    tau = S.gather(1, seed).min(1).values
    cnt = (S >= tau[:, None]).sum(1)
    blk = (cnt.reshape(-1)[: (len(cnt) // 128) * 128].reshape(-1, 128) > 1024).any(1).sum().item()
    print("after", n, "steps: candidates per user: median", cnt.median().item(), "p99", cnt.float().quantile(0.99).item(), "max", cnt.max().item(),
          "users > 1024:", (cnt > 1024).sum().item(), "> 512:", (cnt > 512).sum().item(), "blocks with >1024:", blk)
    ev._seed_skip = 0
    ev.test_mf(ops.SCORE_RUBI_BOTH, state.P, uid, state.Q, [20], state.w, state.wu, cfg["c"])
    torch.cuda.synchronize(); print("stats", ev._stats.tolist())
    seed = list(ev._seeds.values())[0].clone().long()
