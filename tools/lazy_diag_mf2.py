import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_lazy_adam import _mf_states
from macr_amd import ops
d, kind = 32, ops.LOSS_RUBIBCEBOTH
n_users, n_items, B = 2000, 1500, 128
periods = [int(x) for x in sys.argv[1].split(',')]
states, rs = _mf_states(n_users, n_items, d, B, periods)
states = [states[0], states[-1]]
u = rs.choice(n_users, B, replace=False).astype(np.int32)
ij = rs.choice(n_items, 2 * B, replace=False).astype(np.int32)
b = [torch.from_numpy(a).cuda() for a in (u, ij[:B], ij[B:])]
P0, Q0 = states[0].P.clone(), states[0].Q.clone()
print("tables equal before", torch.equal(states[0].P, states[1].P), torch.equal(states[0].Q, states[1].Q), torch.equal(states[0].w, states[1].w))
for s in states:
    s.step(kind, *b, defer=True)
torch.cuda.synchronize()
print("tables equal after", torch.equal(states[0].P, states[1].P), torch.equal(states[0].P, P0))
Bp = 256
off = 64 + 8 * 2 * d
f0 = states[0].ws.view(torch.float32)[off:off + 7 * Bp].view(7, Bp)[:, :B]
f1 = states[1].ws.view(torch.float32)[off:off + 7 * Bp].view(7, Bp)[:, :B]
pref = (P0[b[0].long()].double() * Q0[b[1].long()].double()).sum(1)
print("p dense", f0[0, :4].tolist()); print("p lazy ", f1[0, :4].tolist()); print("p ref  ", pref[:4].tolist())
print("err dense", float((f0[0].double() - pref).abs().max()), "err lazy", float((f1[0].double() - pref).abs().max()))
print("rows differing per array", [(int((f0[k] != f1[k]).sum())) for k in range(7)])

print("gP equal", torch.equal(states[0].gP, states[1].gP), "gw equal", torch.equal(states[0].ws.view(torch.float32)[64:off], states[1].ws.view(torch.float32)[64:off]))
x, y = states[0].ws.view(torch.float32), states[1].ws.view(torch.float32)
neq = (x != y) & ~(torch.isnan(x) & torch.isnan(y))
print("ws regions differing", sorted(set((neq.nonzero().flatten() // 64).tolist())))
