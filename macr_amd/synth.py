"""Synthetic workloads with the shapes of BASELINE.json's configs.

Only Addressa ships a train.txt; Gowalla / ML-10M / Yelp2018 are test-only in the
reference checkout and there is no network, so the benchmark tables and
interaction lists are generated here, seeded, directly on the device:
  users uniform without replacement, positive items Zipf(1.0), negatives uniform,
  train-list lengths ~ lognormal clipped to [1, n_items/2], embeddings
  Xavier-uniform U(-sqrt(6/(rows+d)), +sqrt(6/(rows+d))).
"""
import math

import numpy as np
import torch

WORKLOADS = {
    # name: n_users, n_items, d, batch, n_train(approx), test users, test items/user, hyper-parameters
    "addressa": dict(n_users=13485, n_items=744, d=64, batch=1024, n_train=113345, n_test_users=2090,
                     test_per_user=2, alpha=1e-3, beta=1e-3, lr=1e-3, regs=1e-5, c=40.0),
    "gowalla": dict(n_users=29858, n_items=40981, d=64, batch=4096, n_train=822358, n_test_users=15424,
                    test_per_user=13, alpha=1e-2, beta=1e-3, lr=1e-3, regs=1e-5, c=40.0),
    "ml10m": dict(n_users=69166, n_items=8790, d=64, batch=8192, n_train=4900000, n_test_users=13878,
                  test_per_user=6, alpha=1e-3, beta=1e-3, lr=1e-3, regs=1e-5, c=40.0),
    "yelp2018": dict(n_users=31668, n_items=38048, d=64, batch=4096, n_train=1371000, n_test_users=13957,
                     test_per_user=14, alpha=1e-2, beta=1e-3, lr=1e-3, regs=1e-5, c=40.0),
}


def xavier_table(rows, d, gen, device):
    limit = math.sqrt(6.0 / (rows + d))
    return ((torch.rand((rows, d), generator=gen, device=device, dtype=torch.float32) * 2 - 1) * limit).contiguous()


def zipf_cdf(n, device, s=1.0):
    """the Zipf law's CDF, summed on the HOST in float64: a device cumsum (decoupled look-back scan) is not bit-reproducible
    from run to run, and ranks that must draw the SAME batch (row-sharded training) would differ in a sample now and then"""
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** s
    cdf = np.cumsum(p / p.sum())
    cdf[-1] = 1.0
    return torch.from_numpy(cdf).to(device)


def train_batches(n_steps, n_users, n_items, batch, gen, device, zipf=True, sort_by_pos=False):
    """(n_steps,3,B) int32: users w/o replacement per step, Zipf positives, uniform negatives.  A function of the
    generator's seed alone (inverse-CDF draws against a host-summed CDF: see zipf_cdf).
    sort_by_pos: pre-order the triples of every batch by positive item id (diagnostic only: the training step orders
    its batch itself, on the device, so benchmarks feed batches exactly as a sampler emits them)."""
    cdf = zipf_cdf(n_items, device) if zipf else None
    out = torch.empty((n_steps, 3, batch), dtype=torch.int32, device=device)
    for s in range(n_steps):
        if batch <= n_users:
            u = torch.randperm(n_users, generator=gen, device=device)[:batch]
        else:
            u = torch.randint(0, n_users, (batch,), generator=gen, device=device)
        out[s, 0] = u.to(torch.int32)
        if zipf:
            r = torch.rand((batch,), generator=gen, device=device, dtype=torch.float64)
            out[s, 1] = torch.searchsorted(cdf, r, right=True).clamp_(max=n_items - 1).to(torch.int32)
        else:
            out[s, 1] = torch.randint(0, n_items, (batch,), generator=gen, device=device).to(torch.int32)
        out[s, 2] = torch.randint(0, n_items, (batch,), generator=gen, device=device).to(torch.int32)
        if sort_by_pos:
            order = torch.argsort(out[s, 1], stable=True)
            out[s] = out[s][:, order]
    return out


def interaction_lists(n_rows, n_items, mean_len, seed, zipf=True):
    """list of sorted unique item lists (host, NumPy): lengths lognormal(mean_len), items Zipf."""
    rs = np.random.RandomState(seed)
    sigma = 0.9
    mu = math.log(max(mean_len, 1.0)) - sigma * sigma / 2
    lens = np.clip(np.round(rs.lognormal(mu, sigma, n_rows)), 1, max(1, n_items // 2)).astype(np.int64)
    if zipf:
        p = 1.0 / np.arange(1, n_items + 1)
        cdf = np.cumsum(p / p.sum())
    lists = []
    for n in lens:
        if zipf:
            draw = np.searchsorted(cdf, rs.rand(int(n * 1.5) + 4))
            items = np.unique(np.minimum(draw, n_items - 1))[:n]
        else:
            items = np.unique(rs.randint(0, n_items, int(n)))
        lists.append(items.tolist())
    return lists


def eval_problem(cfg, seed):
    """(query user ids, mask lists (train), ground-truth lists (test)) for a workload."""
    rs = np.random.RandomState(seed)
    U = cfg["n_test_users"]
    users = np.sort(rs.choice(cfg["n_users"], U, replace=False)).astype(np.int32)
    mean_train = cfg["n_train"] / cfg["n_users"]
    mask = interaction_lists(U, cfg["n_items"], mean_train, seed + 1)
    gt = []
    for row in mask:
        seen = set(row)
        t = []
        while len(t) < cfg["test_per_user"]:
            x = int(rs.randint(0, cfg["n_items"]))
            if x not in seen:
                t.append(x); seen.add(x)
        gt.append(sorted(t))
    return users, mask, gt


def lgcn_graph(cfg, seed=9):
    """Synthetic interaction graph of a workload's shape for the LightGCN path: (train lists per user, the `pre` adjacency
    D^-1/2 A D^-1/2 as scipy CSR fp32 of shape (n_users+n_items)^2 -- macr_lightgcn/utility/load_data.py:112-121 --
    with sorted column indices, nnz = 2 n_train)."""
    import scipy.sparse as sp
    n_u, n_i = cfg["n_users"], cfg["n_items"]
    lists = interaction_lists(n_u, n_i, cfg["n_train"] / n_u, seed=seed)
    rows = np.repeat(np.arange(n_u), [len(l) for l in lists])
    cols = np.concatenate(lists)
    R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_u, n_i))
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5).astype(np.float32)
    dinv[np.isinf(dinv)] = 0
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32)
    A.sort_indices()
    return lists, A
