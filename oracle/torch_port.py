"""TEST / BENCH INFRASTRUCTURE ONLY -- the reference's CPU path restated op for op on torch-CPU fp32.

What `bench.py`'s `cpu_baseline` leg times on the GPU box's host cores ("CPU restatement of the
reference path"), and what `tests/test_torch_port_cpu.py` checks against the C oracle (oracle/macr_oracle.c).  Nothing
under macr_amd/, macr_mf/ or macr_lightgcn/ imports this module.

The reference runs `sess.run([opt, loss, mf_loss, reg_loss], feed)` on a TF1 graph (macr_mf/train.py:487-496): gathers,
dense (B,B) intermediates, reverse-mode gradients, `tf.train.AdamOptimizer` on every row of every table.  TensorFlow 1.14
is not installable here (no network), so this is the same GRAPH -- the same tensors materialised, the same
reductions, autograd instead of hand-derived gradients, Adam in TF's form -- on another multi-threaded CPU tensor library.
It is a baseline, not a checker: intra-op threading and summation order are torch's.

  macr_mf/model.py:35-37     gathers                    -> index_select
  macr_mf/model.py:185-222   rubibceboth loss           -> mf_loss()           ((B,1)*(B,) broadcast quirk kept: :204-205)
  macr_mf/model.py:277-287   normalbce loss             -> mf_loss()
  macr_mf/model.py:74/:95    AdamOptimizer.minimize     -> adam_tf()           (lr_t form, eps outside the root, dense)
  macr_lightgcn/LightGCN.py:288-309  propagation        -> lgcn_propagate()    (torch.sparse CSR @ dense)
  macr_lightgcn/LightGCN.py:495-532  bceboth on propagated rows, reg on ego rows -> lgcn_train_step()
  macr_mf/model.py:45,:199   batch_ratings / rubi_ratings_both -> score_matrix()
  macr_mf/train.py:222-259   per-batch scoring + ranking -> evaluate_mf()      (torch.topk for heapq.nlargest)
"""
import numpy as np
import torch

LOSS_NORMALBCE, LOSS_RUBIBCEBOTH = 0, 1


class AdamTF(object):
    """tf.train.AdamOptimizer(lr) state of one `minimize` call: slots m, v per variable and the two beta powers
    (fp32, multiplied once per step) -- SURVEY.md A.2."""

    def __init__(self, params, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.params = params
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.pow1 = torch.tensor(b1, dtype=torch.float32)
        self.pow2 = torch.tensor(b2, dtype=torch.float32)

    @torch.no_grad()
    def step(self, grads):
        lr_t = self.lr * torch.sqrt(1.0 - self.pow2) / (1.0 - self.pow1)
        for p, m, v, g in zip(self.params, self.m, self.v, grads):
            if g is None:                                   # a variable without a gradient is untouched (model.py:117-120)
                continue
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.sub_(lr_t * m / (v.sqrt() + self.eps))
        self.pow1 *= self.b1
        self.pow2 *= self.b2


def mf_loss(kind, eu, ei, ej, w, wu, alpha, beta, decay, batch_size_cfg, reg_rows=None):
    """(loss, mf_loss, reg_loss) on gathered rows (B,d).  reg_rows: the rows the regulariser is taken on (LightGCN: the
    EGO rows, LightGCN.py:525-527); default: the same rows."""
    p = (eu * ei).sum(1)                                    # model.py:186
    n = (eu * ej).sum(1)                                    # :187
    if kind == LOSS_NORMALBCE:
        mf = torch.mean(-torch.log(torch.sigmoid(p) + 1e-9) - torch.log(1.0 - torch.sigmoid(n) + 1e-9))     # :278-280
    else:
        si, sj, su = ei @ w, ej @ w, eu @ wu                # (B,1)  :194-196
        # (B,) * (B,1) * (B,1) -> (B,B): element [r,c] = p[c] * sig(si[r]) * sig(su[r])   (:204-205)
        X = p * torch.sigmoid(si) * torch.sigmoid(su)
        Y = n * torch.sigmoid(sj) * torch.sigmoid(su)
        l_ori = torch.mean(-torch.log(torch.sigmoid(X) + 1e-10) - torch.log(1.0 - torch.sigmoid(Y) + 1e-10))       # :211
        l_item = torch.mean(-torch.log(torch.sigmoid(si) + 1e-10) - torch.log(1.0 - torch.sigmoid(sj) + 1e-10))    # :213
        l_user = torch.mean(-torch.log(torch.sigmoid(su) + 1e-10) - torch.log(1.0 - torch.sigmoid(su) + 1e-10))    # :215
        mf = l_ori + alpha * l_item + beta * l_user         # :217
    ru, ri, rj = reg_rows if reg_rows is not None else (eu, ei, ej)
    regularizer = (0.5 * (ru * ru).sum() + 0.5 * (ri * ri).sum() + 0.5 * (rj * rj).sum()) / batch_size_cfg   # :219-220
    reg = decay * regularizer
    return mf + reg, mf, reg


class MFPort(object):
    """BPRMF on torch-CPU: tables P, Q, branch vectors w (d,1), w_user (d,1), one AdamTF per loss kind."""

    def __init__(self, P, Q, w, wu, lr, decay, alpha, beta, batch_size_cfg):
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
        self.P, self.Q = t(P).requires_grad_(), t(Q).requires_grad_()
        self.w, self.wu = t(w).reshape(-1, 1).requires_grad_(), t(wu).reshape(-1, 1).requires_grad_()
        self.decay, self.alpha, self.beta, self.bs = decay, alpha, beta, batch_size_cfg
        self.opt = {k: AdamTF([self.P, self.Q, self.w, self.wu], lr) for k in (LOSS_NORMALBCE, LOSS_RUBIBCEBOTH)}

    def train_step(self, kind, u, i, j):
        u, i, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, j))
        eu, ei, ej = self.P.index_select(0, u), self.Q.index_select(0, i), self.Q.index_select(0, j)      # :35-37
        loss, mf, reg = mf_loss(kind, eu, ei, ej, self.w, self.wu, self.alpha, self.beta, self.decay, self.bs)
        # tf.gradients: IndexedSlices for the tables, summed over duplicate indices and applied densely (A.2)
        grads = torch.autograd.grad(loss, [self.P, self.Q, self.w, self.wu], allow_unused=True)
        self.opt[kind].step(grads)
        return np.array([loss.item(), mf.item(), reg.item()], np.float32)

    # ------------------------------------------------------------------ evaluation (train.py:162-311, one user batch at a time)
    @torch.no_grad()
    def score_matrix(self, users, c, rubi=True):
        eu = self.P.index_select(0, torch.as_tensor(np.asarray(users), dtype=torch.long))
        S = eu @ self.Q.t()                                 # batch_ratings  model.py:45
        if rubi:                                            # rubi_ratings_both  :199
            S = (S - c) * torch.sigmoid(self.Q @ self.w).t() * torch.sigmoid(eu @ self.wu)
        return S

    @torch.no_grad()
    def evaluate(self, users, mask_lists, c, K=20, batch=4096, rubi=True):
        """top-K ids per query user (train items excluded), user batches of `batch` like train.py:222-259.
        heapq.nlargest over the candidate dict becomes torch.topk on the masked score rows."""
        out = np.empty((len(users), K), np.int64)
        for a in range(0, len(users), batch):
            S = self.score_matrix(users[a:a + batch], c, rubi)
            rows = np.repeat(np.arange(S.shape[0]), [len(m) for m in mask_lists[a:a + batch]])
            cols = np.concatenate([np.asarray(m, np.int64) for m in mask_lists[a:a + batch]]) if len(rows) else rows
            S[torch.from_numpy(rows), torch.from_numpy(cols)] = -float("inf")
            out[a:a + batch] = torch.topk(S, K, dim=1).indices.numpy()
        return out


def csr_to_torch(indptr, indices, data, N):
    return torch.sparse_csr_tensor(torch.as_tensor(np.asarray(indptr), dtype=torch.int64),
                                   torch.as_tensor(np.asarray(indices), dtype=torch.int64),
                                   torch.as_tensor(np.asarray(data, dtype=np.float32)), size=(N, N))


def lgcn_propagate(A, T, n_layers):
    """E = mean(E0, A E0, A^2 E0, ...)  (LightGCN.py:288-309; the 100 row folds concatenate to one product)."""
    embs, cur = [T], T
    for _ in range(n_layers):
        cur = torch.sparse.mm(A, cur)                       # tf.sparse_tensor_dense_matmul  :301
        embs.append(cur)
    return torch.stack(embs, 1).mean(1)                     # :306-307


class LGCNPort(object):
    def __init__(self, T, n_users, n_items, w, wu, A, n_layers, lr, decay, alpha, beta, batch_size_cfg):
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
        self.T = t(T).requires_grad_()
        self.w, self.wu = t(w).reshape(-1, 1).requires_grad_(), t(wu).reshape(-1, 1).requires_grad_()
        self.n_users, self.n_items, self.A, self.L = n_users, n_items, A, n_layers
        self.decay, self.alpha, self.beta, self.bs = decay, alpha, beta, batch_size_cfg
        self.opt = {k: AdamTF([self.T, self.w, self.wu], lr) for k in (LOSS_NORMALBCE, LOSS_RUBIBCEBOTH)}

    def train_step(self, kind, u, i, j):
        u, i, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, j))
        E = lgcn_propagate(self.A, self.T, self.L)
        sel = lambda tab, idx, off: tab.index_select(0, idx + off)
        nu = self.n_users
        loss, mf, reg = mf_loss(kind, sel(E, u, 0), sel(E, i, nu), sel(E, j, nu), self.w, self.wu, self.alpha, self.beta,
                                self.decay, self.bs, reg_rows=(sel(self.T, u, 0), sel(self.T, i, nu), sel(self.T, j, nu)))
        grads = torch.autograd.grad(loss, [self.T, self.w, self.wu], allow_unused=True)
        self.opt[kind].step(grads)
        return np.array([loss.item(), mf.item(), reg.item()], np.float32)

    @torch.no_grad()
    def score_matrix(self, E, users, c, rubi=True):
        eu = E.index_select(0, torch.as_tensor(np.asarray(users), dtype=torch.long))
        ia = E[self.n_users:]
        S = eu @ ia.t()                                     # batch_ratings  LightGCN.py:166
        if rubi:                                            # rubi_ratings_both  :509
            S = (S - c) * torch.sigmoid(ia @ self.w).t() * torch.sigmoid(eu @ self.wu)
        return S
