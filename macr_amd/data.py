"""Data layer: the reference's two `Data` loaders and samplers, re-implemented.

Interfaces kept verbatim (attribute names, container types, call signatures,
RNG streams) so that callers written against the reference keep working:

  MFData    <-> macr_mf/load_data.py  Data(args)            (:24, ctor :504, load_ori_data :26, sample :543)
  LGCNData  <-> macr_lightgcn/utility/load_data.py Data(path, batch_size, args)
                                                            (:14, get_adj_mat :95, create_adj_mat :126,
                                                             sample :174, sample_test :214)

On-disk format (both): one line per user, "uid item item ...", single spaces,
ids dense from 0 (utils/data generation sample.py:394-424).

Differences from the reference, all deliberate:
  * nothing is ever written into the data directory (the reference's
    get_adj_mat saves four .npz files there, utility/load_data.py:105-107,:122);
  * the adjacency matrices are assembled with vectorised NumPy instead of
    lil/dok row slicing; the resulting CSR matrices are element-for-element
    identical (golden G4);
  * the samplers draw from Python's `random` / `numpy.random` in exactly the
    reference order (so a seeded run yields the same (u, i+, i-) stream --
    golden G2/G3) but test list membership through sets;
  * device-friendly CSR/int32 views of the interaction lists are provided for
    the HIP evaluator.
Out of scope: load_imb_data, the dice/npz source, the
CausalE/IPS loaders, plotting helpers.
"""
import collections
import random as rd

import numpy as np
import scipy.sparse as sp


def _read_lists(path):
    """[(uid, [items...]), ...] in file order; lines without items keep an empty list."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip("\n")
            if not line:
                continue
            parts = line.split(" ")
            rows.append((int(parts[0]), [int(x) for x in parts[1:] if x != ""]))
    return rows


class MFData(object):
    """Loader + sampler of the MF CLI (macr_mf/load_data.py:24-566, `mf` branch)."""

    def __init__(self, args):
        self.path = args.data_path + args.dataset + '/'
        self.batch_size = args.batch_size
        self.n_users, self.n_items, self.n_valid = 0, 0, 0
        self.n_train, self.n_test = 0, 0
        self.user_list, self.item_list = [], []
        self.valid_users, self.valid_items = set(), set()
        self.train_user_list = collections.defaultdict(list)
        self.test_user_list = collections.defaultdict(list)
        self.train_item_list = collections.defaultdict(list)
        self.test_item_list = collections.defaultdict(list)
        self.valid_user_list = collections.defaultdict(list)
        self.valid_item_list = collections.defaultdict(list)
        self.users, self.items = set(), set()
        if getattr(args, "data_type", "ori") != "ori":
            raise NotImplementedError("only --data_type ori is on the hot path (load_imb_data is out of scope)")
        self.load_ori_data(args)
        self.valid_users = list(self.valid_users)
        self.valid_items = list(self.valid_items)
        self._train_sets = {}
        print('n_items:', self.n_items, 'n_users:', self.n_users)
        total = sum(len(v) for v in self.train_item_list.values())
        print("sparsity:", 1.0 * total / self.n_items / self.n_users)

    # -- macr_mf/load_data.py:26-118 (mf branch, txt source) -------------------
    def load_ori_data(self, args):
        model = getattr(args, "model", "mf")
        if model not in ("mf", "biasmf") or getattr(args, "source", "normal") == "dice":
            raise NotImplementedError("only --model mf --source normal is on the hot path")
        max_u, max_i = 0, 0
        for user, items in _read_lists(self.path + 'train.txt'):
            if not items:
                continue
            self.train_user_list[user] = items
            for item in items:
                self.train_item_list[item].append(user)
            max_u, max_i = max(max_u, user), max(max_i, max(items))
            self.n_train += len(items)
        valid_set = getattr(args, "valid_set", "test")
        if valid_set == "valid":
            for user, items in _read_lists(self.path + 'valid.txt'):
                if not items:
                    continue
                self.valid_user_list[user] = items
                self.valid_items.update(items)
                for item in items:
                    self.valid_item_list[item].append(user)
                max_u, max_i = max(max_u, user), max(max_i, max(items))
                self.n_valid += len(items)
            self.valid_users = set(self.valid_user_list.keys())
        if valid_set == "test":
            for user, items in _read_lists(self.path + 'test.txt'):
                if not items:
                    continue
                self.test_user_list[user] = items
                for item in items:
                    self.test_item_list[item].append(user)
                max_u, max_i = max(max_u, user), max(max_i, max(items))
                self.n_test += len(items)
            self.test_users = set(self.test_user_list.keys())
        self.n_users, self.n_items = max_u + 1, max_i + 1
        self.users = list(range(self.n_users))
        self.items = list(range(self.n_items))

    # -- macr_mf/load_data.py:543-566 ------------------------------------------
    def sample(self):
        """One batch of (users, pos_items, neg_items) lists; same `random` stream as the reference:
        rd.sample / rd.choice over self.users, rd.choice over the user's train list, rejection
        sampling of the negative with rd.choice over self.items."""
        if self.batch_size <= self.n_users:
            users = rd.sample(self.users, self.batch_size)
        else:
            users = [rd.choice(self.users) for _ in range(self.batch_size)]
        pos_items, neg_items = [], []
        items, train, sets = self.items, self.train_user_list, self._train_sets
        for user in users:
            pos = train[user]                    # defaultdict access, as in the reference
            if pos == []:
                pos_items.append(0)
            else:
                pos_items.append(rd.choice(pos))
            seen = sets.get(user)
            if seen is None:
                seen = sets[user] = frozenset(pos)
            while True:
                neg_item = rd.choice(items)
                if neg_item not in seen:
                    neg_items.append(neg_item)
                    break
        return users, pos_items, neg_items

    # -- what batch_test.py:8 calls; popularity groups, pure bookkeeping (load_data.py:424-466)
    def plot_pics(self):
        def groups(sizes, points, denom):
            sizes = np.asarray(sizes)
            order = np.argsort(sizes)
            belong, count, p = [], [0] * 6, 0
            for score in sizes[order]:
                while p != 5 and points[p] < score:
                    p += 1
                count[p] += 1
                belong.append(p)
            return order, belong, [1.0 * c / denom for c in count]
        sorted_id, belong, rate = groups([len(u) for u in self.train_item_list.values()],
                                         [10, 50, 100, 200, 500], self.n_items)
        usorted_id, ubelong, urate = groups([len(i) for i in self.train_user_list.values()],
                                            [5, 7, 10, 15, 20], self.n_users)
        return sorted_id, belong, rate, usorted_id, ubelong, urate

    # -- device-side views --------------------------------------------------------
    def eval_lists(self, users, valid_set="test"):
        """(mask lists, ground-truth lists) for the given users, in order."""
        gt = self.test_user_list if valid_set == "test" else self.valid_user_list
        return [self.train_user_list.get(u, []) for u in users], [gt[u] for u in users]


class LGCNData(object):
    """Loader + adjacency builder + samplers of the LightGCN CLI (utility/load_data.py:14-254)."""

    def __init__(self, path, batch_size, args):
        self.path = path
        self.batch_size = batch_size
        train_file = path + '/train.txt'
        test_file = path + ('/test.txt' if getattr(args, "valid_set", "test") == "test" else '/valid.txt')
        self.n_users, self.n_items = 0, 0
        self.n_train, self.n_test = 0, 0
        self.neg_pools = {}
        self.exist_users = []
        self.train_items, self.test_set = {}, {}
        self.test_item_set = collections.defaultdict(list)
        max_u, max_i = 0, 0
        for uid, items in _read_lists(train_file):       # :30-40 count pass + :60-72 fill pass
            self.exist_users.append(uid)
            if items:
                max_i = max(max_i, max(items))
                self.train_items[uid] = items
            max_u = max(max_u, uid)
            self.n_train += len(items)
        for uid, items in _read_lists(test_file):        # :42-51 and :74-89
            if items:
                max_i = max(max_i, max(items))
                self.test_set[uid] = items
                for item in items:
                    self.test_item_set[item].append(uid)
            self.n_test += len(items)
        self.n_items, self.n_users = max_i + 1, max_u + 1
        self.print_statistics()
        rows = np.fromiter((u for u, its in self.train_items.items() for _ in its), dtype=np.int64)
        cols = np.fromiter((i for its in self.train_items.values() for i in its), dtype=np.int64)
        R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(self.n_users, self.n_items))
        R.data[:] = 1.0                                  # duplicates collapse to 1 like dok assignment
        self.R = R.todok()
        self._R_csr = R
        self._train_sets = {}

    def print_statistics(self):
        print('n_users=%d, n_items=%d' % (self.n_users, self.n_items))
        print('n_interactions=%d' % (self.n_train + self.n_test))
        print('n_train=%d, n_test=%d, sparsity=%.5f' % (self.n_train, self.n_test,
              (self.n_train + self.n_test) / (self.n_users * self.n_items)))

    # -- utility/load_data.py:95-164 ---------------------------------------------
    def create_adj_mat(self):
        """(plain, norm, mean): A=[[0,R],[R^T,0]]; norm = D^-1 (A+I) (float64, as the reference's
        float32 + sp.eye float64 promotes); mean = D^-1 A."""
        R = self._R_csr
        A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
        A.sort_indices()

        def normalized_adj_single(adj):
            rowsum = np.array(adj.sum(1))
            with np.errstate(divide="ignore"):
                d_inv = np.power(rowsum, -1).flatten()
            d_inv[np.isinf(d_inv)] = 0.
            return sp.diags(d_inv).dot(adj).tocoo()
        norm_adj = normalized_adj_single(A + sp.eye(A.shape[0]))
        mean_adj = normalized_adj_single(A)
        return A.tocsr(), norm_adj.tocsr(), mean_adj.tocsr()

    def get_adj_mat(self):
        """(plain, norm, mean, pre); pre = D^-1/2 A D^-1/2 in float32 (:109-121).  Writes nothing."""
        adj_mat, norm_adj_mat, mean_adj_mat = self.create_adj_mat()
        rowsum = np.array(adj_mat.sum(1))
        with np.errstate(divide="ignore"):
            d_inv = np.power(rowsum, -0.5).flatten()
        d_inv[np.isinf(d_inv)] = 0.
        d_mat_inv = sp.diags(d_inv)
        norm_adj = d_mat_inv.dot(adj_mat).dot(d_mat_inv)
        pre_adj_mat = norm_adj.tocsr()
        return adj_mat, norm_adj_mat, mean_adj_mat, pre_adj_mat

    # -- utility/load_data.py:174-212 --------------------------------------------
    def _seen(self, u):
        s = self._train_sets.get(u)
        if s is None:
            s = self._train_sets[u] = frozenset(self.train_items.get(u, ()))
        return s

    def sample(self):
        """Same streams as the reference: users from Python `random`, one positive and one rejection
        sampled negative per user from numpy.random.randint(size=1)."""
        if self.batch_size <= self.n_users:
            users = rd.sample(self.exist_users, self.batch_size)
        else:
            users = [rd.choice(self.exist_users) for _ in range(self.batch_size)]
        pos_items, neg_items = [], []
        randint = np.random.randint
        for u in users:
            pos = self.train_items[u]
            pos_items.append(pos[randint(low=0, high=len(pos), size=1)[0]])
            seen = self._seen(u)
            while True:
                neg_id = randint(low=0, high=self.n_items, size=1)[0]
                if neg_id not in seen:
                    neg_items.append(neg_id)
                    break
        return users, pos_items, neg_items

    # -- utility/load_data.py:214-254 --------------------------------------------
    def sample_test(self):
        keys = list(self.test_set.keys())                # rd.sample(dict_keys) == rd.sample(tuple(keys))
        if self.batch_size <= self.n_users:
            users = rd.sample(keys, self.batch_size)
        else:
            users = [rd.choice(keys) for _ in range(self.batch_size)]
        pos_items, neg_items = [], []
        randint = np.random.randint
        for u in users:
            pos = self.test_set[u]
            pos_items.append(pos[randint(low=0, high=len(pos), size=1)[0]])
            union = set(pos) | self._seen(u)
            while True:
                neg_id = randint(low=0, high=self.n_items, size=1)[0]
                if neg_id not in union:
                    neg_items.append(neg_id)
                    break
        return users, pos_items, neg_items

    def eval_lists(self, users):
        return [self.train_items.get(u, []) for u in users], [self.test_set[u] for u in users]
