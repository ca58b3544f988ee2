"""Device-side training sampler (macr_sample_triples) -- `--sampler device` of the CLIs.

The default `--sampler reference` keeps the reference's python `random` / numpy streams (golden G2/G3)
and is host-bound at ~1 M triples/s; this one draws every batch on the GPU from a counter-based
generator keyed by (seed, step): same distribution, different stream."""
import ctypes

import torch

from . import _lib
from .ops import CSR, _ptr, _stream, check


class DeviceSampler(object):
    def __init__(self, train_lists, n_users, n_items, batch_size, device, seed=12345, pool=None, ahead=32, exclude=None):
        """train_lists: {user: [items]} (dict or list indexed by user id); pool: user ids to draw from
        (LightGCN draws from `exist_users`, MF from range(n_users)).  ahead: batches drawn per launch by sample()
        (macr_sample_triples_many; the batch of step k is the same whatever `ahead` is).  exclude: {user: [items]} a
        negative must avoid when that is not the list the positives come from (LightGCN's sample_test: positives from
        the test lists, negatives outside test and train lists); needs ahead > 1."""
        rows = [train_lists.get(u, []) if isinstance(train_lists, dict) else train_lists[u] for u in range(n_users)]
        self.csr = CSR.from_lists(rows, device)
        self.excl = None
        if exclude is not None:
            ex = [exclude.get(u, []) if isinstance(exclude, dict) else exclude[u] for u in range(n_users)]
            self.excl = CSR.from_lists([sorted(set(r)) for r in ex], device)
        self.n_items, self.batch_size, self.seed, self.step = n_items, batch_size, int(seed), 0
        self.pool = None if pool is None else torch.as_tensor(list(pool), dtype=torch.int32, device=device)
        self.n_pool = n_users if pool is None else len(pool)
        self.device = device
        self.ahead = max(1, int(ahead))
        self._ring, self._ring_step0, self._ring_buf = None, -1, 0

    def sample(self, out=None):
        """-> (3,B) int32 device tensor (users, pos_items, neg_items); advances the step counter.
        out=None: a view into a buffer of `ahead` batches drawn by one launch (valid until 2*ahead further calls: two
        buffers alternate); out given: that tensor is filled by a launch of its own."""
        if out is None and self.ahead > 1:
            if self._ring is None:
                self._ring = torch.empty((2, self.ahead, 3, self.batch_size), dtype=torch.int32, device=self.device)
            k = self.step - self._ring_step0
            if self._ring_step0 < 0 or k >= self.ahead or k < 0:
                self._ring_buf ^= 1
                self._ring_step0, k = self.step, 0
                check(_lib.lib().macr_sample_triples_many(
                    ctypes.c_uint64(self.seed), ctypes.c_uint64(self.step), self.ahead, self.batch_size, self.n_items,
                    _ptr(self.pool, torch.int32, True), self.n_pool, _ptr(self.csr.ptr, torch.int32),
                    _ptr(self.csr.idx, torch.int32), None if self.excl is None else _ptr(self.excl.ptr, torch.int32),
                    None if self.excl is None else _ptr(self.excl.idx, torch.int32),
                    _ptr(self._ring[self._ring_buf], torch.int32), _stream()))
            self.step += 1
            return self._ring[self._ring_buf, k]
        if self.excl is not None:
            raise ValueError("DeviceSampler(exclude=...) draws through macr_sample_triples_many only: ahead > 1, out=None")
        if out is None:
            out = torch.empty((3, self.batch_size), dtype=torch.int32, device=self.device)
        check(_lib.lib().macr_sample_triples(
            ctypes.c_uint64(self.seed), ctypes.c_uint64(self.step), self.batch_size, self.n_items,
            _ptr(self.pool, torch.int32, True), self.n_pool, _ptr(self.csr.ptr, torch.int32),
            _ptr(self.csr.idx, torch.int32), _ptr(out, torch.int32), _stream()))
        self.step += 1
        return out
