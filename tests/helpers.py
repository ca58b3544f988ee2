"""Shared test helpers: golden loading, regenerable score matrices, oracle-side evaluation pipelines."""
import json
import os
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def golden(half, dataset):
    return json.load(open(os.path.join(GOLD, "golden_%s_%s.json" % (half, dataset))))


def score_matrix(kind, n_users, n_items, seed):
    """Must stay identical to tests/golden/make_golden.py::score_matrix (frozen legacy RandomState)."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((n_users, n_items)).astype(np.float32)
    if kind == "normal":
        return x
    if kind == "ties":
        return (np.round(x * 2.0) / 2.0).clip(-2, 2).astype(np.float32)
    if kind == "popular":
        pop = rs.standard_normal(n_items).astype(np.float32) * 3.0
        return (x + pop[None, :]).astype(np.float32)
    raise ValueError(kind)


def dataset_args(dataset, **kw):
    if dataset == "tiny":
        root, name, bs = GOLD + "/", "tiny_data", 16
    else:
        root, name, bs = os.path.join(REPO, "data") + "/", dataset, 1024
    base = dict(data_path=root, dataset=name, batch_size=bs, data_type="ori", model="mf", source="normal",
                valid_set="test")
    base.update(kw)
    return types.SimpleNamespace(**base)


def masked_scores(scores, mask_lists):
    out = scores.copy()
    for q, row in enumerate(mask_lists):
        if len(row):
            out[q, np.asarray(row, dtype=np.int64)] = -np.inf
    return out
