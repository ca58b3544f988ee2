"""a1: the initialiser (macr_mf/model.py:107-122, :59-60 -- tf.contrib.layers.xavier_initializer()): uniform(-L, L) with
L = sqrt(6 / (fan_in + fan_out)), fan_in = rows, fan_out = cols of the 2-D shape.  TF's Philox stream
itself cannot be replayed; what is pinned here is the distribution."""
import math

import numpy as np
import torch

from macr_amd.mf import xavier_uniform
from macr_amd import synth


def test_xavier_uniform_bound_mean_variance():
    for shape in ((13485, 64), (744, 64), (64, 1), (29858, 128)):
        gen = torch.Generator().manual_seed(12345)
        x = xavier_uniform(shape, gen, torch.device("cpu")).numpy().astype(np.float64)
        L = math.sqrt(6.0 / (shape[0] + shape[1]))
        assert x.shape == shape and np.abs(x).max() <= L
        n = x.size
        if n >= 10000:
            assert np.abs(x).max() >= 0.999 * L                         # the whole interval is used
            assert abs(x.mean()) < 4 * (L / math.sqrt(3)) / math.sqrt(n)  # mean 0 within 4 standard errors
            assert abs(x.var() / (L * L / 3.0) - 1.0) < 0.02            # variance of U(-L, L) is L^2 / 3
    assert math.isclose(math.sqrt(6.0 / 65), 0.3038, abs_tol=1e-4)     # the branch vectors' limit quoted in SURVEY.md A.3


def test_xavier_uniform_is_seeded():
    a = xavier_uniform((100, 64), torch.Generator().manual_seed(7), torch.device("cpu"))
    b = xavier_uniform((100, 64), torch.Generator().manual_seed(7), torch.device("cpu"))
    c = xavier_uniform((100, 64), torch.Generator().manual_seed(8), torch.device("cpu"))
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_bench_tables_follow_the_same_law():
    gen = torch.Generator().manual_seed(1)
    t = synth.xavier_table(5000, 64, gen, torch.device("cpu")).numpy()
    L = math.sqrt(6.0 / (5000 + 64))
    assert np.abs(t).max() <= L and np.abs(t).max() > 0.99 * L
