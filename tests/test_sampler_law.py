"""The device sampler's checker (oracle.sample_triples = macr_amd/csrc/sample_kernels.hip restated) against
the LAW of the reference samplers: macr_mf/load_data.py:543-566 (uniform positive of the user's train list, item 0 for an
empty list, rejection-sampled uniform negative, B distinct users) and macr_lightgcn/utility/load_data.py:174-254.
Exact chi-square tests of each marginal, and two-sample tests against the reference's own sampler stream (macr_amd.data is
that stream: tests/test_data.py compares it with golden G2/G3 bit for bit).  All seeds are fixed: nothing here is flaky."""
import random

import numpy as np
import pytest
from scipy import stats

import oracle
from helpers import dataset_args

P_MIN = 1e-3          # a correct sampler fails one of these fixed-seed tests with probability ~1e-3 x (number of tests) -- once, ever


def tiny_problem(seed=0, n_users=64, n_items=50):
    rs = np.random.RandomState(seed)
    lists = [sorted(rs.choice(n_items, size=rs.randint(1, 20), replace=False).tolist()) for _ in range(n_users)]
    lists[3] = []                                    # empty list -> positive 0 (load_data.py:551-552)
    lists[5] = list(range(n_items - 1))              # one admissible negative only
    return lists, oracle.csr_from_lists(lists), n_items


def test_structure_and_determinism():
    lists, csr, n_items = tiny_problem()
    a = oracle.sample_triples(7, 0, 32, n_items, csr)
    assert np.array_equal(a, oracle.sample_triples(7, 0, 32, n_items, csr))
    assert not np.array_equal(a, oracle.sample_triples(7, 1, 32, n_items, csr))
    assert not np.array_equal(a, oracle.sample_triples(8, 0, 32, n_items, csr))
    for step in range(50):
        u, i, j = oracle.sample_triples(7, step, 32, n_items, csr)
        assert len(set(u.tolist())) == 32                        # rd.sample: B distinct users
        for uu, ii, jj in zip(u, i, j):
            assert jj not in lists[uu]
            assert (ii in lists[uu]) or (lists[uu] == [] and ii == 0)
    u, i, j = oracle.sample_triples(7, 0, 64, n_items, csr)       # B = pool: a permutation of all users
    assert sorted(u.tolist()) == list(range(64))
    u, i, j = oracle.sample_triples(7, 0, 500, n_items, csr)      # B > pool: with replacement (rd.choice branch, :546-547)
    assert set(u.tolist()) <= set(range(64)) and len(set(u.tolist())) > 32
    pool = np.arange(0, 64, 2, dtype=np.int32)                    # LightGCN: users from exist_users only
    u, i, j = oracle.sample_triples(7, 0, 16, n_items, csr, pool=pool)
    assert set(u.tolist()) <= set(pool.tolist()) and len(set(u.tolist())) == 16


def test_users_uniform_without_replacement():
    lists, csr, n_items = tiny_problem()
    B, steps = 16, 4000
    counts = np.zeros(64)
    first = np.zeros(64)
    for step in range(steps):
        u = oracle.sample_triples(11, step, B, n_items, csr)[0]
        counts[u] += 1
        first[u[0]] += 1
    # every user is in a batch with probability B / n_users; the batch's first slot is uniform over the users
    assert stats.chisquare(first).pvalue > P_MIN
    z = (counts - steps * B / 64) / np.sqrt(steps * (B / 64) * (1 - B / 64))
    assert np.abs(z).max() < 4.5 and stats.kstest(z, "norm").pvalue > P_MIN


def test_positive_uniform_over_the_train_list():
    lists, csr, n_items = tiny_problem(seed=1)
    steps = 6000
    hist = {u: np.zeros(len(l)) for u, l in enumerate(lists) if len(l) > 1}
    for step in range(steps):
        u, i, j = oracle.sample_triples(5, step, 64, n_items, csr)
        for uu, ii in zip(u, i):
            if uu in hist:
                hist[uu][lists[uu].index(ii)] += 1
    ps = [stats.chisquare(h).pvalue for h in hist.values()]
    assert min(ps) > P_MIN / len(ps)                                           # every user's list
    assert stats.kstest(ps, "uniform").pvalue > P_MIN                          # and the p-values themselves look uniform


def test_negative_uniform_over_the_complement():
    lists, csr, n_items = tiny_problem(seed=2)
    steps = 6000
    hist = np.zeros((64, n_items))
    for step in range(steps):
        u, i, j = oracle.sample_triples(9, step, 64, n_items, csr)
        hist[u, j] += 1
    ps = []
    for u, l in enumerate(lists):
        comp = np.setdiff1d(np.arange(n_items), l)
        assert hist[u, l].sum() == 0
        if len(comp) > 1:
            ps.append(stats.chisquare(hist[u, comp]).pvalue)
        else:
            assert hist[u, comp[0]] == steps                                   # the single admissible item, every time
    assert min(ps) > P_MIN / len(ps) and stats.kstest(ps, "uniform").pvalue > P_MIN


def test_sample_test_law_with_an_exclusion_list():
    """LightGCN's sample_test (utility/load_data.py:214-254): positives from the TEST lists, negatives outside test and
    train lists, users from the test users."""
    rs = np.random.RandomState(4)
    n_users, n_items = 48, 40
    train = [sorted(rs.choice(n_items, size=rs.randint(1, 12), replace=False).tolist()) for _ in range(n_users)]
    test = {u: sorted(rs.choice(np.setdiff1d(np.arange(n_items), train[u]), size=3, replace=False).tolist())
            for u in range(0, n_users, 2)}
    tl = [test.get(u, []) for u in range(n_users)]
    ex = [sorted(set(tl[u]) | set(train[u])) for u in range(n_users)]
    pool = np.asarray(sorted(test), np.int32)
    hist_p, hist_n = np.zeros((n_users, 3)), np.zeros((n_users, n_items))
    for step in range(5000):
        u, i, j = oracle.sample_triples(2, step, 24, n_items, oracle.csr_from_lists(tl), pool=pool,
                                        exclude=oracle.csr_from_lists(ex))
        assert sorted(u.tolist()) == pool.tolist()
        for uu, ii, jj in zip(u, i, j):
            hist_p[uu, tl[uu].index(ii)] += 1
            hist_n[uu, jj] += 1
    ps = []
    for u in pool:
        assert hist_n[u, ex[u]].sum() == 0
        ps.append(stats.chisquare(hist_p[u]).pvalue)
        ps.append(stats.chisquare(hist_n[u, np.setdiff1d(np.arange(n_items), ex[u])]).pvalue)
    assert min(ps) > P_MIN / len(ps) and stats.kstest(ps, "uniform").pvalue > P_MIN


def _two_sample(ref, got, bins):
    """chi-square homogeneity test of two samples over `bins` classes"""
    table = np.stack([np.bincount(ref, minlength=bins), np.bincount(got, minlength=bins)])
    table = table[:, table.sum(0) > 0]
    return stats.chi2_contingency(table)[1]


@pytest.mark.parametrize("which", ["mf", "lgcn", "lgcn_test"])
def test_marginals_match_the_reference_sampler_stream(which):
    """Same marginals as the reference's sampler on Addressa (the dataset the reference ships): popularity class of the
    positive item, position of the positive in its list, id class of the negative, size class of the drawn user's list."""
    if which == "mf":
        from macr_amd.data import MFData
        data = MFData(dataset_args("addressa"))
        lists = [sorted(data.train_user_list[u]) for u in range(data.n_users)]
        draw_ref = data.sample
        pool, excl = None, None
        n_users, n_items, B = data.n_users, data.n_items, 1024
        src = lists
    else:
        from macr_amd.data import LGCNData
        a = dataset_args("addressa")
        dg = LGCNData(path=a.data_path + a.dataset, batch_size=1024, args=a)
        n_users, n_items, B = dg.n_users, dg.n_items, 1024
        train = [sorted(dg.train_items.get(u, [])) for u in range(n_users)]
        if which == "lgcn":
            draw_ref, src, pool, excl = dg.sample, train, np.asarray(dg.exist_users, np.int32), None
        else:
            test = [sorted(dg.test_set.get(u, [])) for u in range(n_users)]
            draw_ref, src = dg.sample_test, test
            pool = np.asarray(list(dg.test_set.keys()), np.int32)
            excl = [sorted(set(test[u]) | set(train[u])) for u in range(n_users)]
            B = min(B, len(pool))
            dg.batch_size = B
    random.seed(99); np.random.seed(99)
    n_batches = 24
    ref = np.concatenate([np.asarray(draw_ref(), np.int64).reshape(3, -1) for _ in range(n_batches)], axis=1)
    csr = oracle.csr_from_lists(src)
    ex = None if excl is None else oracle.csr_from_lists(excl)
    got = np.concatenate([oracle.sample_triples(2024, s, B, n_items, csr, pool=pool, exclude=ex) for s in range(n_batches)],
                         axis=1).astype(np.int64)
    pop = np.bincount(np.concatenate([np.asarray(l, np.int64) for l in src if len(l)]), minlength=n_items)
    pop_class = np.searchsorted(np.quantile(pop[pop > 0], np.linspace(0, 1, 9)[1:-1]), pop)        # 8 popularity classes
    lens = np.asarray([len(l) for l in src])
    len_class = np.searchsorted(np.quantile(lens[lens > 0], np.linspace(0, 1, 7)[1:-1]), lens)      # 6 list-size classes

    def pos_slot(s):                                           # position of the positive in its list, in eighths
        u, i = s[0], s[1]
        return np.asarray([min(7, 8 * src[a].index(b) // len(src[a])) if len(src[a]) else 0 for a, b in zip(u, i)])

    assert _two_sample(pop_class[ref[1]], pop_class[got[1]], 8) > P_MIN
    assert _two_sample(pos_slot(ref), pos_slot(got), 8) > P_MIN
    assert _two_sample(ref[2] * 16 // n_items, got[2] * 16 // n_items, 16) > P_MIN
    assert _two_sample(len_class[ref[0]], len_class[got[0]], 6) > P_MIN
