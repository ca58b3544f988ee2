"""Parity of the HIP path against the CPU oracle, through the C ABI (-m gpu).

Index work (top-K ids, merges) must be bit-exact; floating-point model arithmetic
uses the tolerances of BASELINE.json's north star (per-step loss 1e-5 relative).
"""
import os

import numpy as np
import pytest
import torch

import oracle

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpu = pytest.mark.gpu
pytestmark = gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from macr_amd import ops as _ops
    return _ops


@pytest.fixture(params=["f32", "bf16", "f16"])
def eval_filter(request, ops, monkeypatch):
    """run the test once per candidate filter of the listing pass (include/macr_hip.h MACR_EVAL_FILTER_*): the ranking
    must be the fp32 ranking bit for bit either way"""
    monkeypatch.setenv("MACR_EVAL_FILTER", request.param)       # what an Evaluator created by the test picks up
    ops.set_eval_filter(request.param)
    yield request.param
    ops.set_eval_filter("env")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def make_problem(seed, n_users, n_items, d, B, scale=0.3, dup=True):
    rs = np.random.RandomState(seed)
    P = (rs.standard_normal((n_users, d)) * scale).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * scale).astype(np.float32)
    w = (rs.standard_normal(d) * 0.3).astype(np.float32)
    wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    u = rs.choice(n_users, B, replace=B > n_users).astype(np.int32)
    i = rs.randint(0, n_items, B).astype(np.int32)
    j = rs.randint(0, n_items, B).astype(np.int32)
    if dup:
        i[: B // 3] = 0                      # hot item (Addressa item 0 sits in half the train lists)
    return P, Q, w, wu, u, i, j


# ----------------------------------------------------------------------------- training step
@pytest.mark.parametrize("kind", [oracle.LOSS_NORMALBCE, oracle.LOSS_RUBIBCEBOTH])
@pytest.mark.parametrize("B,d,n_users,n_items", [(96, 64, 300, 50), (257, 64, 300, 50), (1024, 64, 13485, 744),
                                                 (64, 32, 100, 40), (128, 128, 500, 300), (64, 256, 100, 40),
                                                 (2048, 64, 3000, 900)])
def test_mf_train_step_matches_oracle(ops, kind, B, d, n_users, n_items):
    P, Q, w, wu, u, i, j = make_problem(B + d, n_users, n_items, d, B)
    alpha, beta, decay, lr, bs = 1e-2, 1e-3, 1e-5, 1e-3, 1024
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    hyper = ops.make_hyper(lr, decay, alpha, beta, bs)
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, B)
    rs = np.random.RandomState(9)
    for t in range(3):
        if t:
            u = rs.choice(n_users, B, replace=B > n_users).astype(np.int32)
            i = rs.randint(0, n_items, B).astype(np.int32)
            j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
        got = state.step(kind, dev(u), dev(i), dev(j)).cpu().numpy()
        # per-step loss within 1e-5 relative (BASELINE.json north star)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
        if t == 0:
            # after the first step m = (1-beta1)*g: the dense, de-duplicated gradient itself
            for name, gm, om in (("P", state.mP, st.m[0]), ("Q", state.mQ, st.m[1])):
                g_hip, g_orc = gm.cpu().numpy() / 0.1, om / 0.1
                np.testing.assert_allclose(g_hip, g_orc, rtol=2e-4, atol=1e-6 * np.abs(g_orc).max(), err_msg=name)
            if kind == oracle.LOSS_RUBIBCEBOTH:
                np.testing.assert_allclose(state.mw.cpu().numpy(), st.m[2], rtol=2e-4, atol=1e-6 * np.abs(st.m[2]).max())
                np.testing.assert_allclose(state.mwu.cpu().numpy(), st.m[3], rtol=2e-4, atol=1e-6 * np.abs(st.m[3]).max())
        # tables: every row moves every step (dense Adam); a step is at most ~lr.  The Adam update lr_t*m/(sqrt(v)+eps)
        # amplifies a relative gradient difference only where |g| ~ eps, by at most lr/4 per unit of relative error:
        # 0.2 % of a step bounds it with room for summation-order noise in cancelling gradient sums.
        for name, mine, theirs in (("P", state.P, Po), ("Q", state.Q, Qo), ("w", state.w, wo), ("wu", state.wu, wuo)):
            np.testing.assert_allclose(mine.cpu().numpy(), theirs, rtol=0, atol=2e-3 * lr * (t + 1), err_msg=name)
    # Adam slots after three steps: m and v of every table against the oracle
    for name, mine, theirs in (("mP", state.mP, st.m[0]), ("mQ", state.mQ, st.m[1]), ("vP", state.vP, st.v[0]),
                               ("vQ", state.vQ, st.v[1])):
        np.testing.assert_allclose(mine.cpu().numpy(), theirs, rtol=1e-4, atol=2e-6 * np.abs(theirs).max(), err_msg=name)
    # scratch invariants of the ABI: gradient scratch and touched flags are zero again
    assert float(state.gP.abs().max()) == 0.0 and float(state.gQ.abs().max()) == 0.0
    assert int(state.tP.sum()) == 0 and int(state.tQ.sum()) == 0
    np.testing.assert_allclose(state.adam_pow.cpu().numpy(), st.power, rtol=1e-6)
    if kind == oracle.LOSS_NORMALBCE:        # branch vectors get no gradient -> bitwise untouched
        assert np.array_equal(state.w.cpu().numpy(), w) and np.array_equal(state.wu.cpu().numpy(), wu)


@pytest.mark.parametrize("kind", [oracle.LOSS_NORMALBCE, oracle.LOSS_RUBIBCEBOTH])
@pytest.mark.parametrize("defer", [False, True])
def test_mf_twenty_step_trajectory_stays_on_the_oracle(ops, kind, defer):
    """Teacher-forced drift: 20 steps on the same batches, HIP (its Adam pass uses v_sqrt / v_rcp + one Newton step, <= 1
    ulp; -DMACR_ADAM_IEEE restores sqrtf and division -- this test passes either way) against the oracle (IEEE).  The
    difference must not build up over a trajectory: tables within 0.2 % of ONE step per step taken, losses 1e-5."""
    B, d, n_users, n_items = 512, 64, 2000, 600
    P, Q, w, wu, u, i, j = make_problem(77, n_users, n_items, d, B)
    alpha, beta, decay, lr, bs = 1e-2, 1e-3, 1e-5, 1e-3, 1024
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), ops.make_hyper(lr, decay, alpha, beta, bs), B)
    rs = np.random.RandomState(21)
    steps = 20
    for t in range(steps):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.4, B) - 1).clip(0, n_items - 1).astype(np.int32)
        j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
        got = state.step(kind, dev(u), dev(i), dev(j), defer=defer).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0, err_msg="step %d" % t)
    if defer:
        state.flush()
    for name, mine, theirs in (("P", state.P, Po), ("Q", state.Q, Qo), ("w", state.w, wo), ("wu", state.wu, wuo)):
        diff = np.abs(mine.cpu().numpy() - theirs).max()
        assert diff <= 2e-3 * lr * steps, (name, diff)
        # and the typical row is far inside the bound: nothing drifts systematically
        assert np.abs(mine.cpu().numpy() - theirs).mean() <= 1e-4 * lr * steps, name
    for name, mine, theirs in (("mP", state.mP, st.m[0]), ("mQ", state.mQ, st.m[1]), ("vP", state.vP, st.v[0]),
                               ("vQ", state.vQ, st.v[1])):
        np.testing.assert_allclose(mine.cpu().numpy(), theirs, rtol=2e-4, atol=2e-6 * np.abs(theirs).max(), err_msg=name)


def test_adam_pass_known_answers_from_1e_12_to_1e_1(ops):
    """tf.train.AdamOptimizer as its documentation states it (macr_mf/model.py:74,:95 use it with the defaults):
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;
        theta -= lr_t * m / (sqrt(v) + epsilon)          -- epsilon OUTSIDE the root, lr_t carrying both bias corrections.
    The trajectory tests bound the tables by a fraction of a step, which a misplaced epsilon survives on rows whose gradients
    are far above it.  Here the pass gets gradients of every magnitude from 1e-12 to 1e-1 on tables that start at zero (the
    new value IS the update) and must return the float64 evaluation of that formula to 1e-5 -- around |g| ~ 3e-7, where
    sqrt(v) ~ epsilon, the variants (epsilon inside the root, epsilon-hat of the paper's form, no bias correction) are off by
    factors, which the test checks of itself."""
    B, d = 256, 64
    # (the hyper-parameters as the fp32 graph holds them: 1 - 0.999f is 1.3e-5 away from 0.001)
    lr, b1, b2, eps = (float(np.float32(x)) for x in (1e-3, 0.9, 0.999, 1e-8))
    rs = np.random.RandomState(5)
    P = (rs.standard_normal((B, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((B, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    # every row of both tables is in the batch: the pending pass owns a gradient row for each
    u, i, j = (rs.permutation(B).astype(np.int32) for _ in range(3))
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), ops.make_hyper(lr, 1e-5, 1e-2, 1e-3, 1024), B, lazy_period=1)
    state.step(oracle.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j), defer=True)
    assert int(state.tP.min()) != 0 and int(state.tQ.min()) != 0
    g = {}
    for name, tab, grad in (("P", state.P, state.gP), ("Q", state.Q, state.gQ)):
        mag = 10.0 ** rs.uniform(-12, -1, size=(B, d))
        g[name] = (mag * rs.choice([-1.0, 1.0], size=(B, d))).astype(np.float32)
        grad.copy_(dev(g[name]))                      # the pending pass's gradient rows, replaced
        tab.zero_()
    state.flush()
    lr_t = lr * np.sqrt(1.0 - b2) / (1.0 - b1)
    discriminated = 0
    for name, tab, m_, v_ in (("P", state.P, state.mP, state.vP), ("Q", state.Q, state.mQ, state.vQ)):
        g64 = g[name].astype(np.float64)
        m, v = (1.0 - b1) * g64, (1.0 - b2) * g64 * g64
        want = -lr_t * m / (np.sqrt(v) + eps)
        got = tab.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0, err_msg=name)
        np.testing.assert_allclose(m_.cpu().numpy(), m, rtol=1e-6, atol=0, err_msg="m" + name)
        np.testing.assert_allclose(v_.cpu().numpy(), v, rtol=1e-6, atol=1e-44, err_msg="v" + name)
        # what the bound above excludes: each of these differs from the result by more than 10 % somewhere
        for wrong in (-lr_t * m / np.sqrt(v + eps),                                                   # epsilon inside the root
                      -lr * (m / (1.0 - b1)) / (np.sqrt(v / (1.0 - b2)) + eps),                       # Kingma & Ba's epsilon (not epsilon-hat)
                      -lr * m / (np.sqrt(v) + eps)):                                                  # no bias correction
            discriminated += int(np.max(np.abs(wrong - got) / np.abs(got)) > 0.1)
    assert discriminated == 6
    assert float(state.gP.abs().max()) == 0.0 and float(state.gQ.abs().max()) == 0.0 and int(state.tP.sum()) == 0


def test_mf_trajectory_with_ieee_adam():
    """The same 20-step trajectory test against the build whose Adam pass keeps IEEE sqrtf and division
    (macr_amd/csrc/libmacr_hip_ieee.so, -DMACR_ADAM_IEEE) -- in a process of its own: a process binds one library."""
    import subprocess
    import sys
    lib = os.path.join(REPO, "macr_amd", "csrc", "libmacr_hip_ieee.so")
    assert os.path.exists(lib), "build it: make -C macr_amd/csrc"
    env = dict(os.environ, MACR_HIP_LIB=lib)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                          "twenty_step_trajectory"], env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0 and "4 passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("scale", [0.45, 0.6, 1.2])
@pytest.mark.parametrize("B", [300, 4096])
def test_mf_rubibceboth_large_logits(ops, scale, B):
    """The (B,B) kernel takes its 4-transcendental form only for column tiles with -20 <= p, -20 <= n <= 3 and the
    operation-by-operation form otherwise.  Larger embeddings put tiles on both sides of that test (scale 0.6: dot
    products of std ~3) or almost all on the exact side (1.2: std ~12, saturated sigmoids); loss and gradients must
    follow the oracle (which mirrors the reference's 1-sigmoid subtraction) everywhere."""
    d, n_users, n_items = 64, 5000, 900
    P, Q, w, wu, u, i, j = make_problem(int(scale * 100) + B, n_users, n_items, d, B, scale=scale)
    alpha, beta, decay, lr, bs = 1e-2, 1e-3, 1e-5, 1e-3, 1024
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    n = np.einsum("bd,bd->b", P[u], Q[j])
    assert (scale < 1.0) or (np.abs(n) > 3).mean() > 0.5             # the exact side is really exercised
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), ops.make_hyper(lr, decay, alpha, beta, bs), B)
    want = oracle.mf_train_step(oracle.LOSS_RUBIBCEBOTH, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
    got = state.step(oracle.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
    for name, gm, om in (("P", state.mP, st.m[0]), ("Q", state.mQ, st.m[1])):
        g_hip, g_orc = gm.cpu().numpy() / 0.1, om / 0.1
        np.testing.assert_allclose(g_hip, g_orc, rtol=5e-4, atol=2e-6 * np.abs(g_orc).max(), err_msg=name)
    np.testing.assert_allclose(state.mw.cpu().numpy(), st.m[2], rtol=5e-4, atol=2e-6 * np.abs(st.m[2]).max())
    np.testing.assert_allclose(state.mwu.cpu().numpy(), st.m[3], rtol=5e-4, atol=2e-6 * np.abs(st.m[3]).max())


@pytest.mark.parametrize("B,d", [(512, 32), (1024, 64), (4096, 64), (4096, 128)])
def test_mf_rubibceboth_a_few_columns_outside_the_window(ops, B, d):
    """Full batches (B % 256 == 0): columns outside the window of the (B,B) kernel's 4-transcendental form are taken out of
    the rotation and evaluated by the exact form in the launch's neutral blocks (k_bxb).  A tame batch with a handful of
    extreme columns -- negatives scoring +25 and -70, a positive scoring -40, ten of them inside one 64-column tile (more than
    a tile neutralises) -- must follow the oracle like any other: loss 1e-5, gradients of every row."""
    n_users, n_items = max(6000, 2 * B), max(5000, 3 * B)
    rs = np.random.RandomState(B + d)
    P = (rs.standard_normal((n_users, d)) * 0.25).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.25).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    u = rs.choice(n_users, B, replace=False).astype(np.int32)
    ij = rs.choice(n_items, 2 * B, replace=False).astype(np.int32)
    i, j = ij[:B].copy(), ij[B:].copy()
    def aim(row, target, at):                       # make row . P[u[at]] = target
        pu = P[u[at]]
        row += (target - float(row @ pu)) * pu / float(pu @ pu)
    for at, tgt in ((3, 25.0), (70, -70.0), (200, 9.0), (B - 1, 30.0)):
        aim(Q[j[at]], tgt, at)
    aim(Q[i[130]], -40.0, 130)
    for at in range(256, 266):                      # ten extreme columns in ONE tile
        aim(Q[j[at]], 12.0 + at % 5, at)
    alpha, beta, decay, lr, bs = 1e-2, 1e-3, 1e-5, 1e-3, 1024
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), ops.make_hyper(lr, decay, alpha, beta, bs), B)
    want = oracle.mf_train_step(oracle.LOSS_RUBIBCEBOTH, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
    got = state.step(oracle.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
    for name, gm, om in (("P", state.mP, st.m[0]), ("Q", state.mQ, st.m[1])):
        g_hip, g_orc = gm.cpu().numpy() / 0.1, om / 0.1
        np.testing.assert_allclose(g_hip, g_orc, rtol=5e-4, atol=2e-6 * np.abs(g_orc).max(), err_msg=name)
    np.testing.assert_allclose(state.mw.cpu().numpy(), st.m[2], rtol=5e-4, atol=2e-6 * np.abs(st.m[2]).max())
    np.testing.assert_allclose(state.mwu.cpu().numpy(), st.m[3], rtol=5e-4, atol=2e-6 * np.abs(st.m[3]).max())


@pytest.mark.parametrize("B,d,n_users,n_items,sort", [(96, 64, 300, 50, False), (257, 64, 300, 50, True),
                                                      (1024, 64, 13485, 744, True), (64, 32, 100, 40, True),
                                                      (128, 128, 500, 300, False), (64, 256, 100, 40, True),
                                                      (4096, 64, 3000, 900, True)])
def test_mf_deferred_adam_equals_complete_steps(ops, B, d, n_users, n_items, sort):
    """MACR_STEP_DEFER / MACR_STEP_PENDING: the dense Adam pass of step t runs under the (B,B) kernel of step
    t+1 (rows of batch t+1 first).  Losses per step follow the oracle to 1e-5; after flush the whole state equals
    the state of complete (flags=0) steps -- same arithmetic per row, only float-atomic order differs."""
    P, Q, w, wu, u, i, j = make_problem(B + d + 1, n_users, n_items, d, B)
    alpha, beta, decay, lr, bs = 1e-2, 1e-3, 1e-5, 1e-3, 1024
    kind = oracle.LOSS_RUBIBCEBOTH
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    hyper = ops.make_hyper(lr, decay, alpha, beta, bs)
    lazy = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, B)
    eager = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, B)
    rs = np.random.RandomState(19)
    for t in range(5):
        if t:
            u = rs.choice(n_users, B, replace=B > n_users).astype(np.int32)
            i = rs.randint(0, n_items, B).astype(np.int32)
            j = rs.randint(0, n_items, B).astype(np.int32)
            if t == 2:
                i[: B // 2] = 3                         # a different hot item
        if sort:
            o = np.argsort(i, kind="stable")
            u, i, j = u[o], i[o], j[o]
        want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
        got = lazy.step(kind, dev(u), dev(i), dev(j), defer=True).cpu().numpy()
        ref = eager.step(kind, dev(u), dev(i), dev(j)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=0)
        assert lazy.pending_B == B
    lazy.flush()
    assert lazy.pending_B == 0
    for name in ("P", "Q", "w", "wu", "mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu"):
        a, b = getattr(lazy, name).cpu().numpy(), getattr(eager, name).cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-7 + 1e-5 * np.abs(b).max(), err_msg=name)
    np.testing.assert_allclose(lazy.P.cpu().numpy(), Po, rtol=0, atol=0.02 * lr * 5)
    np.testing.assert_allclose(lazy.Q.cpu().numpy(), Qo, rtol=0, atol=0.02 * lr * 5)
    assert float(lazy.gP.abs().max()) == 0.0 and float(lazy.gQ.abs().max()) == 0.0
    assert int(lazy.tP.sum()) == 0 and int(lazy.tQ.sum()) == 0
    np.testing.assert_allclose(lazy.adam_pow.cpu().numpy(), st.power, rtol=1e-6)
    np.testing.assert_array_equal(lazy.adam_pow.cpu().numpy(), eager.adam_pow.cpu().numpy())


@pytest.mark.parametrize("seed", range(12))
def test_mf_deferred_random_shapes(ops, seed):
    """Random batch sizes (around the 256-row (B,B) tile and the 16-triple backward chunk), tables, widths and
    duplicate patterns: deferred steps and complete steps agree step by step and after the flush."""
    rs = np.random.RandomState(500 + seed)
    B = int(rs.choice([1, 15, 17, 255, 256, 257, 1000, 1024, 3000, 4096, 5000]))
    d = int(rs.choice([32, 64, 64, 128, 256]))
    n_users = int(rs.choice([B + 1, 2 * B + 7, 5000])) if B > 1 else 9
    n_items = int(rs.choice([3, 50, 700, 4000]))
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w = (rs.standard_normal(d) * 0.3).astype(np.float32)
    wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    hyper = ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 1024)
    kind = oracle.LOSS_RUBIBCEBOTH
    lazy = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, B)
    eager = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, B)
    for t in range(4):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.3, B) % n_items).astype(np.int32)            # heavy duplicates among the positives
        j = rs.randint(0, n_items, B).astype(np.int32)
        if rs.rand() < 0.5:
            o = np.argsort(i, kind="stable"); u, i, j = u[o], i[o], j[o]
        a = lazy.step(kind, dev(u), dev(i), dev(j), defer=True).cpu().numpy()
        b = eager.step(kind, dev(u), dev(i), dev(j)).cpu().numpy()
        assert np.isfinite(a).all()
        np.testing.assert_allclose(a, b, rtol=5e-6, atol=0, err_msg=str((B, d, n_users, n_items, t)))
    lazy.flush()
    for name in ("P", "Q", "w", "wu", "mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu"):
        x, y = getattr(lazy, name).cpu().numpy(), getattr(eager, name).cpu().numpy()
        np.testing.assert_allclose(x, y, rtol=3e-4, atol=1e-7 + 2e-5 * np.abs(y).max(), err_msg=name)
    assert float(lazy.gP.abs().max()) == 0.0 and float(lazy.gQ.abs().max()) == 0.0
    assert int(lazy.tP.sum()) == 0 and int(lazy.tQ.sum()) == 0


@pytest.mark.parametrize("kind,B,d,n_users,n_items", [
    (oracle.LOSS_NORMALBCE, 20000, 64, 30000, 5000), (oracle.LOSS_NORMALBCE, 9001, 32, 500, 60),
    (oracle.LOSS_NORMALBCE, 8200, 128, 9000, 700), (oracle.LOSS_RUBIBCEBOTH, 9000, 64, 12000, 3000),
    (oracle.LOSS_RUBIBCEBOTH, 8448, 256, 400, 90)])
def test_mf_large_batch_staged_path_matches_oracle(ops, kind, B, d, n_users, n_items):
    """B > 8192 takes the staging + sorted-references + segment-reduce path (train_kernels.hip, k_seg_reduce): hot
    rows (a third of the positives on one item: runs far longer than a 16-reference chunk), users drawn with
    replacement when B > n_users, three steps so that the gradient scratch invariant is exercised."""
    P, Q, w, wu, u, i, j = make_problem(B + d, n_users, n_items, d, B)
    alpha, beta, decay, lr, bs = 1e-2, 1e-3, 1e-5, 1e-3, 1024
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), ops.make_hyper(lr, decay, alpha, beta, bs), B)
    rs = np.random.RandomState(3)
    for t in range(3):
        if t:
            u = rs.choice(n_users, B, replace=B > n_users).astype(np.int32)
            i = (rs.zipf(1.2, B) % n_items).astype(np.int32)
            j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
        got = state.step(kind, dev(u), dev(i), dev(j)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
        if t == 0:
            for name, gm, om in (("P", state.mP, st.m[0]), ("Q", state.mQ, st.m[1])):
                np.testing.assert_allclose(gm.cpu().numpy() / 0.1, om / 0.1, rtol=2e-4, atol=2e-6 * np.abs(om / 0.1).max(),
                                           err_msg=name)
    for name, mine, theirs in (("P", state.P, Po), ("Q", state.Q, Qo), ("mP", state.mP, st.m[0]), ("mQ", state.mQ, st.m[1]),
                               ("vP", state.vP, st.v[0]), ("vQ", state.vQ, st.v[1])):
        np.testing.assert_allclose(mine.cpu().numpy(), theirs, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(theirs).max(), err_msg=name)
    assert float(state.gP.abs().max()) == 0.0 and float(state.gQ.abs().max()) == 0.0
    assert int(state.tP.sum()) == 0 and int(state.tQ.sum()) == 0


def test_mf_large_batch_staged_path_is_deterministic(ops):
    """Every row has one owner that sums in sorted (= batch) order; only rows cut into several chunks add atomically.
    With no row referenced more than a chunk's worth of times two runs give the same bits."""
    B, d, n_users, n_items = 16384, 64, 40000, 30000
    P, Q, w, wu, u, i, j = make_problem(77, n_users, n_items, d, B, dup=False)
    outs = []
    for _ in range(2):
        state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 1024), B)
        for _t in range(2):
            state.step(oracle.LOSS_NORMALBCE, dev(u), dev(i), dev(j))
        outs.append((state.P.cpu().numpy(), state.Q.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("unfused", [False, True])
@pytest.mark.parametrize("kind_name,B,d", [("LOSS_RUBIBCEBOTH", 777, 64), ("LOSS_RUBIBCE", 256, 32), ("LOSS_RUBIBCEBOTH", 2048, 128),
                                           ("LOSS_NORMALBCE", 1000, 64), ("LOSS_NORMALBCE", 300, 256)])
def test_row_shard_entry_points_world1(ops, kind_name, B, d, unfused, monkeypatch):
    """The macr_shard_* device entry points (row-sharded training) on one rank owning everything: gather, forward,
    (B,B) row blocks, backward into the staging buffer, sorted segment reduce + Adam -- against the oracle step."""
    from macr_amd import sharded_train
    if unfused:                                   # the segment reduce writes every gradient row (before round 4: always)
        monkeypatch.setenv("MACR_SEG_UNFUSED", "1")
    else:                                         # the Adam pass sums the staged rows itself (adam_block INDEXED)
        monkeypatch.delenv("MACR_SEG_UNFUSED", raising=False)
    kind = getattr(ops, kind_name)
    n_users, n_items = 900, 350
    P, Q, w, wu, u, i, j = make_problem(B + d, n_users, n_items, d, B)
    lr, decay, alpha, beta, bs = 1e-3, 1e-5, 1e-2, 1e-3, 512
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    hyper = ops.make_hyper(lr, decay, alpha, beta, bs)
    model = sharded_train.RowShardedMF(dev(P), dev(Q), dev(w), dev(wu),
                                       sharded_train.HipBackend(kind, d, hyper, torch.device("cuda")), rank=0, world=1)
    rs = np.random.RandomState(4)
    for t in range(3):
        if t:
            u = rs.choice(n_users, B, replace=B > n_users).astype(np.int32)
            i = (rs.zipf(1.3, B) % n_items).astype(np.int32)
            j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.mf_train_step(getattr(oracle, kind_name), u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
        ops.timing_begin()
        got = model.step(dev(u), dev(i), dev(j)).cpu().numpy()
        names = {n for n, _ in ops.timing_end(64)}
        assert ("adam_indexed" in names) == (not unfused) and ("seg_reduce" in names) == unfused, names
        np.testing.assert_allclose(got, want, rtol=1e-5)
        if t == 0:
            np.testing.assert_allclose(model.mP.cpu().numpy() / 0.1, st.m[0] / 0.1, rtol=2e-4, atol=2e-6 * np.abs(st.m[0] / 0.1).max())
            np.testing.assert_allclose(model.mQ.cpu().numpy() / 0.1, st.m[1] / 0.1, rtol=2e-4, atol=2e-6 * np.abs(st.m[1] / 0.1).max())
    for name, mine, theirs in (("P", model.P, Po), ("Q", model.Q, Qo), ("w", model.w, wo), ("wu", model.wu, wuo)):
        np.testing.assert_allclose(mine.cpu().numpy(), theirs, rtol=0, atol=2e-3 * lr * 3, err_msg=name)
    assert float(model.gP.abs().max()) == 0.0 and float(model.gQ.abs().max()) == 0.0
    assert int(model.tP.sum()) == 0 and int(model.tQ.sum()) == 0


@pytest.mark.parametrize("world", [2, 3])
def test_evaluator_ranks_interleaved_item_shards(ops, world, eval_filter):
    """The item shard of a row-sharded model is STRIDED (item g on rank g % W, sharded_train.Owned): every rank ranks its
    rows in local ids against the owned part of the train lists, ids are mapped back, the shards merge to the unsharded
    ranking bit for bit -- W ranks played one after the other on this GPU."""
    from macr_amd import sharded_train
    from macr_amd.evaluator import Evaluator
    rs = np.random.RandomState(world)
    n_users, N, d, K = 400, 3001, 64, 20
    P = (rs.standard_normal((n_users, d)) * 0.5).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.5 + rs.standard_normal((N, 1)) * 0.4).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    users = np.sort(rs.choice(n_users, 300, replace=False)).astype(np.int32)
    mask = random_mask(rs, len(users), N, 14, heavy=(0, 7))
    gt = [sorted(rs.choice(N, 5, replace=False).tolist()) for _ in users]
    Pq = dev(P[users])
    whole = Evaluator(mask, gt, N, torch.device("cuda"))
    wv, wi, wc = whole.rank(ops.SCORE_RUBI_BOTH, Pq, None, dev(Q), K, dev(w), dev(wu), 40.0)
    parts_v, parts_i = [], []
    for r in range(world):
        own = sharded_train.Owned(N, r, world)
        ev = Evaluator(mask, gt, N, torch.device("cuda"))
        ev.set_local_items(own)
        for rep in range(2):                                     # second call: seeded thresholds (local ids)
            v, ix = ev.rank_local(ops.SCORE_RUBI_BOTH, Pq, None, dev(np.ascontiguousarray(Q[r::world])), K, dev(w), dev(wu), 40.0)
        lv, li, _ = ops.topk_merge(v, ix)
        assert bool(((li < 0) | (li % world == r)).all())
        parts_v.append(lv); parts_i.append(li)
    mv, mi, mc = ops.topk_merge(torch.stack(parts_v), torch.stack(parts_i))
    assert torch.equal(mi, wi) and torch.equal(mv, wv) and torch.equal(mc, wc)


def test_mf_deferred_mode_flushes_on_batch_size_change(ops):
    P, Q, w, wu, u, i, j = make_problem(5, 300, 50, 64, 200)
    hyper = ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 1024)
    kind = oracle.LOSS_RUBIBCEBOTH
    lazy = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, 64)
    eager = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, 64)
    for B in (64, 64, 200, 200, 96, 200):             # growth re-allocates the workspace; shrink changes its carving
        a, b = lazy.step(kind, dev(u[:B]), dev(i[:B]), dev(j[:B]), defer=True), eager.step(kind, dev(u[:B]), dev(i[:B]), dev(j[:B]))
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-6)
    # normalbce has no (B,B) kernel to hide the pass under: a pending pass is completed first, nothing is left pending
    lazy.step(oracle.LOSS_NORMALBCE, dev(u[:64]), dev(i[:64]), dev(j[:64]), defer=True)
    eager.step(oracle.LOSS_NORMALBCE, dev(u[:64]), dev(i[:64]), dev(j[:64]))
    assert lazy.pending_B == 0
    for name in ("P", "Q", "w", "wu", "mP", "vP", "mQ", "vQ"):
        a, b = getattr(lazy, name).cpu().numpy(), getattr(eager, name).cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-7 + 1e-5 * np.abs(b).max(), err_msg=name)


def test_untouched_rows_follow_dense_adam(ops):
    """TF-1.14 Adam on embedding tables moves EVERY row (SURVEY.md finding 5)."""
    P, Q, w, wu, u, i, j = make_problem(1, 400, 80, 64, 32, dup=False)
    hyper = ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 32)
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu), hyper, 32)
    state.step(oracle.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j))
    P1 = state.P.cpu().numpy().copy()
    untouched = np.setdiff1d(np.arange(400), u)
    assert np.array_equal(P1[untouched], P[untouched])       # m=v=0, g=0: first step leaves them
    state.step(oracle.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j))
    rows_first = u[:4]
    uu = np.setdiff1d(np.arange(400), rows_first)[:32].astype(np.int32)   # rows_first untouched in step 3
    state.step(oracle.LOSS_RUBIBCEBOTH, dev(uu), dev(i), dev(j))
    P3 = state.P.cpu().numpy()
    P2_rows = None
    # rows touched earlier keep moving with zero gradient because m != 0
    assert np.all(np.abs(P3[rows_first] - P1[rows_first]).max(axis=1) > 0)


# ----------------------------------------------------------------------------- LightGCN
def toy_graph(n_users, n_items, seed, density=0.1):
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    R = (rs.rand(n_users, n_items) < density).astype(np.float32)
    R[0, :] = 0
    R[:, 1] = 1                                   # a hub item: one very long row
    R[0, 1] = 0
    A = sp.bmat([[None, sp.csr_matrix(R)], [sp.csr_matrix(R.T), None]]).tocsr()
    deg = np.asarray(A.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5).astype(np.float32)
    dinv[np.isinf(dinv)] = 0
    A_hat = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32)
    A_hat.sort_indices()
    return A_hat


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("L", [0, 1, 2, 3])
def test_lgcn_propagate_matches_oracle(ops, d, L):
    n_users, n_items = 700, 333
    A = toy_graph(n_users, n_items, 3)
    rs = np.random.RandomState(d + L)
    E0 = rs.standard_normal((n_users + n_items, d)).astype(np.float32)
    want = oracle.lgcn_propagate(A.indptr, A.indices, A.data, E0, L)
    adj = ops.CSR.from_scipy(A, "cuda")                    # with the row-split plan (hub item > 512 neighbours)
    assert adj.plan_host is not None and adj.plan_host[2] >= 1          # header: n_split >= 1
    got = ops.lgcn_propagate(adj, dev(E0), L).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    plain = ops.CSR(adj.ptr, adj.idx, adj.val)             # no plan: one wavefront per row
    got2 = ops.lgcn_propagate(plain, dev(E0), L).cpu().numpy()
    np.testing.assert_allclose(got2, want, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("octants", ["1", "0"])
@pytest.mark.parametrize("d", [64, 128, 256])
def test_lgcn_propagate_entry_stream_and_piece_orders(ops, d, octants, monkeypatch):
    """The optional entry-stream kernel of the dense layers (MACR_SPMM_STREAM=1 at plan time: chunks of a plan-owned
    (column, weight) stream, rows ending inside it) and both orders of the hub pieces (MACR_SPMM_OCTANTS: cut at the
    boundaries of eight source-row ranges and placed XCD by XCD, or in slot order) give the oracle's propagation; a graph
    with several hub rows, rows without neighbours and a row-count that is no multiple of anything."""
    import scipy.sparse as sp
    rs = np.random.RandomState(17 + d)
    n_users, n_items = 1900, 517
    R = (rs.rand(n_users, n_items) < 0.02).astype(np.float32)
    R[:, :5] = (rs.rand(n_users, 5) < 0.7)                # five hub items with > 1000 neighbours each
    R[:3, :] = 0; R[:, 40:47] = 0                         # users and items without neighbours
    A = sp.bmat([[None, sp.csr_matrix(R)], [sp.csr_matrix(R.T), None]]).tocsr()
    deg = np.asarray(A.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5).astype(np.float32)
    dinv[np.isinf(dinv)] = 0
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32)
    A.sort_indices()
    E0 = rs.standard_normal((n_users + n_items, d)).astype(np.float32)
    monkeypatch.setenv("MACR_SPMM_OCTANTS", octants)
    for stream in ("0", "1"):
        monkeypatch.setenv("MACR_SPMM_STREAM", stream)
        adj = ops.CSR.from_scipy(A, "cuda")
        assert adj.plan_host[2] >= 5                      # header: n_split
        assert (adj.plan_host[7] > 0) == (stream == "1")  # header: offset of the stream section
        for L in (1, 2, 3):
            want = oracle.lgcn_propagate(A.indptr, A.indices, A.data, E0, L)
            ops.timing_begin()
            got = ops.lgcn_propagate(adj, dev(E0), L).cpu().numpy()
            names = {n for n, _ in ops.timing_end(16)}
            assert ("spmm_stream" in names) == (stream == "1"), names
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("kind", [oracle.LOSS_NORMALBCE, oracle.LOSS_RUBIBCEBOTH])
@pytest.mark.parametrize("d,B", [(64, 256), (128, 100)])
def test_lgcn_train_step_matches_oracle(ops, kind, d, B):
    n_users, n_items, L = 500, 200, 2
    A = toy_graph(n_users, n_items, 5)
    P, Q, w, wu, u, i, j = make_problem(d, n_users, n_items, d, B)
    T = np.concatenate([P, Q]).astype(np.float32)
    alpha, beta, decay, lr, bs = 1e-2, 1e-3, 1e-4, 1e-3, B
    st = oracle.AdamState([T.shape, (d,), (d,)])
    To, wo, wuo = T.copy(), w.copy(), wu.copy()
    state = ops.LGCNState(dev(T), n_users, n_items, dev(w), dev(wu), ops.CSR.from_scipy(A, "cuda"), L,
                          ops.make_hyper(lr, decay, alpha, beta, bs), B)
    rs = np.random.RandomState(2)
    for t in range(3):
        if t:
            u = rs.choice(n_users, B, replace=False).astype(np.int32)
            i = rs.randint(0, n_items, B).astype(np.int32)
            j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.lgcn_train_step(kind, n_users, n_items, L, A.indptr, A.indices, A.data, u, i, j,
                                      To, wo, wuo, st, lr, decay, alpha, beta, bs)
        got = state.step(kind, dev(u), dev(i), dev(j)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
        if t == 0:
            g_hip, g_orc = state.mT.cpu().numpy() / 0.1, st.m[0] / 0.1
            np.testing.assert_allclose(g_hip, g_orc, rtol=5e-4, atol=2e-6 * np.abs(g_orc).max())
        np.testing.assert_allclose(state.T.cpu().numpy(), To, rtol=0, atol=0.02 * lr * (t + 1))


@pytest.mark.parametrize("kind", [oracle.LOSS_NORMALBCE, oracle.LOSS_RUBIBCEBOTH])
@pytest.mark.parametrize("d,B,dense", [(64, 256, False), (64, 300, True), (128, 100, False), (32, 64, False)])
def test_lgcn_train_step_on_a_row_normalised_adjacency(ops, kind, d, B, dense):
    """--adj_type norm / gcmc / mean are D^-1 A (LightGCN.py:667-678, utility/load_data.py:95-164): the forward propagation runs
    on A, the backward propagation on A^T (macr_lgcn_train_step_t; a plan of its own, hub rows on both sides) -- losses,
    first-step gradients and tables against the oracle, which takes A^T explicitly; the step with A in both places differs."""
    import scipy.sparse as sp
    n_users, n_items, L = 900, 500, 2                            # (items 0 and 1 are hub rows of A^T and of A: > 512 neighbours)
    A = toy_graph(n_users, n_items, 5)
    deg = np.asarray((A != 0).sum(1)).ravel()
    An = (sp.diags(np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0)) @ (A != 0).astype(np.float32)).tocsr().astype(np.float32)
    An.sort_indices()
    AT = An.T.tocsr().astype(np.float32)
    AT.sort_indices()
    assert abs(An - AT).max() > 1e-3
    P, Q, w, wu, u, i, j = make_problem(d + 1, n_users, n_items, d, B)
    T = np.concatenate([P, Q]).astype(np.float32)
    alpha, beta, decay, lr = 1e-2, 1e-3, 1e-4, 1e-3
    st = oracle.AdamState([T.shape, (d,), (d,)])
    To, wo, wuo = T.copy(), w.copy(), wu.copy()
    hyper = ops.make_hyper(lr, decay, alpha, beta, B)
    state = ops.LGCNState(dev(T), n_users, n_items, dev(w), dev(wu), ops.CSR.from_scipy(An, "cuda"), L, hyper, B,
                          adj_t=ops.CSR.from_scipy(AT, "cuda"))
    wrong = ops.LGCNState(dev(T), n_users, n_items, dev(w), dev(wu), ops.CSR.from_scipy(An, "cuda"), L, hyper, B)
    rs = np.random.RandomState(4)
    for t in range(3):
        if t:
            u = rs.choice(n_users, B, replace=False).astype(np.int32)
            i = (rs.zipf(1.3, B) % n_items).astype(np.int32)
            j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.lgcn_train_step(kind, n_users, n_items, L, An.indptr, An.indices, An.data, u, i, j, To, wo, wuo, st, lr, decay,
                                      alpha, beta, B, transposed=(AT.indptr, AT.indices, AT.data))
        got = state.step(kind, dev(u), dev(i), dev(j), dense_layers=dense).cpu().numpy()
        wrong.step(kind, dev(u), dev(i), dev(j), dense_layers=dense)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
        if t == 0:
            g_hip, g_orc = state.mT.cpu().numpy() / 0.1, st.m[0] / 0.1
            np.testing.assert_allclose(g_hip, g_orc, rtol=5e-4, atol=2e-6 * np.abs(g_orc).max())
            g_wrong = wrong.mT.cpu().numpy() / 0.1
            assert np.abs(g_wrong - g_orc).max() > 1e-2 * np.abs(g_orc).max()       # (A for A^T: visibly another gradient)
        np.testing.assert_allclose(state.T.cpu().numpy(), To, rtol=0, atol=0.02 * lr * (t + 1))


@pytest.mark.parametrize("kind", [oracle.LOSS_NORMALBCE, oracle.LOSS_RUBIBCEBOTH])
def test_lgcn_loss_only_pass(ops, kind):
    """MACR_STEP_LOSS_ONLY (the reference's "test loss" pass, LightGCN.py:799-819): the losses of a batch, equal to
    the losses the oracle's next step reports for that batch, with the model, the slots and the step count untouched."""
    n_users, n_items, L, d, B = 500, 200, 2, 64, 300
    A = toy_graph(n_users, n_items, 5)
    P, Q, w, wu, u, i, j = make_problem(3, n_users, n_items, d, B)
    T = np.concatenate([P, Q]).astype(np.float32)
    alpha, beta, decay, lr = 1e-2, 1e-3, 1e-4, 1e-3
    state = ops.LGCNState(dev(T), n_users, n_items, dev(w), dev(wu), ops.CSR.from_scipy(A, "cuda"), L,
                          ops.make_hyper(lr, decay, alpha, beta, B), B)
    state.step(kind, dev(u), dev(i), dev(j))                       # one real step so that slots and powers are not trivial
    before = {n: getattr(state, n).clone() for n in ("T", "w", "wu", "mT", "vT", "mw", "vw", "mwu", "vwu", "adam_pow")}
    rs = np.random.RandomState(8)
    u2 = rs.choice(n_users, B, replace=False).astype(np.int32)
    i2 = (rs.zipf(1.3, B) % n_items).astype(np.int32)
    j2 = rs.randint(0, n_items, B).astype(np.int32)
    got = state.step(kind, dev(u2), dev(i2), dev(j2), loss_only=True).cpu().numpy()
    for n, t in before.items():
        assert torch.equal(getattr(state, n), t), n
    To, wo, wuo = state.T.cpu().numpy().copy(), state.w.cpu().numpy().copy(), state.wu.cpu().numpy().copy()
    st = oracle.AdamState([To.shape, (d,), (d,)])
    want = oracle.lgcn_train_step(kind, n_users, n_items, L, A.indptr, A.indices, A.data, u2, i2, j2, To, wo, wuo, st,
                                  lr, decay, alpha, beta, B)
    np.testing.assert_allclose(got, want, rtol=1e-5)
    real = state.step(kind, dev(u2), dev(i2), dev(j2)).cpu().numpy()  # and the real step still works afterwards
    np.testing.assert_allclose(real, want, rtol=1e-5)


# ----------------------------------------------------------------------------- evaluator
def random_mask(rs, U, n_items, mean_len, heavy=()):
    lists = []
    for q in range(U):
        k = int(rs.poisson(mean_len))
        if q in heavy:
            k = n_items - 4                      # leaves 4 candidates (< K)
        k = min(k, n_items)
        lists.append(sorted(rs.choice(n_items, size=k, replace=False).tolist()))
    return lists


@pytest.mark.parametrize("kind", [oracle.SCORE_NORMAL, oracle.SCORE_RUBI_BOTH])
@pytest.mark.parametrize("U,N,d,K,splits,off", [
    (300, 1000, 64, 20, 1, 0), (300, 1000, 64, 20, 3, 0), (257, 2085, 64, 20, 8, 0), (40, 744, 64, 20, 0, 0),
    (1, 33, 64, 20, 1, 0), (5, 17, 64, 20, 1, 0), (64, 1000, 32, 5, 2, 0), (100, 900, 128, 32, 4, 0),
    (70, 500, 256, 1, 1, 0), (300, 1000, 64, 20, 2, 5000), (513, 4099, 64, 30, 0, 0)])
def test_score_topk_bit_exact(ops, kind, U, N, d, K, splits, off, eval_filter):
    rs = np.random.RandomState(U + N + d + K)
    n_users = U + 50
    P = (rs.standard_normal((n_users, d)) * 0.5).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.5).astype(np.float32)
    Q += (rs.standard_normal(N).astype(np.float32) * 0.5)[:, None] * np.sign(P.mean(0, keepdims=True))   # popularity
    w = (rs.standard_normal(d) * 0.3).astype(np.float32)
    wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    user_ids = rs.permutation(n_users)[:U].astype(np.int32)
    mask_lists = random_mask(rs, U, N, 12, heavy=(0, U - 1) if N > 40 else ())
    mask_lists = [[x + off for x in row] for row in mask_lists]
    mptr, midx = oracle.csr_from_lists(mask_lists)
    c = 3.0
    # branch sigmoids: HIP vs oracle within fp32 rounding, then the SAME arrays feed both rankers
    sig_i_hip = ops.branch_sigmoid(dev(Q), dev(w))
    sig_u_hip = ops.branch_sigmoid(dev(P), dev(wu), dev(user_ids))
    np.testing.assert_allclose(sig_i_hip.cpu().numpy(), oracle.branch_sigmoid(Q, w), rtol=2e-6)
    np.testing.assert_allclose(sig_u_hip.cpu().numpy(), oracle.branch_sigmoid(P[user_ids], wu), rtol=2e-6)
    sig_i, sig_u = sig_i_hip.cpu().numpy(), sig_u_hip.cpu().numpy()
    want_v, want_i, want_c = oracle.score_topk(kind, P[user_ids], Q, K, sig_u, sig_i, c, (mptr, midx), off)
    mask = ops.CSR(dev(mptr), dev(midx if len(midx) else np.zeros(1, np.int32)))
    vals, idx = ops.score_topk(kind, dev(P), dev(user_ids), dev(Q), K, sig_u_hip, sig_i_hip, c, mask, off, splits)
    gv, gi, gc = ops.topk_merge(vals, idx)
    assert np.array_equal(gi.cpu().numpy(), want_i)
    assert np.array_equal(gc.cpu().numpy(), want_c)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), want_v.view(np.uint32))      # bit for bit
    # masked fill (what ranking a -inf masked matrix gives, batch_test.py:124-134)
    fv, fi, fc = ops.topk_merge(vals, idx, fill_mask=mask)
    ov, oi, oc = oracle.score_topk(kind, P[user_ids], Q, K, sig_u, sig_i, c, (mptr, midx), off, fill_masked=True)
    assert np.array_equal(fi.cpu().numpy(), oi)
    # the dense score matrix entry point computes the same numbers
    if U * N <= 400000:
        S = ops.score_matrix(kind, dev(P), dev(user_ids), dev(Q), sig_u_hip, sig_i_hip, c).cpu().numpy()
        rows = np.arange(U)[:, None]
        valid = want_i >= 0
        picked = S[rows, np.where(valid, want_i - off, 0)]
        assert np.array_equal(picked[valid].view(np.uint32), want_v[valid].view(np.uint32))


def test_score_topk_item_sharding_equals_unsharded(ops, eval_filter):
    """Item-sharded scoring + merge == single-shard scoring (the 1/2/4/8-GPU contract)."""
    rs = np.random.RandomState(0)
    U, N, d, K = 777, 5003, 64, 20
    P = (rs.standard_normal((U, d)) * 0.5).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.5).astype(np.float32)
    Q[100:200] = Q[300:400]                       # exact score ties across shards
    mask_lists = random_mask(rs, U, N, 20)
    mptr, midx = oracle.csr_from_lists(mask_lists)
    mask = ops.CSR(dev(mptr), dev(midx))
    Pd, Qd = dev(P), dev(Q)
    v1, i1 = ops.score_topk(oracle.SCORE_NORMAL, Pd, None, Qd, K, mask=mask, n_splits=1)
    ref_v, ref_i, ref_c = ops.topk_merge(v1, i1)
    for W in (2, 4, 8):
        bounds = [N * r // W for r in range(W + 1)]
        vs, is_ = [], []
        for r in range(W):
            lo, hi = bounds[r], bounds[r + 1]
            v, i = ops.score_topk(oracle.SCORE_NORMAL, Pd, None, Qd[lo:hi].contiguous(), K, mask=mask,
                                  item_offset=lo, n_splits=2)
            mv, mi, _ = ops.topk_merge(v, i)
            vs.append(mv); is_.append(mi)
        gv, gi, gc = ops.topk_merge(torch.stack(vs), torch.stack(is_))
        assert torch.equal(gi, ref_i) and torch.equal(gv, ref_v) and torch.equal(gc, ref_c)
    wv, wi, wc = oracle.score_topk(oracle.SCORE_NORMAL, P, Q, K, mask=(mptr, midx))
    assert np.array_equal(ref_i.cpu().numpy(), wi)


def sampled_items(N):
    """bool[N]: the items the ranking's sampling pass looks at (eval_kernels.hip sample_log2: every 2^s-th item of a window
    of 2^s tiles, phase = window index mod 2^s)"""
    S = 8
    idx = np.arange(N)
    return (idx % (32 * S)) % S == (idx // (32 * S)) % S


@pytest.mark.parametrize("splits", [1, 3])
def test_score_topk_overflowing_lists_fall_back_to_running_topk(ops, splits, eval_filter):
    """Adversarial score order for the fixed-threshold stream: every sampled item scores far
    below the rest, so the thresholds learnt from the samples admit ~everything and the candidate lists overflow;
    the device-armed fallback (running top-K kernel) must still return the exact ranking."""
    rs = np.random.RandomState(5)
    U, N, d, K = 300, 4096 * 3, 64, 20
    P = np.abs(rs.standard_normal((U, d)) * 0.5).astype(np.float32)
    Q = np.abs(rs.standard_normal((N, d)) * 0.5).astype(np.float32)
    Q[sampled_items(N)] *= -1.0                       # all-positive factors: these items score negative
    mask_lists = random_mask(rs, U, N, 10)
    mptr, midx = oracle.csr_from_lists(mask_lists)
    mask = ops.CSR(dev(mptr), dev(midx))
    want_v, want_i, want_c = oracle.score_topk(oracle.SCORE_NORMAL, P, Q, K, mask=(mptr, midx))
    vals, idx = ops.score_topk(oracle.SCORE_NORMAL, dev(P), None, dev(Q), K, mask=mask, n_splits=splits)
    gv, gi, gc = ops.topk_merge(vals, idx)
    assert np.array_equal(gi.cpu().numpy(), want_i)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), want_v.view(np.uint32))


@pytest.mark.parametrize("seed", range(24))
def test_score_topk_random_shapes(ops, seed, eval_filter):
    """Randomised shapes through the streaming ranking: query counts around the 256-user block, catalogues around
    the 32-item tile / the 8-tile sampling stride / the list-everything limit, every d, K up to 32, shards with
    an offset, heavy and empty masks, duplicated item rows (exact ties)."""
    rs = np.random.RandomState(1000 + seed)
    U = int(rs.choice([1, 3, 31, 33, 255, 257, 300, 700, 1500]))
    N = int(rs.choice([5, 31, 33, 255, 257, 1024, 1025, 1100, 2047, 2500, 4100, 9000]))
    d = int(rs.choice([32, 64, 64, 64, 128, 256]))
    K = int(rs.choice([1, 5, 20, 20, 32]))
    off = int(rs.choice([0, 0, 1000]))
    kind = int(rs.choice([oracle.SCORE_NORMAL, oracle.SCORE_RUBI_BOTH]))
    n_users = U + 17
    P = (rs.standard_normal((n_users, d)) * 0.5).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.5).astype(np.float32)
    if N > 40:
        Q[N // 2:N // 2 + 10] = Q[:10]                              # exact ties far apart
    w = (rs.standard_normal(d) * 0.3).astype(np.float32)
    wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    user_ids = rs.permutation(n_users)[:U].astype(np.int32)
    per_user = int(rs.choice([0, 3, 40]))
    mask_lists = random_mask(rs, U, N, per_user, heavy=(0,) if (N > 40 and per_user) else ()) if per_user else [[] for _ in range(U)]
    mask_lists = [[x + off for x in row] for row in mask_lists]
    mptr, midx = oracle.csr_from_lists(mask_lists)
    sig_i_hip = ops.branch_sigmoid(dev(Q), dev(w))
    sig_u_hip = ops.branch_sigmoid(dev(P), dev(wu), dev(user_ids))
    sig_i, sig_u = sig_i_hip.cpu().numpy(), sig_u_hip.cpu().numpy()
    c = float(rs.choice([0.0, 3.0, 40.0]))
    want_v, want_i, want_c = oracle.score_topk(kind, P[user_ids], Q, K, sig_u, sig_i, c, (mptr, midx), off)
    mask = ops.CSR(dev(mptr), dev(midx if len(midx) else np.zeros(1, np.int32)))
    seeds = torch.full((U, ops.SEED_WIDTH), -1, dtype=torch.int32, device="cuda")
    vals, idx = ops.score_topk(kind, dev(P), dev(user_ids), dev(Q), K, sig_u_hip, sig_i_hip, c, mask, off, seed_out=seeds)
    gv, gi, gc = ops.topk_merge(vals, idx)
    assert np.array_equal(gi.cpu().numpy(), want_i), (U, N, d, K, off, kind, per_user)
    assert np.array_equal(gc.cpu().numpy(), want_c)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), want_v.view(np.uint32))
    # ... and the same ranking again with its own best candidates as threshold seeds (in place), on slightly moved tables
    Q2 = (Q + rs.standard_normal(Q.shape).astype(np.float32) * 0.02).astype(np.float32)
    sig_i2 = ops.branch_sigmoid(dev(Q2), dev(w))
    want_v2, want_i2, want_c2 = oracle.score_topk(kind, P[user_ids], Q2, K, sig_u, sig_i2.cpu().numpy(), c, (mptr, midx), off)
    vals, idx = ops.score_topk(kind, dev(P), dev(user_ids), dev(Q2), K, sig_u_hip, sig_i2, c, mask, off, seed=seeds, seed_out=seeds)
    gv, gi, gc = ops.topk_merge(vals, idx)
    assert np.array_equal(gi.cpu().numpy(), want_i2), ("seeded", U, N, d, K, off, kind, per_user)
    assert np.array_equal(gc.cpu().numpy(), want_c2)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), want_v2.view(np.uint32))


def test_score_topk_all_scores_tie(ops, eval_filter):
    """Zero user vectors: every item scores 0 (and -0), the threshold equals every score, the candidate lists
    overflow and the fallback ranks by ascending id among the unmasked items."""
    rs = np.random.RandomState(2)
    U, N, d, K = 70, 6000, 64, 20
    P = np.zeros((U, d), np.float32)
    Q = rs.standard_normal((N, d)).astype(np.float32)
    mask_lists = random_mask(rs, U, N, 6)
    mask_lists[0] = list(range(0, 40, 2))                       # user 0 masks every other low id
    mptr, midx = oracle.csr_from_lists(mask_lists)
    mask = ops.CSR(dev(mptr), dev(midx))
    want_v, want_i, _ = oracle.score_topk(oracle.SCORE_NORMAL, P, Q, K, mask=(mptr, midx))
    vals, idx = ops.score_topk(oracle.SCORE_NORMAL, dev(P), None, dev(Q), K, mask=mask)
    gv, gi, _ = ops.topk_merge(vals, idx)
    assert np.array_equal(gi.cpu().numpy(), want_i)
    assert np.array_equal(gi.cpu().numpy()[0], np.arange(1, 40, 2)[:K])
    assert np.all(gv.cpu().numpy() == 0.0)


def test_score_topk_forced_fallback_kernel(ops):
    """MACR_TOPK_FALLBACK=1 runs the running top-K kernel unconditionally: it stays covered by the parity suite."""
    import os, subprocess, sys
    env = dict(os.environ, MACR_TOPK_FALLBACK="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_ops.py", "-k",
                        "test_score_topk_bit_exact or test_score_topk_item_sharding"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("rows,cols,K", [(7, 50, 20), (33, 744, 20), (5, 12, 10), (64, 40981, 20), (3, 25, 32), (2, 5, 8)])
def test_topk_scores_bit_exact(ops, rows, cols, K):
    rs = np.random.RandomState(rows * cols)
    s = rs.standard_normal((rows, cols)).astype(np.float32)
    s[0] = np.round(s[0])                          # heavy ties
    if cols > 8:
        s[1, 4:] = -np.inf                         # K > #finite
    want_v, want_i, want_c = oracle.topk_scores(s, K)
    gi, gv = ops.topk_scores(dev(s), K)
    assert np.array_equal(gi.cpu().numpy(), want_i)
    assert np.array_equal(gv.cpu().numpy(), want_v)       # values (−0.0 == +0.0)


def test_topk_merge_matches_oracle(ops):
    rs = np.random.RandomState(4)
    for W, U, K in ((1, 9, 20), (2, 100, 20), (8, 333, 20), (5, 17, 32), (3, 10, 1), (16, 50, 7)):
        vals = np.sort(rs.standard_normal((W, U, K)).astype(np.float32), axis=2)[:, :, ::-1].copy()
        vals = np.round(vals * 4) / 4              # ties across lists
        idx = np.stack([rs.permutation(10000)[: U * K].reshape(U, K) for _ in range(W)]).astype(np.int32)
        # ties inside one list must already be id-ascending (as the producers emit them)
        order = np.lexsort((idx, -vals), axis=2)
        vals = np.take_along_axis(vals, order, 2); idx = np.take_along_axis(idx, order, 2)
        short = rs.rand(W, U) < 0.2                # some short lists
        for s_, u_ in zip(*np.nonzero(short)):
            n = rs.randint(0, K)
            vals[s_, u_, n:] = -np.inf; idx[s_, u_, n:] = -1
        wv, wi, wc = oracle.topk_merge(vals, idx)
        gv, gi, gc = ops.topk_merge(dev(vals), dev(idx))
        assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gc.cpu().numpy(), wc)
        assert np.array_equal(gv.cpu().numpy(), wv)          # values (-0.0 == +0.0 by the tie rule)


def test_metrics_match_oracle(ops):
    rs = np.random.RandomState(8)
    U, K, N = 500, 20, 300
    rank = np.stack([rs.permutation(N)[:K] for _ in range(U)]).astype(np.int32)
    cnt = np.full(U, K, np.int32)
    cnt[:5] = [0, 1, 3, 19, 20]
    for q in range(5):
        rank[q, cnt[q]:] = -1
    gt_lists = [sorted(rs.choice(N, size=rs.randint(1, 30), replace=False).tolist()) for _ in range(U)]
    gptr, gidx = oracle.csr_from_lists(gt_lists)
    gt = ops.CSR(dev(gptr), dev(gidx))
    want = oracle.metrics_foldout(rank, (gptr, gidx))
    got = ops.metrics_foldout(dev(rank), gt).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)
    Ks = [1, 5, 20]
    want = oracle.metrics_mf(rank, cnt, (gptr, gidx), Ks)
    got = ops.metrics_mf(dev(rank), dev(cnt), gt, Ks).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-14, atol=0, equal_nan=True)
    np.testing.assert_allclose(ops.colmean(dev(want[5:])).cpu().numpy(), want[5:].mean(0), rtol=1e-13)
    f = rs.standard_normal((1000, 100)).astype(np.float32)
    np.testing.assert_allclose(ops.colmean(dev(f)).cpu().numpy(), f.astype(np.float64).mean(0), rtol=1e-12, atol=1e-15)


def test_metrics_mf_counts_a_lists_ids_and_colmean_writes_pinned_host_memory(ops):
    """macr_metrics_mf without cnt: a list's length is its number of ids >= 0 (what macr_topk_merge would report), so the
    one list per query of macr_score_topk needs no merge; macr_colmean's result lands in device memory or straight in
    pinned host memory."""
    rs = np.random.RandomState(81)
    U, K, N = 4099, 20, 300
    rank = np.stack([rs.permutation(N)[:K] for _ in range(U)]).astype(np.int32)
    cnt = np.full(U, K, np.int32)
    short = rs.choice(U, 40, replace=False)
    cnt[short] = rs.randint(0, K + 1, short.size)
    for q in short:
        rank[q, cnt[q]:] = -1
    gt_lists = [sorted(rs.choice(N, size=rs.randint(1, 30), replace=False).tolist()) for _ in range(U)]
    gptr, gidx = oracle.csr_from_lists(gt_lists)
    gt = ops.CSR(dev(gptr), dev(gidx))
    for Ks in ([20], [1, 5, 20]):
        want = oracle.metrics_mf(rank, cnt, (gptr, gidx), Ks)
        counted = ops.metrics_mf(dev(rank), None, gt, Ks)
        np.testing.assert_allclose(counted.cpu().numpy(), want, rtol=1e-14, atol=0, equal_nan=True)
        assert torch.equal(torch.nan_to_num(counted), torch.nan_to_num(ops.metrics_mf(dev(rank), dev(cnt), gt, Ks)))
        host = torch.full((4, len(Ks)), -1.0, dtype=torch.float64).pin_memory()
        on_dev = ops.colmean(counted)
        ops.colmean(counted, out=host)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(on_dev.cpu()), torch.nan_to_num(host))


@pytest.mark.parametrize("U", [1, 3, 4, 5, 63, 64, 65, 4099, 15424])
def test_metrics_mf_mean_is_metrics_mf_then_the_mean_in_one_launch(ops, U):
    """macr_metrics_mf_mean: the per-query values are macr_metrics_mf's bit for bit, the means are the oracle's per-query values
    averaged (macr_mf/train.py:286-290) -- a NaN precision (empty list) propagates like numpy's mean --, the same bits on every
    call (the tree is a function of U only), in device or pinned host memory; the ticket is left at zero (calls repeat on one
    workspace)."""
    rs = np.random.RandomState(83 + U)
    K, N = (100 if U in (5, 65, 4099) else 20), 300          # (lists longer than 64: a lane owns two rank positions)
    rank = np.stack([rs.permutation(N)[:K] for _ in range(U)]).astype(np.int32)
    cnt = np.full(U, K, np.int32)
    short = rs.choice(U, min(U, 40), replace=False)
    cnt[short] = rs.randint(1, K + 1, short.size)
    for q in short:
        rank[q, cnt[q]:] = -1
    gt_lists = [sorted(rs.choice(N, size=rs.randint(1, 30), replace=False).tolist()) for _ in range(U)]
    for q, n in zip(rs.choice(U, min(U, 12), replace=False), (63, 64, 65, 100, 128, 200, 64, 65, 1, 2, 129, 299)):
        gt_lists[q] = sorted(rs.choice(N, size=n, replace=False).tolist())      # (up to 64 ids are searched across lanes, longer lists by bisection)
    gptr, gidx = oracle.csr_from_lists(gt_lists)
    gt = ops.CSR(dev(gptr), dev(gidx))
    for Ks in ([20], [1, 5, 20], [1, 2, 3, 4, 5, 10, 15, 20]):
        ws = ops.metrics_mf_mean_workspace(U, len(Ks), torch.device("cuda", 0))
        want = oracle.metrics_mf(rank, cnt, (gptr, gidx), Ks)
        for c in (None, dev(cnt)):
            mean, per_user = ops.metrics_mf_mean(dev(rank), c, gt, Ks, ws, per_user=True)
            assert torch.equal(per_user, ops.metrics_mf(dev(rank), c, gt, Ks))
            np.testing.assert_allclose(mean.cpu().numpy(), want.mean(0), rtol=1e-13, atol=0)
            host = torch.full((4, len(Ks)), -1.0, dtype=torch.float64).pin_memory()
            again = ops.metrics_mf_mean(dev(rank), c, gt, Ks, ws, out=host)
            torch.cuda.synchronize()
            assert again.data_ptr() == host.data_ptr() and torch.equal(host, mean.cpu())
            assert int(ws[:4].view(torch.int32).item()) == 0
    if U >= 16:
        # an empty list: its precision is NaN (0 / 0 in train.py:36 with an empty r), and so is the mean of that column
        rank[3, :] = -1
        mean = ops.metrics_mf_mean(dev(rank), None, gt, [20], ops.metrics_mf_mean_workspace(U, 1, torch.device("cuda", 0)))
        m = mean.cpu().numpy()
        assert np.isnan(m[0, 0]) and np.all(np.isfinite(m[1:, 0]))


@pytest.mark.parametrize("K", [1, 20, 64, 100])
def test_metrics_foldout_completes_short_lists_like_the_merge(ops, K):
    """macr_metrics_foldout_fill on the one list per query of macr_score_topk = macr_topk_merge with its -inf fill
    (batch_test.py:124-134) followed by macr_metrics_foldout, and the oracle's metrics of the filled rankings; the
    one-wave-per-query kernel (K <= 128) against the oracle for every K; wide column means against numpy."""
    rs = np.random.RandomState(300 + K)
    U, N = 700, 400
    rank = np.stack([rs.permutation(N)[:K] for _ in range(U)]).astype(np.int32)
    vals = -np.sort(rs.standard_normal((U, K)).astype(np.float32), axis=1)
    short = rs.choice(U, 60, replace=False)
    mask_lists = [sorted(rs.choice(N, size=rs.randint(0, 2 * K + 2), replace=False).tolist()) for _ in range(U)]
    for q in short:
        n = rs.randint(0, K)
        rank[q, n:] = -1; vals[q, n:] = -np.inf
    gt_lists = [sorted(rs.choice(N, size=rs.randint(1, 30), replace=False).tolist()) for _ in range(U)]
    gptr, gidx = oracle.csr_from_lists(gt_lists)
    gt = ops.CSR(dev(gptr), dev(gidx))
    mask = ops.CSR.from_lists(mask_lists, "cuda")
    _, filled, _ = ops.topk_merge(dev(vals[None]), dev(rank[None]), mask)
    via_merge = ops.metrics_foldout(filled, gt, hr_in_ap_slot=True)
    direct = ops.metrics_foldout(dev(rank), gt, hr_in_ap_slot=True, fill_mask=mask)
    assert torch.equal(via_merge, direct)
    want = oracle.metrics_foldout(filled.cpu().numpy(), (gptr, gidx))
    got = ops.metrics_foldout(filled, gt).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)
    np.testing.assert_allclose(ops.colmean(direct).cpu().numpy(), direct.cpu().numpy().astype(np.float64).mean(0), rtol=1e-12, atol=1e-15)
    assert torch.equal(ops.colmean(direct), ops.colmean(direct))


def test_first_round_alone_says_whether_it_stands(ops, eval_filter):
    """macr_score_topk_first_round: with thresholds that hold (sampling pass, good or damaged seeds) it returns exactly
    what macr_score_topk returns and stats == {0, 0} -- written to device memory or to pinned host memory; with seeds the
    model has moved away from, stats[0] counts the query blocks the complete call would list again."""
    rs = np.random.RandomState(131)
    U, N, d, K, S = 700, 7000, 64, 20, ops.SEED_WIDTH
    P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    mask = random_mask(rs, U, N, 30, heavy=(5,))
    mcsr = ops.CSR.from_lists(mask, "cuda")
    sig_i = ops.branch_sigmoid(dev(Q), dev(w)); sig_u = ops.branch_sigmoid(dev(P), dev(wu))
    wv, wi, _ = oracle.score_topk(oracle.SCORE_RUBI_BOTH, P, Q, K, sig_u.cpu().numpy(), sig_i.cpu().numpy(), 30.0,
                                  oracle.csr_from_lists(mask))
    prev = torch.full((U, S), -1, dtype=torch.int32, device="cuda")
    ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q), K, sig_u, sig_i, 30.0, mcsr, seed_out=prev)
    rnd = torch.from_numpy(np.stack([rs.choice(N, S, replace=False) for _ in range(U)]).astype(np.int32)).cuda()
    host_stats = torch.full((2,), 77, dtype=torch.int32).pin_memory()
    dev_stats = torch.full((2,), 77, dtype=torch.int32, device="cuda")
    for name, seed, stats in (("sampled", None, dev_stats), ("sampled, host stats", None, host_stats),
                              ("seeded", prev.clone(), host_stats), ("stale", rnd, host_stats)):
        stats.fill_(77)
        seeds_out = torch.full((U, S), -1, dtype=torch.int32, device="cuda")
        v, ix = ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q), K, sig_u, sig_i, 30.0, mcsr, seed=seed,
                               seed_out=seeds_out, stats=stats, first_round=True)
        torch.cuda.synchronize()
        got = stats.cpu().numpy().tolist()
        if name == "stale":
            assert got[0] > 0 and got[1] == 0, (name, got)
            # ... and macr_score_topk_repair_round finishes that very call: same arguments, workspace and outputs
            v2, ix2 = ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q), K, sig_u, sig_i, 30.0, mcsr, seed=seed,
                                     seed_out=seeds_out, stats=stats,
                                     repair_of=(v, ix, ops._topk_ws_cache[v.device]))
            torch.cuda.synchronize()
            assert stats.cpu().numpy().tolist() == [got[0], 0], name
            assert v2 is v and np.array_equal(ix[0].cpu().numpy(), wi) and np.array_equal(v[0].cpu().numpy().view(np.uint32), wv.view(np.uint32))
            assert torch.equal(seeds_out[:, :K], ix[0]), name
            continue
        assert got == [0, 0], (name, got)
        assert np.array_equal(ix[0].cpu().numpy(), wi), name
        assert np.array_equal(v[0].cpu().numpy().view(np.uint32), wv.view(np.uint32)), name
        assert torch.equal(seeds_out[:, :K], ix[0]), name
    with pytest.raises(Exception):
        ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q), K, sig_u, sig_i, 30.0, mcsr, first_round=True)     # stats required


@pytest.mark.parametrize("shape", [(700, 7000, 64, 20), (300, 900, 32, 5), (513, 20011, 128, 32), (260, 5000, 64, 100)])
def test_prologue_is_the_branch_sigmoids_and_the_workspace_initialisation(ops, eval_filter, shape):
    """macr_score_topk_prologue + a ranking call with MACR_EVAL_WS_READY: the branch factors are macr_branch_sigmoid's bit for
    bit and the ranking -- complete call, sampled first round, seeded first round -- is what the same call returns when it
    initialises its own workspace, also when the workspace was left dirty by another ranking (and by garbage) before."""
    U, N, d, K = shape
    rs = np.random.RandomState(141 + U)
    S = ops.SEED_WIDTH
    P = (rs.standard_normal((U + 50, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    uid = dev(rs.permutation(U + 50)[:U].astype(np.int32))
    mask = random_mask(rs, U, N, 30, heavy=(5,))
    mcsr = ops.CSR.from_lists(mask, "cuda")
    Pd, Qd, wd, wud = dev(P), dev(Q), dev(w), dev(wu)
    sig_i = ops.branch_sigmoid(Qd, wd); sig_u = ops.branch_sigmoid(Pd, wud, uid)
    seeds = torch.full((U, S), -1, dtype=torch.int32, device="cuda")
    want_v, want_i = ops.score_topk(ops.SCORE_RUBI_BOTH, Pd, uid, Qd, K, sig_u, sig_i, 30.0, mcsr, seed_out=seeds, filter=eval_filter)
    want_v, want_i = want_v.clone(), want_i.clone()
    stats = torch.full((2,), 77, dtype=torch.int32).pin_memory()
    use_seeds = K <= 32 and ops._lib.lib().macr_score_topk_uses_seeds(U, N, d)
    for name0, seed, first in (("complete", None, False), ("sampled first round", None, True), ("seeded first round", seeds.clone(), True)):
        if seed is not None and not use_seeds:
            continue
        # (fused: macr_score_topk_prologue_prep -- the fp16 filter's operand copies written by the same launch, abi 15)
        for fused in ((False, True) if (eval_filter == "f16" and use_seeds) else (False,)):
            name = name0 + (" + prep" if fused else "")
            ops._topk_ws_cache[Qd.device].fill_(0xA5)                 # whatever was there before
            if fused:
                gi, gu = ops.score_topk_prologue_prep(ops.SCORE_RUBI_BOTH, Pd, uid, Qd, K, wd, wud, 30.0, seeded_first_round=first and seed is not None)
            else:
                gi, gu = ops.score_topk_prologue(Pd, uid, Qd, K, wd, wud, seeded_first_round=first and seed is not None, filter=eval_filter)
            assert torch.equal(gi, sig_i) and torch.equal(gu, sig_u), name
            stats.fill_(77)
            v, ix = ops.score_topk(ops.SCORE_RUBI_BOTH, Pd, uid, Qd, K, gu, gi, 30.0, mcsr, seed=seed, seed_out=torch.empty_like(seeds),
                                   stats=stats, first_round=first, filter=eval_filter, ws_ready=True, prep_ready=fused)
            torch.cuda.synchronize()
            if eval_filter == "f16" and stats.tolist()[0] != 0:
                # (y - 30) sig_i sig_u on untrained rows packs the top of every query closer than the fp16 filter's margin at d = 128:
                # lists overflow and query blocks are listed again -- the Evaluator then steps down to bf16.  Same ranking either way.
                assert stats.tolist()[1] == 0, (name, stats.tolist())
                if first:
                    continue                                        # (the first round says so itself: its rows are not the ranking)
            else:
                assert stats.tolist() == [0, 0], (name, stats.tolist())
            assert torch.equal(ix, want_i) and torch.equal(v.view(torch.int32), want_v.view(torch.int32)), name
    # one-branch scores: no query factors
    gi, gu = ops.score_topk_prologue(Pd, uid, Qd, K, wd, None, filter=eval_filter)
    assert gu is None and torch.equal(gi, sig_i)
    v, ix = ops.score_topk(ops.SCORE_RUBI, Pd, uid, Qd, K, None, gi, 30.0, mcsr, filter=eval_filter, ws_ready=True)
    v0, ix0 = ops.score_topk(ops.SCORE_RUBI, Pd, uid, Qd, K, None, sig_i, 30.0, mcsr, filter=eval_filter)
    assert torch.equal(ix, ix0) and torch.equal(v.view(torch.int32), v0.view(torch.int32))


def test_golden_cpp_evaluator_cases(ops, golden_dir):
    """G7: raw outputs of the reference's C++ evaluator (tools.h + evaluate_foldout.h) on stored inputs."""
    import os
    z = np.load(os.path.join(golden_dir, "G7_cpp_eval_cases.npz"))
    for name in "abcd":
        s, k = z[name + "_scores"], int(z[name + "_k"])
        lens = z[name + "_gt_len"]
        flat = z[name + "_gt_flat"]
        gptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        gi, _ = ops.topk_scores(dev(s), k)
        if name != "d":                            # tie-free: rankings must equal the reference's exactly
            assert np.array_equal(gi.cpu().numpy(), z[name + "_rankings"])
        res = ops.metrics_foldout(gi, ops.CSR(dev(gptr), dev(flat))).cpu().numpy()
        np.testing.assert_allclose(res, z[name + "_results"], rtol=2e-7, atol=0)


def test_golden_cpp_evaluator_cases_through_the_reference_abi(golden_dir):
    """G7 through libmacr_eval_compat.so: c_top_k_array_index / evaluate_foldout with the reference's own signatures
    (host pointers, int** ground truths in any order, void return) -- what the reference's .pyx links against."""
    import ctypes
    import os
    from macr_amd import _lib
    L = _lib.compat_lib()
    z = np.load(os.path.join(golden_dir, "G7_cpp_eval_cases.npz"))
    for name in "abcd":
        s, k = np.ascontiguousarray(z[name + "_scores"], dtype=np.float32), int(z[name + "_k"])
        lens, flat = z[name + "_gt_len"].astype(np.int32), z[name + "_gt_flat"].astype(np.int32)
        rows, cols = s.shape
        rank = np.full((rows, k), -7, np.int32)
        L.c_top_k_array_index(s.ctypes.data, cols, rows, k, 4, rank.ctypes.data)
        assert L.macr_eval_compat_status() == 0, L.macr_eval_compat_error()
        if name != "d":
            assert np.array_equal(rank, z[name + "_rankings"])
        # int **ground_truths: borrowed row pointers, rows given in DESCENDING order to show any order is accepted
        starts = np.concatenate([[0], np.cumsum(lens)])
        rows_gt = [np.ascontiguousarray(flat[starts[u]:starts[u + 1]][::-1]) for u in range(rows)]
        ptrs = (ctypes.c_void_p * rows)(*[r.ctypes.data if len(r) else None for r in rows_gt])
        res = np.zeros((rows, 5 * k), np.float32)
        L.evaluate_foldout(rows, rank.ctypes.data, k, ctypes.cast(ptrs, ctypes.c_void_p), lens.ctypes.data, 4, res.ctypes.data)
        assert L.macr_eval_compat_status() == 0, L.macr_eval_compat_error()
        np.testing.assert_allclose(res, z[name + "_results"], rtol=2e-7, atol=0)
    # no error channel in the reference: a failed call poisons its output and sets the status
    bad = np.zeros((2, 200), np.float32)
    out = np.zeros((2, 200), np.int32)
    L.c_top_k_array_index(bad.ctypes.data, 200, 2, 0, 1, out.ctypes.data)           # top_k = 0
    assert L.macr_eval_compat_status() != 0
    L.c_top_k_array_index(None, 200, 2, 20, 1, out.ctypes.data)
    assert L.macr_eval_compat_status() != 0 and (out.ravel()[:40] == -1).all()       # rows_num * top_k ids, contiguous
    # top_k beyond 32 (tools.h:13-22 has no bound; the tuning scripts rank 100): the wide kernel, same tie rule
    rs = np.random.RandomState(3)
    # ... and beyond 128 (round 5: rounds of 128 positions bounded by the last key of the round before -- any rank_len, as
    # evaluate_foldout.h:115-118 and tools.h:24 take it), through the metrics too
    for cols, k in [(744, 100), (90, 64), (5000, 128), (50, 50), (33, 33), (5000, 129), (744, 300), (40981, 500), (700, 700),
                    (260, 257)]:
        sc = rs.standard_normal((37, cols)).astype(np.float32)
        sc[:, ::7] = np.round(sc[:, ::7])                                            # ties
        sc[3, : cols // 2] = -np.inf
        out = np.zeros((37, k), np.int32)
        L.c_top_k_array_index(sc.ctypes.data, cols, 37, k, 4, out.ctypes.data)
        assert L.macr_eval_compat_status() == 0, L.macr_eval_compat_error()
        want = oracle.topk_scores(sc, k)[1]
        assert np.array_equal(out, want), (cols, k)
        if k > 128:
            lens = np.full(37, 9, np.int32)
            rows_gt = [np.ascontiguousarray(rs.choice(cols, 9, replace=False).astype(np.int32)) for _ in range(37)]
            ptrs = (ctypes.c_void_p * 37)(*[r.ctypes.data for r in rows_gt])
            res = np.zeros((37, 5 * k), np.float32)
            L.evaluate_foldout(37, out.ctypes.data, k, ctypes.cast(ptrs, ctypes.c_void_p), lens.ctypes.data, 4, res.ctypes.data)
            assert L.macr_eval_compat_status() == 0, L.macr_eval_compat_error()
            gp = np.arange(0, 9 * 37 + 1, 9, dtype=np.int32)
            ref = oracle.metrics_foldout(out, (gp, np.concatenate([np.sort(r) for r in rows_gt]).astype(np.int32)))
            np.testing.assert_allclose(res, ref, rtol=2e-7, atol=0)


def test_errors_are_loud(ops):
    with pytest.raises(ops.MacrError):
        ops.topk_scores(dev(np.zeros((2, 5), np.float32)), 0)               # K < 1
    with pytest.raises(ops.MacrError):
        ops.branch_sigmoid(dev(np.zeros((4, 48), np.float32)), dev(np.zeros(48, np.float32)))   # unsupported d
    with pytest.raises(ops.MacrError):
        ops.branch_sigmoid(torch.zeros(4, 64), torch.zeros(64))             # CPU tensors: no fallback


def test_seeded_thresholds_do_not_change_the_ranking(ops, eval_filter):
    """macr_score_topk with seed ids (k_tau_seed): whatever the seeds -- the best candidates of the previous ranking,
    random items, masked, repeated or invalid ids -- the result is the exact ranking (the seeds only decide how tight
    the threshold is); bit-equal to the oracle."""
    rs = np.random.RandomState(31)
    U, N, d, K, S = 900, 6000, 64, 20, ops.SEED_WIDTH
    P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    Q2 = (Q + rs.standard_normal((N, d)).astype(np.float32) * 0.05).astype(np.float32)       # the tables moved a little
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    mask = random_mask(rs, U, N, 30, heavy=(5,))                 # user 5 keeps 4 candidates (< K)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    sig_i = ops.branch_sigmoid(dev(Q2), dev(w)); sig_u = ops.branch_sigmoid(dev(P), dev(wu))
    wv, wi, wc = oracle.score_topk(oracle.SCORE_RUBI_BOTH, P, Q2, K, sig_u.cpu().numpy(), sig_i.cpu().numpy(), 30.0,
                                   oracle.csr_from_lists(mask))
    # previous ranking on the OLD tables leaves its best S candidates per query: the first K are its result
    so = ops.branch_sigmoid(dev(Q), dev(w))
    prev = torch.full((U, S), -7, dtype=torch.int32, device="cuda")
    v0, i0 = ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q), K, sig_u, so, 30.0, mcsr, seed_out=prev)
    assert torch.equal(prev[:, :K], i0[0]) and int(prev.min()) >= -1
    ov, oi, _ = oracle.score_topk(oracle.SCORE_RUBI_BOTH, P, Q, S, sig_u.cpu().numpy(), so.cpu().numpy(), 30.0, oracle.csr_from_lists(mask))
    if eval_filter == "f16":
        # (the first K are exact; the candidates behind them keep their FILTER scores, and the fp16 filter's order of near-equal
        # scores need not be the exact one: good seeds all the same -- every one of them among the exact top 2 S)
        o2 = oracle.score_topk(oracle.SCORE_RUBI_BOTH, P, Q, 2 * S, sig_u.cpu().numpy(), so.cpu().numpy(), 30.0, oracle.csr_from_lists(mask))[1]
        pn = prev.cpu().numpy()
        assert np.array_equal(pn[:, :K], oi[:, :K])
        assert all(set(pn[q][pn[q] >= 0].tolist()) <= set(o2[q].tolist()) for q in range(U))
    else:
        assert np.array_equal(prev.cpu().numpy(), oi)            # ... and all S are the exact top S
    rnd = torch.from_numpy(np.stack([rs.choice(N, S, replace=False) for _ in range(U)]).astype(np.int32)).cuda()
    bad = prev.clone(); bad[::7, 3] = -1; bad[1::7, 0] = N + 5; bad[2::7, 4] = bad[2::7, 9]        # invalid and repeated ids
    bad[3::7, 1] = torch.tensor([mask[q][0] for q in range(3, U, 7)], dtype=torch.int32, device="cuda")   # and masked items
    worst = prev.clone(); worst[:, K - 2:] = worst[:, K - 2:K - 1]   # K-1 distinct seeds: one short of a bound
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    seen = {}
    for name, seed in (("previous", prev), ("random", rnd), ("damaged", bad), ("too few", worst), ("none", None)):
        out = None if seed is None else seed.clone()
        v, ix = ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q2), K, sig_u, sig_i, 30.0, mcsr, seed=out, seed_out=out,
                               stats=stats)
        val, idx, cnt = ops.topk_merge(v, ix)
        assert np.array_equal(idx.cpu().numpy(), wi), name
        assert np.array_equal(val.cpu().numpy().view(np.uint32), wv.view(np.uint32)), name
        if out is not None:                                      # in place: the seeds of the next ranking
            assert torch.equal(out[:, :K], ix[0]), name
        seen[name] = stats.cpu().numpy().copy()
    # good seeds and the sampling pass list a few dozen items per query.  A damaged set still bounds (K of its 32 seeds
    # are good).  Random seeds put the threshold deep in the catalogue, too few seeds give none: those query blocks are
    # listed again behind a sampling pass -- and that is enough (no exact fallback).
    assert seen["previous"].tolist() == [0, 0] and seen["none"].tolist() == [0, 0] and seen["damaged"].tolist() == [0, 0]
    assert seen["random"][0] > 0 and seen["random"][1] == 0
    assert seen["too few"][0] == (U + 255) // 256 and seen["too few"][1] == 0


def listing_visit_rank(N):
    """int[N]: position of every item in the order the listing pass visits the catalogue (eval_kernels.hip
    k_score_stream: visit v is tile (v * S) mod T of 32 items, S the first number coprime to T from 0.618 T on)"""
    from math import gcd
    T = (N + 31) // 32
    S = max(int(np.float32(0.6180339) * np.float32(T)), 1)
    while gcd(S, T) != 1:
        S += 1
    rank = np.empty(T, np.int64)
    rank[(np.arange(T, dtype=np.int64) * S) % T] = np.arange(T)
    idx = np.arange(N)
    return rank[idx // 32] * 32 + idx % 32


def test_score_topk_second_overflow_arms_the_exact_kernel(ops, eval_filter):
    """Scores that rise along the order the listing pass visits the catalogue in: whatever a first (cut) list holds is
    the bottom of its range of visits.  With seeds (the earliest-visited items) the repair round samples the re-listed
    query blocks and its thresholds hold; without seeds on a catalogue whose sampled items score below everything else,
    the sampled thresholds are useless, the thresholds taken from the cut lists are loose again, the lists overflow a
    second time and the running top-K kernel ranks.  Exact either way."""
    rs = np.random.RandomState(77)
    U, N, d, K = 200, 4096 * 3, 64, 20
    vrank = listing_visit_rank(N)
    P = np.abs(rs.standard_normal((U, d)) * 0.5).astype(np.float32)
    Q = (np.abs(rs.standard_normal((N, d)) * 0.5) * (1.0 + 4.0 * vrank[:, None] / N)).astype(np.float32)
    mask = random_mask(rs, U, N, 10)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    first = np.argsort(vrank)[:90]
    low = np.stack([np.array([x for x in first if x not in set(mask[q])][:ops.SEED_WIDTH]) for q in range(U)]).astype(np.int32)
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    for name, items, seed, want_stats in (("seeded", Q, dev(low), [(U + 255) // 256, 0]),
                                          ("sampled tiles negative", np.where(sampled_items(N)[:, None], -Q, Q), None,
                                           [(U + 255) // 256, 1])):
        wv, wi, _ = oracle.score_topk(oracle.SCORE_NORMAL, P, items, K, mask=oracle.csr_from_lists(mask))
        v, ix = ops.score_topk(ops.SCORE_NORMAL, dev(P), None, dev(items), K, mask=mcsr, seed=seed, stats=stats)
        val, idx, _ = ops.topk_merge(v, ix)
        assert np.array_equal(idx.cpu().numpy(), wi), name
        assert np.array_equal(val.cpu().numpy().view(np.uint32), wv.view(np.uint32)), name
        assert stats.cpu().numpy().tolist() == want_stats, name


def test_first_round_plus_repair_round_is_the_complete_call(ops, eval_filter):
    """The two halves of macr_score_topk on the cases of the test above: a seeded first round whose lists overflow is
    finished by the repair round (its thresholds hold); an unseeded one on the catalogue whose sampled items score below
    everything else overflows again in the repair round and the exact kernel ranks -- same results and stats as the
    complete call."""
    rs = np.random.RandomState(77)
    U, N, d, K = 200, 4096 * 3, 64, 20
    vrank = listing_visit_rank(N)
    P = np.abs(rs.standard_normal((U, d)) * 0.5).astype(np.float32)
    Q = (np.abs(rs.standard_normal((N, d)) * 0.5) * (1.0 + 4.0 * vrank[:, None] / N)).astype(np.float32)
    mask = random_mask(rs, U, N, 10)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    first = np.argsort(vrank)[:90]
    low = np.stack([np.array([x for x in first if x not in set(mask[q])][:ops.SEED_WIDTH]) for q in range(U)]).astype(np.int32)
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    for name, items, seed, want_stats in (("seeded", Q, dev(low), [(U + 255) // 256, 0]),
                                          ("sampled tiles negative", np.where(sampled_items(N)[:, None], -Q, Q), None,
                                           [(U + 255) // 256, 1])):
        wv, wi, _ = oracle.score_topk(oracle.SCORE_NORMAL, P, items, K, mask=oracle.csr_from_lists(mask))
        v, ix = ops.score_topk(ops.SCORE_NORMAL, dev(P), None, dev(items), K, mask=mcsr, seed=seed, stats=stats, first_round=True)
        assert stats.cpu().numpy().tolist() == [want_stats[0], 0], name
        ops.score_topk(ops.SCORE_NORMAL, dev(P), None, dev(items), K, mask=mcsr, seed=seed, stats=stats,
                       repair_of=(v, ix, ops._topk_ws_cache[v.device]))
        assert np.array_equal(ix[0].cpu().numpy(), wi), name
        assert np.array_equal(v[0].cpu().numpy().view(np.uint32), wv.view(np.uint32)), name
        assert stats.cpu().numpy().tolist() == want_stats, name


def test_one_degenerate_query_sends_only_its_block_through_the_exact_kernel(ops, eval_filter):
    """A user whose branch factor sigmoid(e_u . w_user) has gone to zero scores every item 0: every item ties, every list of
    hers overflows in both rounds and the exact kernel must rank her (ties by ascending id).  The exact kernel then runs for
    the blocks of 256 queries the repair round listed again and for no other: the ranking of ALL queries is the oracle's, and
    the stats say one block was listed again and the fallback ran.  (Seen at the configs[4] shape late in a run: one such user
    among 100 000 sent the whole evaluation through the exact kernel, 445 ms instead of 87.)"""
    rs = np.random.RandomState(5)
    U, N, d, K = 1500, 4096 * 3, 64, 20
    P = (rs.standard_normal((U, d)) * 0.5).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.5).astype(np.float32)
    sig_i = (1.0 / (1.0 + np.exp(-rs.standard_normal(N)))).astype(np.float32)
    sig_u = (1.0 / (1.0 + np.exp(-rs.standard_normal(U)))).astype(np.float32)
    bad = 700                                               # in the third block of 256 queries
    sig_u[bad] = 0.0
    mask = random_mask(rs, U, N, 10)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    wv, wi, _ = oracle.score_topk(oracle.SCORE_RUBI_BOTH, P, Q, K, sig_u=sig_u, sig_i=sig_i, c=3.0, mask=oracle.csr_from_lists(mask))
    v, ix = ops.score_topk(ops.SCORE_RUBI_BOTH, dev(P), None, dev(Q), K, dev(sig_u), dev(sig_i), 3.0, mask=mcsr, stats=stats)
    val, idx, _ = ops.topk_merge(v, ix)
    assert np.array_equal(idx.cpu().numpy(), wi)
    ok = np.arange(U) != bad
    assert np.array_equal(val.cpu().numpy()[ok].view(np.uint32), wv[ok].view(np.uint32))
    assert np.array_equal(val.cpu().numpy()[bad], wv[bad])          # (zeros of either sign)
    unmasked = [x for x in range(N) if x not in set(mask[bad])][:K]
    assert idx[bad].cpu().numpy().tolist() == unmasked      # every score of hers is +-0: ascending id decides
    assert stats.cpu().numpy().tolist() == [1, 1]


@pytest.mark.parametrize("L,hubs", [(1, False), (2, True), (3, True)])
def test_lgcn_batch_row_sparse_layers_equal_dense_layers(ops, L, hubs):
    """The LightGCN step computes its last forward layer for the batch's rows only and gathers, in the first backward
    layer, from the batch's rows only (spmm_kernels.hip kSparseOut / kSparseIn).  Against MACR_STEP_DENSE_LAYERS on the
    same state: identical losses (the forward values are bit-identical), gradients and updated tables equal up to the
    summation order of the first backward layer -- with hub rows that the plan splits, and with 1, 2 and 3 layers."""
    import scipy.sparse as sp
    rs = np.random.RandomState(12)
    n_users, n_items, d, B = 3000, 1200, 64, 512
    rows = rs.randint(0, n_users, 30000); cols = (rs.zipf(1.25, 30000) % n_items) if hubs else rs.randint(0, n_items, 30000)
    R = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_users, n_items)).tocsr()
    R.data[:] = 1.0
    N = n_users + n_items
    A = sp.bmat([[None, R], [R.T, None]]).tocsr()
    deg = np.asarray(A.sum(1)).ravel(); dinv = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0.0)
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32); A.sort_indices()
    if hubs:
        assert np.diff(A.indptr).max() > 512                                # at least one row is split by the plan
    T0 = (rs.standard_normal((N, d)) * 0.1).astype(np.float32)
    w0, wu0 = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    adj = ops.CSR.from_scipy(A, "cuda")
    hyper = ops.make_hyper(1e-3, 1e-4, 1e-2, 1e-3, B)
    states = [ops.LGCNState(dev(T0.copy()), n_users, n_items, dev(w0.copy()), dev(wu0.copy()), adj, L, hyper, B) for _ in range(2)]
    for t in range(3):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.25, B) % n_items).astype(np.int32)
        j = rs.randint(0, n_items, B).astype(np.int32)
        la = states[0].step(ops.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j)).cpu().numpy().copy()
        lb = states[1].step(ops.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j), dense_layers=True).cpu().numpy().copy()
        if t == 0:
            assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
            ga, gb = states[0].mT.cpu().numpy() / 0.1, states[1].mT.cpu().numpy() / 0.1
            np.testing.assert_allclose(ga, gb, rtol=2e-5, atol=2e-7 * np.abs(gb).max())
        np.testing.assert_allclose(la, lb, rtol=2e-6)
        np.testing.assert_allclose(states[0].T.cpu().numpy(), states[1].T.cpu().numpy(), rtol=0, atol=2e-3 * 1e-3 * (t + 1))
    # the loss-only pass takes the sparse forward too
    lo_a = states[0].step(ops.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j), loss_only=True).cpu().numpy().copy()
    lo_b = states[0].step(ops.LOSS_RUBIBCEBOTH, dev(u), dev(i), dev(j), loss_only=True, dense_layers=True).cpu().numpy().copy()
    assert np.array_equal(lo_a.view(np.uint32), lo_b.view(np.uint32))


@pytest.mark.gpu
def test_tables_of_another_dtype_are_refused(ops):
    """the kernels read fp32 through raw pointers: a float64 table (e.g. made under another default dtype) must be
    refused by the host layer, never reinterpreted"""
    P = torch.zeros((64, 64), dtype=torch.float64, device="cuda")
    Q = torch.zeros((32, 64), dtype=torch.float32, device="cuda")
    w = torch.zeros(64, dtype=torch.float32, device="cuda")
    with pytest.raises(TypeError):
        ops.MFState(P, Q, w, w.clone(), ops.make_hyper(1e-3, 1e-5, 1e-3, 1e-3, 16), 16)


@pytest.mark.gpu
@pytest.mark.parametrize("hot", [False, True])
@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("kind", ["normalbce", "rubibceboth"])
def test_large_batch_indexed_adam_equals_segment_reduce(ops, d, kind, hot, monkeypatch):
    """B > 8192, a step complete in one call: the Adam pass sums the staged gradient rows of every row with at most 16
    references itself (adam_block INDEXED; k_seg_scan writes the flags that name them), rows with more go through gP/gQ
    (k_seg_sum).  MACR_SEG_UNFUSED=1 brings k_seg_reduce back, which writes every row's sum to gP/gQ in chunks of its
    own: same step up to the order of the additions; scratch and flags clean after every step on both paths.  hot: Zipf
    positives (one item with thousands of references: the gallop + bisection for the end of its run, many work items),
    otherwise no row above 16 references (no work item at all)."""
    rs = np.random.RandomState(5 + d)
    n_users, n_items, B = 30000, (9000 if hot else 60000), 20000
    K = ops.LOSS_NORMALBCE if kind == "normalbce" else ops.LOSS_RUBIBCEBOTH
    P0 = (rs.standard_normal((n_users, d)) * 0.1).astype(np.float32)
    Q0 = (rs.standard_normal((n_items, d)) * 0.1).astype(np.float32)
    w0, wu0 = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    hyper = ops.make_hyper(1e-3, 1e-4, 1e-2, 1e-3, B)
    states = [ops.MFState(dev(P0), dev(Q0), dev(w0), dev(wu0), hyper, B) for _ in range(2)]
    for t in range(3):
        u = rs.randint(0, n_users, B).astype(np.int32)
        i = ((rs.zipf(1.2, B) % n_items) if hot else rs.randint(0, n_items, B)).astype(np.int32)
        j = rs.randint(0, n_items, B).astype(np.int32)
        most = max(np.bincount(np.concatenate([i, j])).max(), np.bincount(u).max())
        assert most > 1000 if hot else most <= 16
        out = []
        for k, st in enumerate(states):
            if k == 1:
                monkeypatch.setenv("MACR_SEG_UNFUSED", "1")
            else:
                monkeypatch.delenv("MACR_SEG_UNFUSED", raising=False)
            ops.timing_begin()
            out.append(st.step(K, dev(u), dev(i), dev(j)).cpu().numpy().copy())
            names = {n for n, _ in ops.timing_end(64)}
            want = {"adam_indexed", "seg_index", "seg_sum"} if k == 0 else {"adam_dense", "seg_reduce"}
            assert names & {"adam_indexed", "seg_index", "seg_sum", "adam_dense", "seg_reduce"} == want, names
        np.testing.assert_allclose(out[0], out[1], rtol=1e-6)
        for name in ("P", "Q", "mP", "vP", "mQ", "vQ", "w", "wu"):
            a, b = getattr(states[0], name), getattr(states[1], name)
            tol = 1e-3 * 1e-3 if name in ("P", "Q", "w", "wu") else 2e-5 * float(b.abs().max())
            torch.testing.assert_close(a, b, rtol=0, atol=tol, msg=lambda m: "%s step %d: %s" % (name, t, m))
        for st in states:
            assert not st.tP.any() and not st.tQ.any() and not st.gP.any() and not st.gQ.any()
    monkeypatch.delenv("MACR_SEG_UNFUSED", raising=False)


@pytest.mark.parametrize("d", [64, 256])
def test_bf16_filter_scores_a_crowded_top_exactly(ops, d):
    """Two hundred near-copies of one popular item: for every user the best ~200 scores differ by less than the bf16
    filter's error bound, so its 64 best candidates by bf16 score do not settle the exact top 20 -- the selection must
    notice (more than 64 candidates inside two margins of the K-th) and score every listed candidate in fp32 instead.
    The ranking is the oracle's, bit for bit, near-ties included, in one round under either filter."""
    rs = np.random.RandomState(91)
    U, N, K = 300, 6000, 20
    P = (rs.standard_normal((U, d)) * 0.3 + 0.4).astype(np.float32)              # users share a direction
    Q = (rs.standard_normal((N, d)) * 0.3).astype(np.float32)
    pop = (np.ones(d) * 0.5).astype(np.float32)
    where = rs.choice(N, 200, replace=False)
    Q[where] = pop + (rs.standard_normal((200, d)) * 2e-7).astype(np.float32)    # 200 items within ~1e-6 of each other
    mask = random_mask(rs, U, N, 8)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    wv, wi, _ = oracle.score_topk(oracle.SCORE_NORMAL, P, Q, K, mask=oracle.csr_from_lists(mask))
    assert np.isin(wi, where).mean() > 0.95                                      # the crowd is the top
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    seeds = torch.full((U, ops.SEED_WIDTH), -1, dtype=torch.int32, device="cuda")
    try:
        for filt in ("f32", "bf16", "f16"):
            ops.set_eval_filter(filt)
            for seed in (None, seeds):                                           # sampled thresholds, then seeded ones
                v, ix = ops.score_topk(ops.SCORE_NORMAL, dev(P), None, dev(Q), K, mask=mcsr, stats=stats, seed=seed, seed_out=seeds)
                val, idx, _ = ops.topk_merge(v, ix)
                assert np.array_equal(idx.cpu().numpy(), wi), filt
                assert np.array_equal(val.cpu().numpy().view(np.uint32), wv.view(np.uint32)), filt
                assert stats.cpu().numpy().tolist() == [0, 0], (filt, stats.cpu().numpy())
    finally:
        ops.set_eval_filter("env")


@pytest.mark.parametrize("bad_blocks", [(0,), (1, 3), (0, 1, 2, 3, 4)])
def test_repair_round_of_a_few_query_blocks(ops, eval_filter, bad_blocks):
    """Seeds that are useless for SOME blocks of 256 queries (random item ids) and fresh for the others: the repair round
    lists the bad blocks again -- in the compact list layout when they are few (their lists take the whole buffer, more
    result slots per query block, more of the grid) -- and must return the oracle's ranking for everybody, the untouched
    blocks included; stats[0] = the number of re-listed blocks."""
    rs = np.random.RandomState(123)
    U, N, d, K = 5 * 256 - 40, 9000, 64, 20
    P = (rs.standard_normal((U, d)) * 0.5).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.5).astype(np.float32)
    Q += (rs.standard_normal(N).astype(np.float32) * 0.5)[:, None] * np.sign(P.mean(0, keepdims=True))
    mask = random_mask(rs, U, N, 10)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    wv, wi, _ = oracle.score_topk(oracle.SCORE_NORMAL, P, Q, K, mask=oracle.csr_from_lists(mask))
    seeds = torch.full((U, ops.SEED_WIDTH), -1, dtype=torch.int32, device="cuda")
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.score_topk(ops.SCORE_NORMAL, dev(P), None, dev(Q), K, mask=mcsr, seed_out=seeds)
    bad = seeds.clone()
    for b in bad_blocks:      # the lowest-scoring items as seeds: thresholds far too loose, lists overflow
        lo, hi = b * 256, min(U, (b + 1) * 256)
        worst = np.argsort((P[lo:hi] @ Q.T), axis=1)[:, :ops.SEED_WIDTH].astype(np.int32)
        bad[lo:hi] = dev(worst)
    v, ix = ops.score_topk(ops.SCORE_NORMAL, dev(P), None, dev(Q), K, mask=mcsr, seed=bad, seed_out=bad, stats=stats)
    val, idx, _ = ops.topk_merge(v, ix)
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.array_equal(val.cpu().numpy().view(np.uint32), wv.view(np.uint32))
    assert stats.cpu().numpy().tolist() == [len(bad_blocks), 0]
    # ... and the seeds it left are good again: a ranking from them needs no repair
    v, ix = ops.score_topk(ops.SCORE_NORMAL, dev(P), None, dev(Q), K, mask=mcsr, seed=bad, seed_out=bad, stats=stats)
    val, idx, _ = ops.topk_merge(v, ix)
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert stats.cpu().numpy().tolist() == [0, 0]


# ---------------------------------------------------------------------------------------------------------------------
# The invariant the bf16 candidate filter rests on: |raw product of the bf16 kernels - fp32 fmaf chain| <= the margin the
# filter subtracts from a query's threshold, element-wise (eval_kernels.hip filter_rel / filter_margin).  Random data never
# attains a worst-case bound, so the operands here are built to: same-sign products (sum_k |u_k q_k| = |s|), u parallel to
# q (Cauchy-Schwarz tight), split residuals at their maximum, magnitudes over 24 binades in one row, one dominant term,
# values on bf16 rounding ties.  macr_test_bf16_products runs the listing pass's own MFMA sequence.
# ---------------------------------------------------------------------------------------------------------------------
def _with_low_bits(x, low):
    """fp32 array whose low 16 mantissa bits are `low` (what the two bf16 roundings see)"""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((b & np.uint32(0xffff0000)) | np.asarray(low, dtype=np.uint32)).view(np.float32)


def adversarial_operands(case, U, N, d, rs):
    ties = np.array([0x8000, 0x7f80, 0x807f, 0x7fff, 0x8080, 0xff80, 0x0080, 0x7f7f, 0x80ff, 0xffff], dtype=np.uint32)
    if case == "same_sign":                      # every product positive: no cancellation anywhere in the accumulation
        P = rs.uniform(0.5, 1.0, (U, d)); Q = rs.uniform(0.5, 1.0, (N, d))
    elif case == "aligned":                      # items are (scaled) copies of the users: |u . q| = |u| |q|
        P = rs.standard_normal((U, d))
        Q = P[rs.randint(0, U, N)] * rs.uniform(0.9, 1.0, (N, 1)) * rs.choice([-1.0, 1.0], (N, 1))
    elif case == "worst_residual":               # same sign, parallel, every mantissa with its low 16 bits on a tie / just off it
        P = _with_low_bits(rs.uniform(1.0, 2.0, (U, d)), ties[rs.randint(0, len(ties), (U, d))])
        Q = _with_low_bits(rs.uniform(1.0, 2.0, (N, d)), ties[rs.randint(0, len(ties), (N, d))])
    elif case == "binades":                      # magnitudes 2^-20 .. 2^4 inside one row, random signs
        P = np.ldexp(rs.uniform(1.0, 2.0, (U, d)), rs.randint(-20, 5, (U, d))) * rs.choice([-1.0, 1.0], (U, d))
        Q = np.ldexp(rs.uniform(1.0, 2.0, (N, d)), rs.randint(-20, 5, (N, d))) * rs.choice([-1.0, 1.0], (N, d))
    elif case == "binades_same_sign":
        P = np.ldexp(rs.uniform(1.0, 2.0, (U, d)), rs.randint(-20, 5, (U, d)))
        Q = np.ldexp(rs.uniform(1.0, 2.0, (N, d)), rs.randint(-20, 5, (N, d)))
    elif case == "dominant":                     # one term carries the product, the others sit 2^-12 .. 2^-24 below it
        P = rs.standard_normal((U, d)) * 2.0 ** -12; Q = rs.standard_normal((N, d)) * 2.0 ** -12
        k = rs.randint(0, d, U); P[np.arange(U), k] = rs.uniform(1.0, 2.0, U) * 64.0
        Q[np.arange(N), rs.randint(0, d, N)] = rs.uniform(1.0, 2.0, N) * 64.0
        Q[: N // 2, :] = np.abs(Q[: N // 2, :]); Q[np.arange(N // 2), k[rs.randint(0, U, N // 2)]] = 100.0
    elif case == "ties":                         # every value exactly half way between two bf16 numbers (both roundings tie)
        P = _with_low_bits(rs.standard_normal((U, d)), 0x8000); Q = _with_low_bits(rs.standard_normal((N, d)), 0x8000)
    elif case == "lo_ties":                      # hi exact-ish, the SECOND rounding on a tie
        P = _with_low_bits(rs.standard_normal((U, d)), 0x0080); Q = _with_low_bits(rs.standard_normal((N, d)), 0x7f80)
    elif case == "big_plus_small":               # 2^24 + sixteen terms below half an ulp of it: how the accumulator rounds
        P = np.ones((U, d)); Q = np.full((N, d), 1.0 + 2.0 ** -7)
        P[:, 0] = 4096.0; Q[:, 0] = 4096.0
        Q[N // 2:, 1:] *= -1.0
    elif case == "gaussian":
        P = rs.standard_normal((U, d)) * 0.3; Q = rs.standard_normal((N, d)) * 0.3
    else:
        raise ValueError(case)
    return np.ascontiguousarray(P, dtype=np.float32), np.ascontiguousarray(Q, dtype=np.float32)


BF16_CASES = ["same_sign", "aligned", "worst_residual", "binades", "binades_same_sign", "dominant", "ties", "lo_ties",
              "big_plus_small", "gaussian"]


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("case", BF16_CASES)
def test_bf16_filter_margin_bounds_the_product_error(ops, case, d, record_property):
    rs = np.random.RandomState(1000 + d + 7 * BF16_CASES.index(case))
    U, N = 96 + 5, 480 + 3                       # ragged against the 32 x 32 tiles
    P, Q = adversarial_operands(case, U, N, d, rs)
    want = oracle.score_matrix(oracle.SCORE_NORMAL, P, Q)                 # the k-ascending fp32 fmaf chain of every kernel
    for c in (0.0, 40.0):
        prod, margin = ops.test_bf16_products(dev(P), dev(Q), c)
        prod, margin = prod.cpu().numpy(), margin.cpu().numpy()
        assert np.isfinite(prod).all() and np.isfinite(margin).all() and (margin > 0).all()
        err = np.abs(prod.astype(np.float64) - want.astype(np.float64))
        worst = (err / margin[:, None].astype(np.float64)).max()
        assert (err <= margin[:, None]).all(), "case %s d=%d c=%g: error reaches %.3f of the margin" % (case, d, c, worst)
        if c == 0.0:
            # how much of the RELATIVE term the error uses where the norm bound is tight (reported, and kept below 1)
            un = np.sqrt((P.astype(np.float64) ** 2).sum(1)); qn = np.sqrt((Q.astype(np.float64) ** 2).sum(1))
            rel_used = (err / (un[:, None] * qn.max())).max() / (3.2 / 65536 + 8.0 * d / 16777216)
            record_property("rel_used", float(rel_used))
            print("bf16 bound %-18s d=%3d: max err/margin %.3f, of the relative term %.3f" % (case, d, worst, rel_used))
            assert rel_used <= 1.0
    # the margin is not vacuous either: it stays within a small factor of the documented constant times |u| max|q|
    lim = (3.2 / 65536 + 8.0 * d / 16777216 + 1.2e-6) * 1.001 * un * qn.max() * 1.0003 + 1e-6 * 40.0 + 1e-30
    assert (margin <= lim * 1.01).all()


ALL_KINDS = [oracle.SCORE_NORMAL, oracle.SCORE_RUBI_BOTH, oracle.SCORE_RUBI, oracle.SCORE_DIRECT_MINUS, oracle.SCORE_DIRECT_MINUS_BOTH]


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("case", ["same_sign", "aligned", "worst_residual", "binades", "dominant", "ties", "gaussian"])
def test_bf16_listed_scores_stay_within_the_margin(ops, case, d):
    """The listing pass macr_score_topk runs carries the epilogue in its operands (item rows scaled by sig_i, the bias
    -c sig_i as a three-term bf16 slab in the last MFMA, query rows scaled for DIRECT_MINUS_BOTH): for every score kind,
    the score it would list (macr_test_bf16_scores: its own MFMA sequence) stays within the query's margin of the fp32
    score -- the oracle's epilogue on the fmaf chain -- on the adversarial operands of the product bound, with sigmoids
    over thirty binades and c of either sign and size."""
    rs = np.random.RandomState(2000 + d + 11 * BF16_CASES.index(case))
    U, N = 64 + 5, 320 + 3
    P, Q = adversarial_operands(case, U, N, d, rs)
    sig_u = (1.0 / (1.0 + np.exp(-rs.standard_normal(U) * 3.0))).astype(np.float32)
    sig_i = (1.0 / (1.0 + np.exp(-rs.standard_normal(N) * 3.0))).astype(np.float32)
    sig_u[:4] = [1.0, 1e-3, 1e-12, 1e-35]          # (the last one is below the kernels' 1e-30 guard)
    sig_i[:4] = [1.0, 1e-4, 1e-20, 1e-38]
    for kind in ALL_KINDS:
        for c in (0.0, 40.0, -3.5, 1000.0):
            if kind == oracle.SCORE_NORMAL and c != 0.0:
                continue
            want = oracle.score_matrix(kind, P, Q, sig_u, sig_i, c)
            got, margin = ops.test_bf16_scores(kind, dev(P), dev(Q), dev(sig_u), dev(sig_i), c)
            got, margin = got.cpu().numpy(), margin.cpu().numpy()
            assert np.isfinite(got).all() and np.isfinite(margin).all() and (margin > 0).all()
            err = np.abs(got.astype(np.float64) - want.astype(np.float64))
            worst = (err / margin[:, None].astype(np.float64)).max()
            assert (err <= margin[:, None]).all(), "kind %d case %s d=%d c=%g: error reaches %.3f of the margin" % (kind, case, d, c, worst)


@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("case", ["same_sign", "aligned", "worst_residual", "binades", "dominant", "ties", "gaussian"])
def test_f16_listed_scores_stay_within_the_margin(ops, case, d, record_property):
    """The fp16 filter's invariant (eval_kernels.hip filter_margin_h): the score k_score_stream_h would list -- one fp16 number
    per operand, one MFMA per 16 k, the bias -c sig_i (x sig_u for DIRECT_MINUS_BOTH) as six cross terms of two three-term
    fp16 splits in the slab -- stays within the query's margin of the fp32 score, element-wise, for every score kind on the
    adversarial operands of the bf16 bound ("binades": magnitudes down to 2^-20 in one row, i.e. fp16 SUBNORMALS, which the
    matrix core must not flush), sigmoids over thirty binades, c of either sign and size."""
    rs = np.random.RandomState(2500 + d + 11 * BF16_CASES.index(case))
    U, N = 64 + 5, 320 + 3
    P, Q = adversarial_operands(case, U, N, d, rs)
    sig_u = (1.0 / (1.0 + np.exp(-rs.standard_normal(U) * 3.0))).astype(np.float32)
    sig_i = (1.0 / (1.0 + np.exp(-rs.standard_normal(N) * 3.0))).astype(np.float32)
    sig_u[:4] = [1.0, 1e-3, 1e-12, 1e-35]
    sig_i[:4] = [1.0, 1e-4, 1e-20, 1e-38]
    worst_rel = 0.0
    for kind in ALL_KINDS:
        for c in (0.0, 40.0, -3.5, 1000.0):
            if kind == oracle.SCORE_NORMAL and c != 0.0:
                continue
            want = oracle.score_matrix(kind, P, Q, sig_u, sig_i, c)
            got, margin = ops.test_f16_scores(kind, dev(P), dev(Q), dev(sig_u), dev(sig_i), c)
            got, margin = got.cpu().numpy(), margin.cpu().numpy()
            # (RUBI_BOTH: the margin is the accumulator's times sig_u -- 0 for the query whose sig_u = 1e-35 flushes her scores to 0)
            assert np.isfinite(got).all() and np.isfinite(margin).all() and (margin[sig_u > 1e-30] > 0).all()
            err = np.abs(got.astype(np.float64) - want.astype(np.float64))
            worst = (err / margin[:, None].astype(np.float64)).max()
            assert (err <= margin[:, None]).all(), "kind %d case %s d=%d c=%g: error reaches %.3f of the margin" % (kind, case, d, c, worst)
            worst_rel = max(worst_rel, worst)
    record_property("worst_err_over_margin", float(worst_rel))
    print("f16 bound %-16s d=%3d: max err/margin %.3f" % (case, d, worst_rel))
    # the margin is the documented one: (1.001 * 2^-10 + 8 d 2^-24) |u| max|q| + 1e-6 (|u| max|q| + |c|) + 3e-8 (sqrt(d) (|u| + max|q|) + 2)
    un = np.sqrt((P.astype(np.float64) ** 2).sum(1)); qn = np.sqrt((Q.astype(np.float64) ** 2).sum(1)).max()
    _, margin0 = ops.test_f16_scores(oracle.SCORE_NORMAL, dev(P), dev(Q), None, None, 0.0)
    lim = ((1.001 / 1024 + 8.0 * d / 16777216 + 1.0e-6) * un * qn * 1.0003 * 1.0001 + 3.0e-8 * (np.sqrt(d) * (un + qn) * 1.0002 + 2.0) + 1e-30)
    assert (margin0.cpu().numpy() <= lim * 1.001).all()


def test_f16_filter_outside_fp16_range_still_ranks_exactly(ops):
    """An operand beyond fp16's range (|x| > 65504 -- no trained model has one) is clamped in the copy and its row reports an
    infinite norm: that query's margin is infinite, everything is listed for her, the lists overflow and the exact kernel
    ranks her block.  The ranking stays the oracle's, bit for bit: one huge QUERY element (one block pays), then one huge
    ITEM element (every query pays)."""
    rs = np.random.RandomState(77)
    U, N, d, K = 300, 4000, 64, 20
    P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    mask = random_mask(rs, U, N, 10)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    try:
        ops.set_eval_filter("f16")
        for who in ("query", "item"):
            P2, Q2 = P.copy(), Q.copy()
            if who == "query":
                P2[7, 3] = 1.0e5
            else:
                Q2[11, 5] = -2.0e5
            wv, wi, _ = oracle.score_topk(oracle.SCORE_NORMAL, P2, Q2, K, mask=oracle.csr_from_lists(mask))
            v, ix = ops.score_topk(ops.SCORE_NORMAL, dev(P2), None, dev(Q2), K, mask=mcsr, stats=stats)
            val, idx, _ = ops.topk_merge(v, ix)
            assert np.array_equal(idx.cpu().numpy(), wi), who
            assert np.array_equal(val.cpu().numpy().view(np.uint32), wv.view(np.uint32)), who
            assert stats.cpu().numpy()[1] != 0, who           # (the exact kernel did rank: the filter had no bound to offer)
    finally:
        ops.set_eval_filter("env")


@pytest.mark.parametrize("d", [32, 64, 128])
@pytest.mark.parametrize("kind", ALL_KINDS)
def test_every_score_kind_ranks_bit_exact_under_both_filters(ops, kind, d, eval_filter):
    """Every test-time tensor of the reference (model.py:45, :141-142, :199-201) through the fused ranking -- sampled,
    seeded with the previous call's candidates, and as the first round alone -- against the oracle, bit for bit; branch
    sigmoids from 1 down to 1e-12 (sig_u down to 1e-35, below the kernels' guard for 1 / sig_u, for
    DIRECT_MINUS_BOTH), c of either sign.  Under the bf16 filter the
    listing pass carries the epilogue of each kind in its operand copies (k_score_stream_c)."""
    rs = np.random.RandomState(4000 + 17 * kind + d)
    U, N, K = 600, 5000, 20
    P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    sig_u = (1.0 / (1.0 + np.exp(-rs.standard_normal(U) * 2.0))).astype(np.float32)
    sig_i = (1.0 / (1.0 + np.exp(-rs.standard_normal(N) * 2.0))).astype(np.float32)
    # (products of two tiny sigmoids would be denormal scores, which CPU and GPU arithmetic need not treat alike: the user
    # factor goes below the kernels' 1e-30 guard only where it multiplies c * sig_i, not the whole score)
    # (a user factor that scales the WHOLE score below the margin's absolute term makes every item a candidate: the block
    # is then listed again and, in the end, ranked by the exact kernel -- right, but not what this test is about)
    sig_u[:5] = [1.0, 1e-3, 1e-12, 1e-31, 1e-35] if kind == oracle.SCORE_DIRECT_MINUS_BOTH else [1.0, 1e-2, 0.03, 0.1, 0.5]
    sig_i[:5] = [1.0, 1e-4, 1e-8, 1e-12, 0.25]
    mask = random_mask(rs, U, N, 30, heavy=(7,))
    mcsr = ops.CSR.from_lists(mask, "cuda")
    for c in ((0.0,) if kind == oracle.SCORE_NORMAL else (40.0, -2.5)):
        wv, wi, _ = oracle.score_topk(kind, P, Q, K, sig_u, sig_i, c, oracle.csr_from_lists(mask))
        seeds = torch.full((U, ops.SEED_WIDTH), -1, dtype=torch.int32, device="cuda")
        stats = torch.zeros(2, dtype=torch.int32, device="cuda")
        for mode in ("sampled", "seeded", "first round"):
            v, ix = ops.score_topk(kind, dev(P), None, dev(Q), K, dev(sig_u), dev(sig_i), c, mcsr,
                                   seed=None if mode == "sampled" else seeds, seed_out=seeds, stats=stats,
                                   first_round=mode == "first round")
            st = stats.cpu().numpy().tolist()
            assert st[1] == 0, (mode, c, st)                    # the exact fallback kernel is not what ranks here
            if mode == "first round" and st[0] != 0:
                continue                                       # (it says so itself: the complete call is the answer then)
            assert np.array_equal(ix[0].cpu().numpy(), wi), (mode, c)
            assert np.array_equal(v[0].cpu().numpy().view(np.uint32), wv.view(np.uint32)), (mode, c)


def test_bf16_filter_margin_constant_covers_its_stated_bound():
    """filter_rel(d) >= 3.2 * 2^-16 + 6 d * 2^-24 for every supported d (round 3 shipped 1e-4 < 1.39e-4 at d = 256); the
    header's prose quotes the same numbers."""
    src = open(os.path.join(REPO, "macr_amd", "csrc", "eval_kernels.hip")).read()
    assert "return 3.2f / 65536.f + 8.f * (float)d / 16777216.f;" in src
    assert "return 1.001f / 1024.f + 8.f * (float)d / 16777216.f;" in src        # the fp16 filter's: 2 * 2^-11 + second order
    assert "kFilterRel" not in src
    for d in (32, 64, 128, 256):
        assert 3.2 / 65536 + 8.0 * d / 16777216 >= 3.2 / 65536 + 6.0 * d / 16777216 + d / 16777216


# ---------------------------------------------------------------------------------------------------------------------
# --Ks beyond 32 (macr_mf/parse.py:31, utility/parser.py:63 take any list; evaluate_foldout.h:115 any rank_len): the wide
# ranking of macr_score_topk / macr_topk_merge / the metrics, K <= 128 -- dense score rows in blocks + streaming selection
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", [oracle.SCORE_NORMAL, oracle.SCORE_RUBI_BOTH, oracle.SCORE_DIRECT_MINUS])
@pytest.mark.parametrize("U,N,d,K,splits,off", [
    (300, 3000, 64, 50, 1, 0), (257, 2085, 64, 100, 3, 0), (40, 744, 64, 128, 0, 0), (5, 60, 64, 100, 1, 0),
    (64, 1000, 32, 33, 2, 0), (100, 900, 128, 64, 4, 0), (70, 500, 256, 100, 1, 0), (300, 1000, 64, 50, 2, 5000)])
def test_score_topk_wide_bit_exact(ops, kind, U, N, d, K, splits, off, eval_filter):
    rs = np.random.RandomState(U + N + d + K)
    n_users = U + 50
    P = (rs.standard_normal((n_users, d)) * 0.5).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.5).astype(np.float32)
    Q += (rs.standard_normal(N).astype(np.float32) * 0.5)[:, None] * np.sign(P.mean(0, keepdims=True))
    w = (rs.standard_normal(d) * 0.3).astype(np.float32)
    wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    user_ids = rs.permutation(n_users)[:U].astype(np.int32)
    mask_lists = random_mask(rs, U, N, 12, heavy=(0, U - 1) if N > 100 else ())
    mask_lists = [[x + off for x in row] for row in mask_lists]
    mptr, midx = oracle.csr_from_lists(mask_lists)
    c = 3.0
    sig_i_hip = ops.branch_sigmoid(dev(Q), dev(w))
    sig_u_hip = ops.branch_sigmoid(dev(P), dev(wu), dev(user_ids))
    sig_i, sig_u = sig_i_hip.cpu().numpy(), sig_u_hip.cpu().numpy()
    want_v, want_i, want_c = oracle.score_topk(kind, P[user_ids], Q, K, sig_u, sig_i, c, (mptr, midx), off)
    mask = ops.CSR(dev(mptr), dev(midx if len(midx) else np.zeros(1, np.int32)))
    vals, idx = ops.score_topk(kind, dev(P), dev(user_ids), dev(Q), K, sig_u_hip, sig_i_hip, c, mask, off, splits)
    gv, gi, gc = ops.topk_merge(vals, idx)
    assert np.array_equal(gi.cpu().numpy(), want_i)
    assert np.array_equal(gc.cpu().numpy(), want_c)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), want_v.view(np.uint32))      # bit for bit
    fv, fi, fc = ops.topk_merge(vals, idx, fill_mask=mask)
    ov, oi, oc = oracle.score_topk(kind, P[user_ids], Q, K, sig_u, sig_i, c, (mptr, midx), off, fill_masked=True)
    assert np.array_equal(fi.cpu().numpy(), oi)
    # item shards of the wide ranking merge to the unsharded result (what the ranks exchange: K pairs per query and shard)
    cut = [0, N // 3, N // 3 + 1, N]
    parts_v, parts_i = [], []
    for a, b in zip(cut[:-1], cut[1:]):
        v, ix = ops.score_topk(kind, dev(P), dev(user_ids), dev(Q[a:b]), K, sig_u_hip, sig_i_hip[a:b].contiguous(), c, mask, off + a, 1)
        lv, li, _ = ops.topk_merge(v, ix)
        parts_v.append(lv); parts_i.append(li)
    mv, mi, mc = ops.topk_merge(torch.stack(parts_v), torch.stack(parts_i))
    assert np.array_equal(mi.cpu().numpy(), want_i) and np.array_equal(mc.cpu().numpy(), want_c)
    assert np.array_equal(mv.cpu().numpy().view(np.uint32), want_v.view(np.uint32))


@pytest.mark.parametrize("K,W", [(50, 3), (100, 8), (128, 2), (33, 64)])
def test_topk_merge_wide_matches_oracle(ops, K, W):
    rs = np.random.RandomState(K + W)
    U = 37
    vals = np.sort(np.round(rs.standard_normal((W, U, K)).astype(np.float32) * 4) / 4 + np.float32(0), axis=2)[:, :, ::-1].copy()   # ties across lists (no -0.0)
    idxs = np.empty((W, U, K), np.int32)
    for q in range(U):
        ids = rs.permutation(W * K * 2)[: W * K].astype(np.int32).reshape(W, K)
        idxs[:, q, :] = ids
    short = rs.rand(W, U) < 0.3                                  # some lists hold fewer than K candidates
    for s_ in range(W):
        for q in range(U):
            if short[s_, q]:
                n = rs.randint(0, K)
                idxs[s_, q, n:] = -1; vals[s_, q, n:] = -np.inf
    # lists must be sorted by (score desc, id asc) where scores tie: re-sort each
    for s_ in range(W):
        for q in range(U):
            order = np.lexsort((idxs[s_, q], -vals[s_, q]))
            keep = idxs[s_, q][order] >= 0
            order = np.concatenate([order[keep], order[~keep]])
            vals[s_, q] = vals[s_, q][order]; idxs[s_, q] = idxs[s_, q][order]
    wv, wi, wc = oracle.topk_merge(vals, idxs)
    gv, gi, gc = ops.topk_merge(dev(vals), dev(idxs))
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gc.cpu().numpy(), wc)
    assert np.array_equal(gv.cpu().numpy().view(np.uint32), wv.view(np.uint32))


def test_metrics_and_evaluator_with_wide_Ks(ops, eval_filter):
    """Ks = [20, 50, 100] through both evaluator flavours and the c sweep against the oracle's metrics"""
    from macr_amd.evaluator import Evaluator
    rs = np.random.RandomState(77)
    n_users, N, d = 900, 2500, 64
    P = (rs.standard_normal((n_users, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4 + rs.standard_normal((N, 1)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    users = np.sort(rs.choice(n_users, 500, replace=False)).astype(np.int32)
    mask = random_mask(rs, len(users), N, 15)
    gt = [sorted(rs.choice(N, rs.randint(1, 30), replace=False).tolist()) for _ in users]
    Ks, c, kind = [20, 50, 100], 40.0, oracle.SCORE_RUBI_BOTH
    ev = Evaluator(mask, gt, N, torch.device("cuda"))
    uid, Pd, Qd, wd, wud = dev(users), dev(P), dev(Q), dev(w), dev(wu)
    sig_i = ops.branch_sigmoid(Qd, wd).cpu().numpy()
    sig_u = ops.branch_sigmoid(Pd, wud, uid).cpu().numpy()
    mcsr, gcsr = oracle.csr_from_lists(mask), oracle.csr_from_lists(gt)
    for rep in range(2):                                         # second call: the graph replay
        _, oi, oc = oracle.score_topk(kind, P[users], Q, max(Ks), sig_u, sig_i, c, mcsr)
        got = ev.test_mf(kind, Pd, uid, Qd, Ks, wd, wud, c)
        want = oracle.metrics_mf(oi, oc, gcsr, Ks).mean(0)
        for row, k in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
            np.testing.assert_allclose(got[k], want[row], rtol=1e-12, err_msg=k)
        got = ev.test_lgcn(kind, Pd, uid, Qd, Ks, wd, wud, c)
        _, oi, _ = oracle.score_topk(kind, P[users], Q, max(Ks), sig_u, sig_i, c, mcsr, fill_masked=True)
        res = oracle.metrics_foldout(oi, gcsr)
        K = max(Ks)
        res[:, 2 * K:3 * K] = (res[:, K:2 * K] != 0)
        fin = res.astype(np.float64).mean(0).reshape(5, K)[:, np.asarray(sorted(Ks)) - 1]
        np.testing.assert_allclose(got["hr"], fin[2], rtol=1e-6)
        np.testing.assert_allclose(got["ndcg"], fin[3], rtol=1e-6)
    sweep = ev.test_mf_sweep(kind, Pd, uid, Qd, Ks, wd, wud, [0.0, 20.0, 40.0])
    for cc, res_c in zip([0.0, 20.0, 40.0], sweep):
        _, oi, oc = oracle.score_topk(kind, P[users], Q, max(Ks), sig_u, sig_i, cc, mcsr)
        want = oracle.metrics_mf(oi, oc, gcsr, Ks).mean(0)
        np.testing.assert_allclose(res_c["recall"], want[1], rtol=1e-12)


# ----------------------------------------------------------------------------- routing of the split step (macr_shard_route)
@pytest.mark.parametrize("W,layout,B", [(2, "interleaved", 1000), (3, "range", 777), (8, "interleaved", 8192), (16, "range", 4096),
                                        (8, "interleaved", 65536), (1, "interleaved", 300)])
def test_shard_route_is_the_two_stable_sorts_of_the_batch(ops, W, layout, B):
    """macr_shard_route against the definition (RowShardedMF.route's torch version, restated in numpy): counts[q][p], the
    references a rank owns ordered by (destination slice, reference), the references of its slice ordered by (owner,
    reference) -- for every rank, interleaved rows and contiguous ranges, uneven slices, Zipf-skewed items."""
    import ctypes
    from macr_amd import _lib
    rs = np.random.RandomState(W * 1000 + B)
    n_users, n_items = 50000, 7000
    u = rs.randint(0, n_users, B).astype(np.int32)
    i = np.minimum(rs.zipf(1.3, B), n_items).astype(np.int32) - 1
    j = rs.randint(0, n_items, B).astype(np.int32)
    cuts = np.sort(rs.choice(np.arange(1, B), W - 1, replace=False)) if W > 1 else np.zeros(0, np.int64)
    ends = np.concatenate([cuts, [B]]).astype(np.int32)
    def bounds(n):      # contiguous ranges of (almost) equal size
        return np.asarray([(n * (q + 1)) // W for q in range(W)], dtype=np.int32)
    bu, bi = bounds(n_users), bounds(n_items)
    def owner(rows, b):
        return rows % W if layout == "interleaved" else np.searchsorted(b, rows, side="right")
    own = np.concatenate([owner(u, bu), owner(i, bi), owner(j, bi)])
    dest = np.tile(np.searchsorted(ends, np.arange(B), side="right"), 3)
    want_counts = np.bincount(own * W + dest, minlength=W * W).reshape(W, W)
    t = lambda a: torch.from_numpy(a).cuda()
    du, di, dj, dends = t(u), t(i), t(j), t(ends)
    dbu, dbi = (None, None) if layout == "interleaved" else (t(bu), t(bi))
    counts = torch.zeros(W * W, dtype=torch.int32, device="cuda")
    send = torch.full((3 * B,), -1, dtype=torch.int32, device="cuda")
    recv = torch.full((3 * B,), -1, dtype=torch.int32, device="cuda")
    p = ops._ptr
    for rank in range(W):
        send.fill_(-1); recv.fill_(-1)
        _lib.check(_lib.lib().macr_shard_route(B, W, rank, p(du), p(di), p(dj), p(dbu, allow_none=True), p(dbi, allow_none=True),
                                               p(dends), p(counts), p(send), p(recv), ops._stream(0)))
        assert np.array_equal(counts.cpu().numpy().reshape(W, W), want_counts)
        ref = np.arange(3 * B)
        mine = ref[own == rank]
        want_send = mine[np.argsort(dest[own == rank], kind="stable")]
        got_send = send.cpu().numpy()
        assert np.array_equal(got_send[: len(want_send)], want_send) and (got_send[len(want_send):] == -1).all()
        sl = ref[dest == rank]
        want_recv = sl[np.argsort(own[dest == rank], kind="stable")]
        got_recv = recv.cpu().numpy()
        assert np.array_equal(got_recv[: len(want_recv)], want_recv) and (got_recv[len(want_recv):] == -1).all()
