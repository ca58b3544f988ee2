"""The PRODUCT evaluators and CLIs on the GPU (-m gpu): against the reference-generated goldens
(G5-G7), against the oracle end to end on Addressa, and as command lines."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLD, REPO, dataset_args, golden, masked_scores, score_matrix
from macr_amd.data import LGCNData, MFData

pytestmark = pytest.mark.gpu


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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ----------------------------------------------------------------------------- goldens through the HIP kernels
@pytest.mark.parametrize("dataset", ["addressa", "tiny"])
def test_G5_mf_evaluator_on_hip(ops, dataset):
    """macr_mf/train.py test() outputs (G5) reproduced by macr_topk_scores + macr_metrics_mf + macr_colmean."""
    g = golden("mf", dataset)["G5"]
    data = MFData(dataset_args(dataset))
    users = list(data.test_user_list.keys())
    mask, gt = data.eval_lists(users)
    gtc = ops.CSR.from_lists(gt, "cuda")
    Kmax = max(g["Ks"])
    cnt = dev(np.asarray([min(Kmax, data.n_items - len(set(m))) for m in mask], np.int32))
    for kind, seed in g["seeds"].items():
        full = score_matrix(kind, data.n_users, data.n_items, seed)
        s = masked_scores(full[np.asarray(users)], mask)         # candidates = all items - train items
        idx, _ = ops.topk_scores(dev(s), Kmax)
        m = ops.colmean(ops.metrics_mf(idx, cnt, gtc, g["Ks"])).cpu().numpy()
        want = g["results"]["%s/o" % kind]
        for row, k in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
            np.testing.assert_allclose(m[row], want[k], rtol=1e-12, atol=1e-15, err_msg="%s %s" % (kind, k))


@pytest.mark.parametrize("dataset", ["addressa", "tiny"])
def test_G6_lgcn_evaluator_on_hip(ops, dataset):
    """utility/batch_test.py test() outputs (G6) via the drop-in eval_score_matrix_foldout path."""
    from macr_amd.evaluator import eval_score_matrix_foldout
    g = golden("lgcn", dataset)["G6"]
    a = dataset_args(dataset)
    dg = LGCNData(path=a.data_path + a.dataset, batch_size=a.batch_size, args=a)
    users = list(dg.test_set.keys())
    mask, gt = dg.eval_lists(users)
    top_show = np.sort(np.asarray(g["Ks"]))
    max_top = int(top_show.max())
    for kind, seed in g["seeds"].items():
        if kind == "ties":
            continue
        full = score_matrix(kind, dg.n_users, dg.n_items, seed)
        res = eval_score_matrix_foldout(masked_scores(full[np.asarray(users)], mask), gt, max_top)
        assert res.dtype == np.float32 and res.shape == (len(users), 5 * max_top)
        res[:, 2 * max_top:3 * max_top] = (res[:, max_top:2 * max_top] != 0)
        final = res.astype(np.float64).mean(0).reshape(5, max_top)[:, top_show - 1]
        want = g["results"]["%s/normal" % kind]
        np.testing.assert_allclose(final[2], want["hr"], rtol=2e-6)
        np.testing.assert_allclose(final[1], want["recall"], rtol=2e-6)
        np.testing.assert_allclose(final[3], want["ndcg"], rtol=2e-6)


# ----------------------------------------------------------------------------- fused path, end to end on Addressa
@pytest.mark.parametrize("kind", [0, 1])
def test_fused_evaluators_match_oracle_on_addressa(ops, kind, eval_filter):
    from macr_amd.evaluator import Evaluator
    data = MFData(dataset_args("addressa"))
    users = list(data.test_user_list.keys())
    mask, gt = data.eval_lists(users)
    rs = np.random.RandomState(5)
    d = 64
    P = (rs.standard_normal((data.n_users, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((data.n_items, d)) * 0.4 + rs.standard_normal((data.n_items, 1)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    c, Ks = 40.0, [5, 20]
    ev = Evaluator(mask, gt, data.n_items, torch.device("cuda"))
    uid = dev(np.asarray(users, np.int32))
    Pd, Qd, wd, wud = dev(P), dev(Q), dev(w), dev(wu)
    sig_i = ops.branch_sigmoid(Qd, wd).cpu().numpy()
    sig_u = ops.branch_sigmoid(Pd, wud, uid).cpu().numpy()
    mcsr, gcsr = oracle.csr_from_lists(mask), oracle.csr_from_lists(gt)
    _, oi, oc = oracle.score_topk(kind, P[users], Q, max(Ks), sig_u, sig_i, c, mcsr)
    got = ev.test_mf(kind, Pd, uid, Qd, Ks, wd, wud, c)
    want = oracle.metrics_mf(oi, oc, gcsr, Ks).mean(0)
    for row, k in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
        np.testing.assert_allclose(got[k], want[row], rtol=1e-12, err_msg=k)         # HR/NDCG@20 within 1e-4 (north star)
    got = ev.test_lgcn(kind, Pd, uid, Qd, Ks, wd, wud, c)
    _, oi, _ = oracle.score_topk(kind, P[users], Q, max(Ks), sig_u, sig_i, c, mcsr, fill_masked=True)
    res = oracle.metrics_foldout(oi, gcsr)
    K = max(Ks)
    res[:, 2 * K:3 * K] = (res[:, K:2 * K] != 0)
    fin = res.astype(np.float64).mean(0).reshape(5, K)[:, np.asarray(sorted(Ks)) - 1]
    np.testing.assert_allclose(got["hr"], fin[2], rtol=1e-6)
    np.testing.assert_allclose(got["recall"], fin[1], rtol=1e-6)
    np.testing.assert_allclose(got["ndcg"], fin[3], rtol=1e-6)
    # queries ranked in chunks (bounded workspace) give the same rankings
    whole = ev.rank(kind, Pd, uid, Qd, max(Ks), wd, wud, c)
    ev.max_queries_per_pass = 1000
    parts = ev.rank(kind, Pd, uid, Qd, max(Ks), wd, wud, c)
    for a, b in zip(whole, parts):
        assert torch.equal(a, b)


def test_session_shim_matches_fast_path(ops):
    """sess.run(fetches, feed_dict) (the reference's call boundary) == the direct train_step / ratings calls."""
    import types
    from macr_amd.mf import BPRMF, Session
    args = types.SimpleNamespace(regs=1e-5, embed_size=64, lr=1e-3, batch_size=256, verbose=0, c=40.0, alpha=1e-2, beta=1e-3)
    cfg = dict(n_users=900, n_items=300)
    a, b = BPRMF(args, cfg, seed=7), BPRMF(args, cfg, seed=7)
    sess = Session(a)
    rs = np.random.RandomState(0)
    u = rs.choice(900, 256, replace=False).tolist(); i = rs.randint(0, 300, 256).tolist(); j = rs.randint(0, 300, 256).tolist()
    for name in ("two_bce_both", "bce"):
        fetch = [getattr(a, "opt_" + name), getattr(a, "loss_" + name), getattr(a, "mf_loss_" + name),
                 getattr(a, "reg_loss_" + name)]
        _, loss, mf, reg = sess.run(fetch, feed_dict={a.users: u, a.pos_items: i, a.neg_items: j})
        kind = getattr(b, "opt_" + name).kind
        direct = b.train_step(kind, b.to_device_batch(u, i, j)).cpu().numpy()
        np.testing.assert_allclose([loss, mf, reg], direct, rtol=1e-6)
    a.update_c(sess, 40.0); b.update_c(None, 40.0)
    S = sess.run(a.rubi_ratings_both, {a.users: u[:50], a.pos_items: list(range(300))})
    assert S.shape == (50, 300) and S.dtype == np.float32
    np.testing.assert_allclose(S, b.ratings(ops.SCORE_RUBI_BOTH, u[:50]).cpu().numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(NotImplementedError):
        sess.run(a.opt_two, {a.users: u, a.pos_items: i, a.neg_items: j})


# ----------------------------------------------------------------------------- device sampler
def test_device_sampler_distribution(ops):
    from macr_amd.sampler import DeviceSampler
    rs = np.random.RandomState(0)
    n_users, n_items, B = 500, 200, 256
    train = {u: sorted(rs.choice(n_items, size=rs.randint(1, 40), replace=False).tolist()) for u in range(n_users)}
    train[7] = []                                   # empty list -> positive 0 (load_data.py:551-552)
    train[9] = list(range(n_items - 1))             # one admissible negative only
    smp = DeviceSampler(train, n_users, n_items, B, torch.device("cuda"), seed=3)
    seen_users = np.zeros(n_users)
    neg_hist = np.zeros(n_items)
    for step in range(200):
        u, i, j = smp.sample().cpu().numpy()
        assert len(set(u.tolist())) == B            # rd.sample: without replacement inside a batch
        for uu, ii, jj in zip(u, i, j):
            assert jj not in train[uu]
            assert (ii in train[uu]) or (train[uu] == [] and ii == 0)
        seen_users[u] += 1
        neg_hist[j] += 1
    # every user is drawn with probability B/n_users per batch
    exp = 200 * B / n_users
    assert abs(seen_users.mean() - exp) < 1e-9 and seen_users.std() < 4 * np.sqrt(exp)
    assert neg_hist.min() > 0
    a = DeviceSampler(train, n_users, n_items, B, torch.device("cuda"), seed=3).sample().cpu().numpy()
    b = DeviceSampler(train, n_users, n_items, B, torch.device("cuda"), seed=3).sample().cpu().numpy()
    c = DeviceSampler(train, n_users, n_items, B, torch.device("cuda"), seed=4).sample().cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c)       # pure function of (seed, step)
    # batches drawn ahead, 32 or 5 per launch (macr_sample_triples_many), are the batches of one launch per step
    one = DeviceSampler(train, n_users, n_items, B, torch.device("cuda"), seed=3, ahead=1)
    many = DeviceSampler(train, n_users, n_items, B, torch.device("cuda"), seed=3)
    five = DeviceSampler(train, n_users, n_items, B, torch.device("cuda"), seed=3, ahead=5)
    into = torch.empty((3, B), dtype=torch.int32, device="cuda")
    for step in range(70):
        want = one.sample().clone()
        assert torch.equal(many.sample(), want) and torch.equal(five.sample(), want), step
        if step % 9 == 0:                           # a caller-owned buffer takes a launch of its own, same stream of batches
            assert torch.equal(one.sample(out=into), many.sample()) and five.sample() is not None
    # LightGCN's sample_test: positives from the TEST lists, negatives outside test and train lists
    test_l = {u: sorted(rs.choice(n_items, size=3, replace=False).tolist()) for u in range(0, n_users, 3)}
    both = {u: test_l[u] + train[u] for u in test_l}
    ts = DeviceSampler(test_l, n_users, n_items, 128, torch.device("cuda"), seed=8, pool=sorted(test_l), exclude=both)
    for step in range(40):
        u, i, j = ts.sample().cpu().numpy()
        for uu, ii, jj in zip(u, i, j):
            assert uu in test_l and ii in test_l[uu] and jj not in test_l[uu] and jj not in train[uu]
    big = DeviceSampler(train, n_users, n_items, 2048, torch.device("cuda"), seed=1, pool=list(range(0, 500, 2)))
    u, i, j = big.sample().cpu().numpy()           # B > pool: with replacement, only pool users
    assert set(u.tolist()) <= set(range(0, 500, 2)) and len(u) == 2048


def test_device_sampler_equals_its_oracle_bit_for_bit(ops):
    """k_sample_triples against oracle.sample_triples (oracle/macr_oracle.c::orc_sample_triples: splitmix64 draws, the
    4-round Feistel permutation with cycle walking, multiply-shift ranges, rejection + binary search) -- integer work, so
    equality is exact; the oracle's LAW is checked against the reference samplers in tests/test_sampler_law.py."""
    import oracle
    from macr_amd.sampler import DeviceSampler
    rs = np.random.RandomState(5)
    dev = torch.device("cuda")
    for n_users, n_items, B in [(500, 200, 256), (500, 200, 500), (37, 1000, 37), (300, 64, 1024), (70000, 9000, 8192)]:
        train = {u: sorted(rs.choice(n_items, size=rs.randint(0, min(40, n_items // 2)), replace=False).tolist())
                 for u in range(n_users)}
        train[1] = list(range(n_items - 1))
        csr = oracle.csr_from_lists([train[u] for u in range(n_users)])
        for ahead in (1, 7):
            smp = DeviceSampler(train, n_users, n_items, B, dev, seed=77 + ahead, ahead=ahead)
            for step in range(10):
                got = smp.sample().cpu().numpy()
                assert np.array_equal(got, oracle.sample_triples(77 + ahead, step, B, n_items, csr)), (n_users, B, ahead, step)
    # a user pool (LightGCN's exist_users) and an exclusion list (sample_test)
    n_users, n_items, B = 400, 150, 128
    train = {u: sorted(rs.choice(n_items, size=rs.randint(1, 30), replace=False).tolist()) for u in range(n_users)}
    test_l = {u: sorted(rs.choice(np.setdiff1d(np.arange(n_items), train[u]), size=3, replace=False).tolist())
              for u in range(0, n_users, 3)}
    both = {u: sorted(set(test_l[u]) | set(train[u])) for u in test_l}
    pool = sorted(test_l)
    ts = DeviceSampler(test_l, n_users, n_items, B, dev, seed=8, pool=pool, exclude=both)
    tl = oracle.csr_from_lists([test_l.get(u, []) for u in range(n_users)])
    ex = oracle.csr_from_lists([both.get(u, []) for u in range(n_users)])
    for step in range(40):
        want = oracle.sample_triples(8, step, B, n_items, tl, pool=np.asarray(pool, np.int32), exclude=ex)
        assert np.array_equal(ts.sample().cpu().numpy(), want), step
    ps = DeviceSampler(train, n_users, n_items, 300, dev, seed=9, pool=list(range(0, n_users, 2)))   # B > pool: with replacement
    tr = oracle.csr_from_lists([train[u] for u in range(n_users)])
    for step in range(5):
        want = oracle.sample_triples(9, step, 300, n_items, tr, pool=np.arange(0, n_users, 2, dtype=np.int32))
        assert np.array_equal(ps.sample().cpu().numpy(), want), step


# ----------------------------------------------------------------------------- command lines
def _run_cli(cmd, cwd):
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    out = subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


@pytest.mark.parametrize("train,test_,sampler", [("normalbce", "normal", "reference"), ("rubibceboth", "rubi", "reference"),
                                                  ("rubibceboth", "rubi", "device")])
def test_mf_cli_addressa(tmp_path, train, test_, sampler):
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    out = _run_cli([os.path.join(REPO, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size", "1024",
                    "--cuda", "0", "--saveID", "t", "--log_interval", "2", "--lr", "0.001", "--epoch", "4",
                    "--train", train, "--test", test_, "--c", "40", "--alpha", "1e-3", "--beta", "1e-3",
                    "--sampler", sampler], str(tmp_path))
    lines = [l for l in out.splitlines() if "train==[" in l]
    assert len(lines) == 4, out
    evals = [l for l in lines if "hit=[" in l]
    assert len(evals) == 2 and all(("Epoch" in l) or l.startswith("c:40.00") for l in evals)
    hit = float(evals[-1].split("hit=[")[1].split(",")[0])
    assert 0.0 < hit < 1.0
    assert os.path.exists(tmp_path / ("mf_addressa_checkpoint/wd_1e-05_lr_0.001_t/3_ckpt.pt"))


def test_lightgcn_cli_addressa(tmp_path):
    out = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN.py"), "--data_path", os.path.join(REPO, "data") + "/",
                    "--dataset", "addressa", "--verbose", "1", "--layer_size", "[64,64]", "--Ks", "[20]", "--loss",
                    "bceboth", "--test", "rubiboth", "--c", "40", "--epoch", "4", "--early_stop", "1", "--lr", "0.001",
                    "--batch_size", "1024", "--gpu_id", "0", "--log_interval", "2", "--alpha", "1e-2", "--beta", "1e-3",
                    "--weights_path", str(tmp_path) + "/"], str(tmp_path))
    assert out.count("train==[") == 2 and out.count("c:40.00 recall=[") == 2, out
    hit = float(out.split("hit=[")[-1].split(",")[0])
    assert 0.0 < hit < 1.0
    assert not [f for f in os.listdir(os.path.join(REPO, "data", "addressa")) if f.endswith(".npz")]


def test_clis_abort_on_nan_loss(tmp_path):
    """the reference stops a run whose epoch loss is NaN (`ERROR: loss is nan.` + sys.exit(): macr_mf/train.py:500-502,
    macr_lightgcn/LightGCN.py:783-785).  A learning rate of 1e30 overflows the tables within two steps (inf - inf in the
    next gradients): both CLIs must print the line, stop before --epoch is reached and exit with status 0 like sys.exit()."""
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    out = _run_cli([os.path.join(REPO, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size", "1024",
                    "--cuda", "0", "--saveID", "n", "--log_interval", "1", "--lr", "1e30", "--epoch", "6", "--verbose", "1",
                    "--train", "rubibceboth", "--test", "rubi", "--c", "40", "--save_flag", "0"], str(tmp_path))
    assert "ERROR: loss is nan." in out, out[-2000:]
    assert out.count("train==[") < 6, out[-2000:]
    out = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN.py"), "--data_path", os.path.join(REPO, "data") + "/",
                    "--dataset", "addressa", "--verbose", "1", "--layer_size", "[64,64]", "--Ks", "[20]", "--loss", "bceboth",
                    "--test", "rubiboth", "--c", "40", "--epoch", "6", "--lr", "1e30", "--batch_size", "1024", "--gpu_id", "0",
                    "--log_interval", "1", "--save_flag", "0", "--weights_path", str(tmp_path) + "/"], str(tmp_path))
    assert "ERROR: loss is nan." in out, out[-2000:]
    assert out.count("train==[") < 6, out[-2000:]


def test_cli_test_finds_its_evaluator_by_list_identity(tmp_path):
    """macr_mf/train.py::test() is stateless like the reference's (:162): any list may arrive.  The evaluator is built once per
    list; a second call with the SAME list object must not hash the list again, an equal list in a new object is found by
    content, a different list gets its own evaluator -- and all of them rank like a fresh evaluator."""
    code = r'''
import sys, os
os.chdir("%s"); sys.argv = ["train.py", "--dataset", "addressa", "--batch_size", "1024", "--train", "rubibceboth", "--test", "rubi"]
sys.path.insert(0, "%s")
import numpy as np, torch
import train as cli
from model import BPRMF, Session
model = BPRMF(cli.args, dict(n_users=cli.data.n_users, n_items=cli.data.n_items), seed=3)
sess = Session(model)
users = list(cli.data.test_user_list.keys())
r1 = cli.test(sess, model, users, model_type="rubi_both")
n0 = cli._evaluators.content_lookups
r2 = cli.test(sess, model, users, model_type="rubi_both")
assert cli._evaluators.content_lookups == n0, "same list object: no content hash"
r3 = cli.test(sess, model, list(users), model_type="rubi_both")
assert cli._evaluators.content_lookups == n0 + 1 and len(cli._evaluators) == 1
half = users[: len(users) // 2]
r4 = cli.test(sess, model, half, model_type="rubi_both")
assert len(cli._evaluators) == 2
users2 = list(users); users2[0], users2[1] = users2[1], users2[0]       # same users, another order: another list
r5 = cli.test(sess, model, users2, model_type="rubi_both")
for k in r1:
    assert np.array_equal(r1[k], r2[k]) and np.array_equal(r1[k], r3[k])
    assert np.allclose(r1[k], r5[k], rtol=1e-12)
assert not np.array_equal(r1["recall"], r4["recall"])
print("OK", float(r1["hit_ratio"][0]))
''' % (str(tmp_path), os.path.join(REPO, "macr_mf"))
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_clis_take_Ks_beyond_32(tmp_path):
    """--Ks "[20,50,100]" (the reference takes any list: macr_mf/parse.py:31, utility/parser.py:63) through both CLIs"""
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    out = _run_cli([os.path.join(REPO, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size", "1024",
                    "--cuda", "0", "--saveID", "k", "--log_interval", "2", "--lr", "0.001", "--epoch", "2",
                    "--train", "rubibceboth", "--test", "rubi", "--c", "40", "--Ks", "[20,50,100]", "--save_flag", "0"],
                   str(tmp_path))
    ev = [l for l in out.splitlines() if "hit=[" in l][-1]
    hits = [float(x) for x in ev.split("hit=[")[1].split("]")[0].split(",")]        # (the log shows the first and the last K: train.py:532)
    assert len(hits) == 2 and 0.0 < hits[0] < hits[1] <= 1.0, ev
    out = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN.py"), "--data_path", os.path.join(REPO, "data") + "/",
                    "--dataset", "addressa", "--verbose", "1", "--layer_size", "[64,64]", "--Ks", "[20,50,100]", "--loss",
                    "bceboth", "--test", "rubiboth", "--c", "40", "--epoch", "2", "--lr", "0.001", "--batch_size", "1024",
                    "--gpu_id", "0", "--log_interval", "2", "--weights_path", str(tmp_path) + "/"], str(tmp_path))
    ev = [l for l in out.splitlines() if "hit=[" in l][-1]
    hits = [float(x) for x in ev.split("hit=[")[1].split("]")[0].replace(",", " ").split()]
    assert len(hits) in (2, 3) and 0.0 < hits[0] < hits[-1] <= 1.0, ev


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """bench.py's N>1 path (replica training, rank-0 broadcast, item-sharded evaluation, all-gather, merge, the
    configs[4] leg) with two ranks on ONE GPU through the gloo test rig, started EXACTLY as the driver starts it --
    `python bench.py --gpus 2 --steps 20 --warmup 5`, no launcher: bench.py becomes its own torch.distributed.run.
    The single-rank run is the reference for the eval metrics."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--workload", "addressa", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--eval-reps", "1",
              "--regions", "1", "--no-e2e", "--eval-train-steps", "0"]   # same training steps in both runs: same model
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=root, capture_output=True, text=True,
                         timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, MACR_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + common +
                         ["--c4-users", "200000", "--c4-items", "50000", "--c4-eval-users", "2000"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-1000:] + two.stderr[-3000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["value"] > 0 and b["config"]["global_batch"] == 2 * a["config"]["global_batch"]
    assert b["nranks"] == 2 and b["multi_gpu"]["collectives_us"]["eval_all_gather_topk"] > 0
    c4 = b["config4"]
    assert c4["scaling"] == "strong" and c4["value"] > 0 and c4["eval_users_per_s"] > 0 and c4["collectives_ms"]
    assert a["nranks"] == 1 and "config4" not in a
    for k, v in a["eval_metrics"].items():            # same model (rank 0's), item-sharded: same ranking metrics
        assert abs(b["eval_metrics"][k] - v) <= 2e-3 * max(abs(v), 1e-3), (k, v, b["eval_metrics"][k])


def test_repair_round_continues_on_the_first_rounds_workspace(ops):
    """the repair graph of an evaluation is captured long after its first-round graph -- at the first
    evaluation whose seeds went stale -- and the shared per-device ranking workspace may have been regrown by a larger
    evaluator in between.  The repair round must run on the workspace the first round was captured with (thresholds,
    overflow counters and candidate lists live there), not on whatever the cache holds now."""
    from macr_amd.evaluator import Evaluator
    rs = np.random.RandomState(17)
    d, n_items, n_users = 64, 9000, 30000
    P = dev((rs.standard_normal((n_users, d)) * 0.4).astype(np.float32))
    Q = dev((rs.standard_normal((n_items, d)) * 0.4).astype(np.float32))
    w, wu = dev((rs.standard_normal(d) * 0.3).astype(np.float32)), dev((rs.standard_normal(d) * 0.3).astype(np.float32))
    device = torch.device("cuda", torch.cuda.current_device())

    def mk(U, seed):
        r = np.random.RandomState(seed)
        users = np.sort(r.choice(n_users, U, replace=False)).astype(np.int32)
        mask = [sorted(r.choice(n_items, 20, replace=False).tolist()) for _ in range(U)]
        gt = [sorted(r.choice(n_items, 5, replace=False).tolist()) for _ in range(U)]
        return Evaluator(mask, gt, n_items, device), dev(users)

    def direct(ev, uid):
        ev2 = Evaluator(ev._mask_lists, [[0]] * ev.n_queries, n_items, device)
        ev2.gt, ev2.use_graph, ev2.use_seeds = ev.gt, False, False
        return ev2.test_mf(1, P, uid, Q, [20], w, wu, 40.0)

    ops._topk_ws_cache.pop(device, None)        # (earlier tests may have grown the cache past what `big` below asks for)
    small, uid_s = mk(700, 1)
    if not small._shape_uses_seeds(n_items, d):
        pytest.skip("this shape lists every item: no seeds, no repair round")
    small.test_mf(1, P, uid_s, Q, [20], w, wu, 40.0)                  # sampled first round: captured, leaves seeds
    small.test_mf(1, P, uid_s, Q, [20], w, wu, 40.0)                  # seeded first round: captured
    assert small.last_eval_info()["seeded"] and not small.last_eval_info()["redone"]
    ws_small = ops._topk_ws_cache[device]
    big, uid_b = mk(20000, 2)
    big.test_mf(1, P, uid_b, Q, [20], w, wu, 40.0)                    # regrows the cached workspace
    assert ops._topk_ws_cache[device] is not ws_small
    ops._topk_ws_cache[device].fill_(0x5A)                            # whatever the cache holds now is not small's state
    Q.neg_()                                                          # the model moves away from every seed
    got = small.test_mf(1, P, uid_s, Q, [20], w, wu, 40.0)            # seeded first round fails -> repair graph captured NOW
    info = small.last_eval_info()
    assert info["seeded"] and info["redone"] and info["query_blocks_relisted"] > 0, info
    want = direct(small, uid_s)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-12, err_msg=k)
    Q.neg_()


def test_evaluator_graphs_survive_scratch_reallocation(ops):
    """Evaluations are replayed as HIP graphs that bake in the addresses of the shared ranking workspace and of the
    cached mask bitmaps; a later, larger evaluator makes the caches allocate new buffers.  The first evaluator's
    graph must keep working on its own (kept-alive) scratch, and in-place table updates must show in the replays."""
    from macr_amd.evaluator import Evaluator
    rs = np.random.RandomState(3)
    d, n_items = 64, 3000
    P = dev((rs.standard_normal((6000, d)) * 0.4).astype(np.float32))
    Q = dev((rs.standard_normal((n_items, d)) * 0.4).astype(np.float32))
    w, wu = dev((rs.standard_normal(d) * 0.3).astype(np.float32)), dev((rs.standard_normal(d) * 0.3).astype(np.float32))

    def mk(U, seed):
        r = np.random.RandomState(seed)
        users = np.sort(r.choice(6000, U, replace=False)).astype(np.int32)
        mask = [sorted(r.choice(n_items, 20, replace=False).tolist()) for _ in range(U)]
        gt = [sorted(r.choice(n_items, 5, replace=False).tolist()) for _ in range(U)]
        return Evaluator(mask, gt, n_items, torch.device("cuda", torch.cuda.current_device())), dev(users)

    def direct(ev, uid):
        ev2 = Evaluator.__new__(Evaluator); ev2.__dict__.update(ev.__dict__); ev2.use_graph = False
        return ev2.test_mf(1, P, uid, Q, [20], w, wu, 40.0)

    small, uid_s = mk(300, 1)
    a1 = small.test_mf(1, P, uid_s, Q, [20], w, wu, 40.0)            # captured
    big, uid_b = mk(5000, 2)
    b1 = big.test_mf(1, P, uid_b, Q, [20], w, wu, 40.0)              # larger workspace replaces the cached one
    junk = [torch.full((1 << 20,), 7.0, device="cuda") for _ in range(8)]   # reuse whatever was freed
    a2 = small.test_mf(1, P, uid_s, Q, [20], w, wu, 40.0)            # replay of the first graph
    for k in a1:
        assert np.array_equal(a1[k], a2[k]), k
        assert np.array_equal(a1[k], direct(small, uid_s)[k]), k
        assert np.array_equal(b1[k], direct(big, uid_b)[k]), k
    Q.mul_(-1.0)                                                     # tables change in place between evaluations
    a3 = small.test_mf(1, P, uid_s, Q, [20], w, wu, 40.0)
    want = direct(small, uid_s)
    for k in a3:
        assert np.array_equal(a3[k], want[k]), k
    assert any(not np.array_equal(a3[k], a1[k]) for k in a3)
    del junk


def test_cli_evaluations_identical_with_and_without_graph_replay(tmp_path):
    """Three evaluations of a training run, replayed as a HIP graph vs launched directly: the printed metrics must be
    identical.  (Regression: hipMemsetAsync nodes of a captured graph lost their effect from the second replay on --
    tools/graph_memset_check.py -- and later evaluations ranked against stale thresholds.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.symlink(os.path.join(root, "data"), tmp_path / "data")
    logs = []
    for graph in ("1", "0"):
        env = dict(os.environ, MACR_EVAL_GRAPH=graph)
        r = subprocess.run([sys.executable, os.path.join(root, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size",
                            "1024", "--cuda", "0", "--saveID", "97", "--log_interval", "10", "--lr", "0.001", "--train",
                            "normalbce", "--test", "normal", "--epoch", "30"],
                           cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if "hit=" in l]
        assert len(lines) == 3, r.stdout[-2000:]
        logs.append([[float(l.split(k + "=[")[1].split(",")[0]) for k in ("recall", "hit", "ndcg")] for l in lines])
    # two training runs differ in the last bits (float atomics), so rankings may flip for a handful of users; the
    # regression moved the second and third evaluation by 2.5-6 %
    np.testing.assert_allclose(np.asarray(logs[0]), np.asarray(logs[1]), rtol=5e-3)


# ----------------------------------------------------------------------------- (f) rows: tuner, checkpoints, variants
def test_c_sweep_replays_one_graph_and_equals_separate_evaluations(ops):
    """tune.py:545-578 / LightGCN_tune.py:852-870: every c of a sweep goes through ONE evaluator.  c lives in a device
    scalar the kernels read at run time, so the sweep replays a single captured graph; each c must give exactly what
    a fresh, graph-free evaluation of that c gives."""
    from macr_amd.evaluator import Evaluator
    rs = np.random.RandomState(5)
    d, n_users, n_items, U = 64, 4000, 3000, 1500
    P = dev((rs.standard_normal((n_users, d)) * 0.4).astype(np.float32))
    Q = dev((rs.standard_normal((n_items, d)) * 0.4).astype(np.float32))
    w, wu = dev((rs.standard_normal(d) * 0.3).astype(np.float32)), dev((rs.standard_normal(d) * 0.3).astype(np.float32))
    users = np.sort(rs.choice(n_users, U, replace=False)).astype(np.int32)
    mask = [sorted(rs.choice(n_items, 20, replace=False).tolist()) for _ in range(U)]
    gt = [sorted(rs.choice(n_items, 5, replace=False).tolist()) for _ in range(U)]
    uid = dev(users)
    ev = Evaluator(mask, gt, n_items, torch.device("cuda"))
    sweep = {}
    for c in np.linspace(-5.0, 40.0, 10):
        sweep[float(c)] = ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [5, 20], w, wu, float(c))
    # one capture per launch sequence serves all ten values: the first ranking samples its thresholds, the others seed them
    # with the previous value's best candidates (or sample again when those went stale)
    # (optimistic mode adds the complete sampled sequence for a value whose seeds were stale)
    assert ev.use_graph and len(ev._graphs) <= 3 and ev._graph_misses <= 3
    for c, got in sweep.items():
        fresh = Evaluator(mask, gt, n_items, torch.device("cuda"))
        fresh.use_graph = False
        want = fresh.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [5, 20], w, wu, c)
        for k in want:
            assert np.array_equal(got[k], want[k]), (c, k)
    vals = [tuple(v["hit_ratio"]) for v in sweep.values()]
    assert len(set(vals)) > 1                                     # c does change the ranking


def _metric_tail(line):
    return line.split("recall=[")[1]


def test_mf_cli_tune_pretrain_resume(tmp_path):
    """f1/f3: tune.py sweeps c; checkpoints written by train.py are read back by --pretrain 1 (evaluate, exit) and by
    the additive --resume 1 (continue training); a reloaded model evaluates to the digits it was saved with."""
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    common = ["--dataset", "addressa", "--batch_size", "1024", "--cuda", "0", "--saveID", "r", "--log_interval", "2",
              "--lr", "0.001", "--train", "rubibceboth", "--test", "rubi", "--alpha", "1e-3", "--beta", "1e-3"]
    out = _run_cli([os.path.join(REPO, "macr_mf", "tune.py")] + common + ["--epoch", "4", "--start", "20", "--end", "40",
                   "--step", "3"], str(tmp_path))
    lines = [l for l in out.splitlines() if l.startswith("c:")]
    assert [l[:7] for l in lines] == ["c:20.00", "c:30.00", "c:40.00"] * 2, out
    last40 = lines[-1]
    ck = tmp_path / "mf_addressa_checkpoint/wd_1e-05_lr_0.001_r"
    assert os.path.exists(ck / "1_ckpt.pt") and os.path.exists(ck / "3_ckpt.pt")
    out2 = _run_cli([os.path.join(REPO, "macr_mf", "train.py")] + common + ["--pretrain", "1", "--c", "40"], str(tmp_path))
    got = [l for l in out2.splitlines() if l.startswith("epoch 3 c:40.00")]
    assert len(got) == 1 and _metric_tail(got[0]) == _metric_tail(last40), (got, last40)
    out3 = _run_cli([os.path.join(REPO, "macr_mf", "train.py")] + common + ["--resume", "1", "--epoch", "6", "--c", "40"],
                    str(tmp_path))
    assert "resumed from epoch 3" in out3
    assert [l.split()[1] for l in out3.splitlines() if l.startswith("Epoch ")] == ["4", "5"], out3
    assert os.path.exists(ck / "5_ckpt.pt")


def test_mf_cli_rubibce(tmp_path):
    """--train rubibce --test rubi: opt_two_bce (model.py:67-69,:158-183) scored with rubi_ratings (:141, test()
    model_type 'rubi_c', train.py:551)."""
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    out = _run_cli([os.path.join(REPO, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size", "1024", "--cuda", "0",
                    "--saveID", "b", "--log_interval", "2", "--lr", "0.001", "--epoch", "2", "--train", "rubibce", "--test",
                    "rubi", "--c", "30", "--alpha", "1e-3", "--save_flag", "0"], str(tmp_path))
    line = [l for l in out.splitlines() if l.startswith("c:30.00")]
    assert len(line) == 1 and 0.0 < float(line[0].split("hit=[")[1].split(",")[0]) < 1.0, out


def test_lightgcn_cli_test_loss_pretrain(tmp_path):
    """--test normal prints the reference's test-loss columns (LightGCN.py:799-819) -- finite numbers from the
    loss-only pass -- and --pretrain 1 reads the saved weights back (:707-719)."""
    common = ["--data_path", os.path.join(REPO, "data") + "/", "--dataset", "addressa", "--verbose", "1", "--layer_size",
              "[64,64]", "--Ks", "[20]", "--lr", "0.001", "--batch_size", "1024", "--gpu_id", "0", "--log_interval", "2",
              "--alpha", "1e-2", "--beta", "1e-3", "--weights_path", str(tmp_path) + "/", "--saveID", "q"]
    out = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN.py")] + common + ["--loss", "bce", "--test", "normal",
                   "--epoch", "2"], str(tmp_path))
    line = [l for l in out.splitlines() if "test==[" in l]
    assert len(line) == 1, out
    nums = line[0].split("test==[")[1].split("]")[0].replace("=", "+").split("+")
    vals = [float(x) for x in nums]
    assert all(np.isfinite(vals)) and vals[0] > 0 and abs(vals[0] - (vals[1] + vals[2])) < 1e-4, line
    out2 = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN.py")] + common + ["--loss", "bce", "--test", "normal",
                    "--pretrain", "1", "--c", "10"], str(tmp_path))
    assert out2.count("c:0: recall=") == 1 and out2.count("c:10.0: recall=") == 1, out2


def test_lightgcn_cli_trains_on_the_row_normalised_adjacencies(tmp_path):
    """--adj_type norm | gcmc (LightGCN.py:667-678: D^-1 A, not symmetric) through the CLI: the model takes the transposed
    matrix for its backward pass (macr_lgcn_train_step_t) and learns -- the loss falls over the epochs and the evaluation runs."""
    for adj_type, banner in (("norm", "use the normalized adjacency matrix"), ("gcmc", "use the gcmc adjacency matrix")):
        out = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN.py"), "--data_path", os.path.join(REPO, "data") + "/",
                        "--dataset", "addressa", "--verbose", "1", "--layer_size", "[64,64]", "--Ks", "[20]", "--loss", "bceboth",
                        "--test", "rubiboth", "--c", "40", "--epoch", "6", "--lr", "0.001", "--batch_size", "1024", "--gpu_id", "0",
                        "--log_interval", "3", "--adj_type", adj_type, "--save_flag", "0", "--sampler", "device",
                        "--weights_path", str(tmp_path) + "/"], str(tmp_path))
        assert banner in out, out[-1500:]
        losses = [float(l.split("train==[")[1].split("=")[0]) for l in out.splitlines() if "train==[" in l]
        assert len(losses) >= 4 and losses[-1] < losses[0], out[-1500:]
        hit = float(out.split("hit=[")[-1].split(",")[0])
        assert 0.0 < hit < 1.0


def test_mf_cli_two_ranks_one_gpu(tmp_path):
    """torch.distributed.run with two ranks (gloo rig on one GPU): replicas train, rank 0 alone prints and saves, the
    evaluation is item-sharded after a broadcast of rank 0's parameters -- same metrics as the single-rank run up to
    the run-to-run noise of floating-point atomics."""
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    args_ = [os.path.join(REPO, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size", "1024", "--cuda", "0",
             "--saveID", "m", "--log_interval", "2", "--lr", "0.001", "--epoch", "2", "--train", "rubibceboth", "--test", "rubi",
             "--c", "40", "--alpha", "1e-3", "--beta", "1e-3"]
    one = _run_cli(args_, str(tmp_path))
    env = dict(os.environ, MACR_DIST_BACKEND="gloo", PYTHONUNBUFFERED="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533"] + args_, cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
    l1 = [l for l in one.splitlines() if l.startswith("c:40.00")]
    l2 = [l for l in two.stdout.splitlines() if l.startswith("c:40.00")]
    assert len(l1) == 1 and len(l2) == 1, two.stdout                       # rank 0 alone reports
    h1, h2 = (float(l.split("hit=[")[1].split(",")[0]) for l in (l1[0], l2[0]))
    assert abs(h1 - h2) <= 0.02 * max(h1, 1e-3), (h1, h2)


@pytest.mark.parametrize("train,test_", [("rubibceboth", "rubi"), ("normalbce", "normal")])
def test_mf_cli_row_sharded_two_ranks_one_gpu(tmp_path, train, test_):
    """--row_shard 1 under torch.distributed.run (gloo rig, both ranks on this GPU): ONE model, table rows interleaved over
    the ranks, item-sharded evaluation of the strided shards, checkpoint of the reassembled tables -- same losses and
    metrics as the unsharded single-process run from the same seed (same initial model, same batches)."""
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    args_ = [os.path.join(REPO, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size", "1024", "--cuda", "0",
             "--saveID", "rs", "--log_interval", "2", "--lr", "0.001", "--epoch", "2", "--train", train, "--test", test_,
             "--c", "40", "--alpha", "1e-3", "--beta", "1e-3"]
    one = _run_cli(args_, str(tmp_path))
    ckpt = tmp_path / "mf_addressa_checkpoint/wd_1e-05_lr_0.001_rs/1_ckpt.pt"
    ref_sd = torch.load(ckpt, map_location="cpu")
    os.remove(ckpt)
    env = dict(os.environ, MACR_DIST_BACKEND="gloo", PYTHONUNBUFFERED="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547"] + args_ + ["--row_shard", "1"],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
    tag = "c:40.00" if test_ == "rubi" else "Epoch 1"
    l1 = [l for l in one.splitlines() if l.startswith(tag) and "hit=[" in l]
    l2 = [l for l in two.stdout.splitlines() if l.startswith(tag) and "hit=[" in l]
    assert len(l1) == 1 and len(l2) == 1, two.stdout
    loss1, loss2 = (float(l.split("train==[")[1].split("=")[0]) for l in (l1[0], l2[0]))
    assert abs(loss1 - loss2) <= 1e-5 * abs(loss1), (loss1, loss2)
    h1, h2 = (float(l.split("hit=[")[1].split(",")[0]) for l in (l1[0], l2[0]))
    assert abs(h1 - h2) <= 0.02 * max(h1, 1e-3), (h1, h2)
    sd = torch.load(ckpt, map_location="cpu")                              # written by rank 0 from the reassembled tables
    for name in ("user_embedding", "item_embedding"):
        assert sd[name].shape == ref_sd[name].shape
        assert float((sd[name] - ref_sd[name]).abs().max()) <= 2e-3 * 1e-3 * 30, name


def test_mf_cli_row_sharded_resume_two_ranks_one_gpu(tmp_path):
    """--resume 1 with --row_shard 1 on two ranks (round 4 deadlocked here: only rank 0 had a model when the collective
    state_dict() ran): the main rank reads the checkpoint, its tensors are broadcast, every rank takes its rows.  Two epochs +
    a resumed third and fourth must print the losses of an uninterrupted four-epoch run of the same sharded model, and an
    UNSHARDED run resumes a sharded run's checkpoint (and the other way round) instead of failing on missing keys."""
    os.symlink(os.path.join(REPO, "data"), tmp_path / "data")
    base = [os.path.join(REPO, "macr_mf", "train.py"), "--dataset", "addressa", "--batch_size", "1024", "--cuda", "0",
            "--log_interval", "2", "--lr", "0.001", "--train", "rubibceboth", "--test", "rubi", "--c", "40", "--alpha", "1e-3",
            "--beta", "1e-3"]
    env = dict(os.environ, MACR_DIST_BACKEND="gloo", PYTHONUNBUFFERED="1")

    def two(extra, port):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port)] + base + ["--row_shard", "1"] + extra,
                           cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return r.stdout

    def losses(text):
        out, cur = {}, None
        for l in text.splitlines():
            if l.startswith("Epoch "):
                cur = int(l.split()[1].rstrip(":"))
            if (l.startswith("Epoch ") or l.startswith("c:")) and "train==[" in l:
                out[cur] = float(l.split("train==[")[1].split("=")[0])
        return out
    whole = losses(two(["--saveID", "whole", "--epoch", "4"], 29551))
    two(["--saveID", "cut", "--epoch", "2"], 29552)
    out = two(["--saveID", "cut", "--epoch", "4", "--resume", "1"], 29553)
    assert "resumed from epoch 1" in out
    got = losses(out)
    assert sorted(got) == [2, 3] and all(abs(got[e] - whole[e]) <= 2e-5 * abs(whole[e]) + 1.1e-5 for e in got), (got, whole)
    # across modes: the sharded run's checkpoint continues unsharded, the unsharded one continues sharded
    out = _run_cli(base + ["--saveID", "cut", "--epoch", "6", "--resume", "1"], str(tmp_path))
    assert "resumed from epoch 3" in out and sorted(losses(out)) == [4, 5], out
    out = two(["--saveID", "cut", "--epoch", "8", "--resume", "1"], 29554)
    assert "resumed from epoch 5" in out and sorted(losses(out)) == [6, 7], out


def test_row_sharded_training_two_ranks_one_gpu(tmp_path):
    """configs[4]'s training path at test size: P, Q and their Adam state range-sharded over two ranks (gloo rig on one
    GPU), the macr_shard_* device entry points, three collectives per step; losses and the reassembled tables must
    equal the single-GPU step and the oracle within the single-GPU tolerances."""
    import json
    env = dict(os.environ, PYTHONUNBUFFERED="1", MACR_SHARD_SPLIT="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(REPO, "tests", "shard_worker.py")],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"], res
    assert res["world"] == 2 and res["rows_on_rank0"] < 0.51 * res["rows_total"]     # a rank holds half of the rows


def test_row_sharded_split_step_two_ranks_one_gpu(tmp_path):
    """The split step (round 5): forward and backward of each rank's slice only, rows to the slices and gradient rows back to
    the owners by two all-to-alls -- same losses and tables as the single-GPU step and the oracle, and a rank moves about
    (W-1)/W * 3B/W rows each way instead of reducing a 3B-row buffer."""
    import json
    env = dict(os.environ, PYTHONUNBUFFERED="1", MACR_SHARD_SPLIT="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(REPO, "tests", "shard_worker.py")],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and res["split"], res
    assert 0 < res["wire_rows"] < 0.6 * res["batch_rows"], res       # ~ 2 * (1/2) * 3B/2 at two ranks


@pytest.mark.parametrize("split", ["0", "1"])
def test_row_sharded_lazy_adam_two_ranks_one_gpu(tmp_path, split):
    """Both step forms with the lazy dense Adam pass (period 2, five steps: rows are behind when the next batch gathers them, the
    owners bring them up to date on their way out) -- same losses and tables as the single-GPU step and the oracle."""
    import json
    env = dict(os.environ, PYTHONUNBUFFERED="1", MACR_SHARD_SPLIT=split, MACR_LAZY_ADAM="2", MACR_TEST_STEPS="5")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(REPO, "tests", "shard_worker.py")],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and res["lazy_period"] == 2 and res["split"] == (split == "1"), res


def test_rccl_code_paths_in_a_world_of_one(tmp_path):
    """The collectives of the multi-GPU paths on backend "nccl" (RCCL) with device tensors, driven from ONE GPU
    (MACR_FORCE_COLLECTIVES=1: world-size-1 collectives are issued instead of skipped): the packed int64 all-gather of
    per-shard top-K lists, max_over_ranks, broadcast_params, and a row-sharded training step's two all-reduces and its
    broadcast -- results equal to the single-GPU step."""
    import json
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "rccl_world1_worker.py")], cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"], res


def test_injected_weights_round_trip(ops):
    """a1: parity runs inject the reference's initial values (`weights=`): the model must hold exactly those numbers,
    share them between its optimizer instances, and hand them back through state_dict()."""
    import types
    from macr_amd.mf import BPRMF
    rs = np.random.RandomState(3)
    n_users, n_items, d = 70, 40, 64
    wts = {"user_embedding": rs.standard_normal((n_users, d)).astype(np.float32),
           "item_embedding": rs.standard_normal((n_items, d)).astype(np.float32),
           "w": rs.standard_normal((d, 1)).astype(np.float32), "w_user": rs.standard_normal((d, 1)).astype(np.float32)}
    args = types.SimpleNamespace(regs=1e-5, embed_size=d, lr=1e-3, batch_size=32, verbose=0, c=40.0, alpha=1e-2, beta=1e-3)
    m = BPRMF(args, dict(n_users=n_users, n_items=n_items), weights=wts)
    assert np.array_equal(m.user_embedding.cpu().numpy(), wts["user_embedding"])
    assert np.array_equal(m.item_embedding.cpu().numpy(), wts["item_embedding"])
    assert np.array_equal(m.w.cpu().numpy(), wts["w"].reshape(-1)) and np.array_equal(m.w_user.cpu().numpy(), wts["w_user"].reshape(-1))
    for kind in (ops.LOSS_NORMALBCE, ops.LOSS_RUBIBCEBOTH, ops.LOSS_RUBIBCE):      # one storage behind every optimizer
        st = m.opt_state(kind)
        assert st.P.data_ptr() == m.user_embedding.data_ptr() and st.w.data_ptr() == m.w.data_ptr()
    sd = m.state_dict()
    assert np.array_equal(sd["user_embedding"].cpu().numpy(), wts["user_embedding"])
    # without injection: Xavier-uniform, inside the limit sqrt(6 / (rows + d))
    m2 = BPRMF(args, dict(n_users=n_users, n_items=n_items), seed=5)
    L = np.sqrt(6.0 / (n_users + d))
    x = m2.user_embedding.cpu().numpy()
    assert np.abs(x).max() <= L and np.abs(x).max() > 0.9 * L
    assert np.abs(m2.w.cpu().numpy()).max() <= np.sqrt(6.0 / (d + 1))


@pytest.mark.parametrize("d", [20, 48, 100, 200])
def test_any_embed_size_runs_zero_padded_and_exact(ops, d):
    """--embed_size is any integer in the reference (macr_mf/parse.py:27, utility/parser.py:32).  The models run other
    widths at the next of 32/64/128/256 with zero columns (ops.padded_dim): losses and parameters are the oracle's at
    width d, the padding stays exactly zero, and the ranking is the oracle's bit for bit."""
    import types
    import scipy.sparse as sp
    from macr_amd.mf import BPRMF
    from macr_amd.lightgcn import LightGCN
    from macr_amd.evaluator import Evaluator
    rs = np.random.RandomState(d)
    n_users, n_items, B, K = 300, 120, 64, 20
    wts = {"user_embedding": (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32),
           "item_embedding": (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32),
           "w": (rs.standard_normal((d, 1)) * 0.3).astype(np.float32), "w_user": (rs.standard_normal((d, 1)) * 0.3).astype(np.float32)}
    lr, decay, alpha, beta = 1e-3, 1e-5, 1e-2, 1e-3
    args = types.SimpleNamespace(regs=decay, embed_size=d, lr=lr, batch_size=B, verbose=0, c=40.0, alpha=alpha, beta=beta)
    m = BPRMF(args, dict(n_users=n_users, n_items=n_items), weights=wts)
    dp = ops.padded_dim(d)
    assert m.user_embedding.shape == (n_users, dp) and m.w.shape == (dp,)
    Po, Qo = wts["user_embedding"].copy(), wts["item_embedding"].copy()
    wo, wuo = wts["w"].reshape(-1).copy(), wts["w_user"].reshape(-1).copy()
    st = oracle.AdamState([Po.shape, Qo.shape, (d,), (d,)])
    kind = ops.LOSS_RUBIBCEBOTH
    for t in range(3):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = rs.randint(0, n_items, B).astype(np.int32); j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, B)
        got = m.train_step(kind, m.to_device_batch(u, i, j)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5)
    m.sync()
    P, Q = m.user_embedding.cpu().numpy(), m.item_embedding.cpu().numpy()
    np.testing.assert_allclose(P[:, :d], Po, rtol=0, atol=2e-3 * lr * 3)
    np.testing.assert_allclose(Q[:, :d], Qo, rtol=0, atol=2e-3 * lr * 3)
    np.testing.assert_allclose(m.w.cpu().numpy()[:d], wo, rtol=0, atol=2e-3 * lr * 3)
    assert not P[:, d:].any() and not Q[:, d:].any() and not m.w.cpu().numpy()[d:].any() and not m.w_user.cpu().numpy()[d:].any()
    # ranking of the padded model == oracle ranking of its first d columns, ids and score bits
    users = np.arange(0, n_users, 3, dtype=np.int32)
    mask = [sorted(rs.choice(n_items, 10, replace=False).tolist()) for _ in users]
    gt = [sorted(rs.choice(n_items, 4, replace=False).tolist()) for _ in users]
    ev = Evaluator(mask, gt, n_items, torch.device("cuda"))
    uid = torch.from_numpy(users).cuda()
    val, idx, cnt = ev.rank(ops.SCORE_RUBI_BOTH, m.user_embedding, uid, m.item_embedding, K, m.w, m.w_user, 40.0)
    sig_i = ops.branch_sigmoid(m.item_embedding, m.w).cpu().numpy()
    sig_u = ops.branch_sigmoid(m.user_embedding, m.w_user, uid).cpu().numpy()
    wv, wi, wc = oracle.score_topk(oracle.SCORE_RUBI_BOTH, np.ascontiguousarray(P[users, :d]), np.ascontiguousarray(Q[:, :d]), K,
                                   sig_u, sig_i, 40.0, oracle.csr_from_lists(mask))
    assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(val.cpu().numpy().view(np.uint32), wv.view(np.uint32))
    # LightGCN: propagation is linear, zero columns stay zero
    R = (rs.rand(n_users, n_items) < 0.08).astype(np.float32)
    R[:, 0] = 1
    A = sp.bmat([[None, sp.csr_matrix(R)], [sp.csr_matrix(R.T), None]]).tocsr()
    deg = np.asarray(A.sum(1)).ravel()
    dinv = np.where(deg > 0, np.power(np.maximum(deg, 1), -0.5), 0).astype(np.float32)
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32); A.sort_indices()
    largs = types.SimpleNamespace(adj_type="pre", alg_type="lightgcn", lr=lr, embed_size=d, batch_size=B, layer_size="[%d,%d]" % (d, d),
                                  regs="[%g]" % decay, verbose=0, Ks="[20]", alpha=alpha, beta=beta, dataset="x", node_dropout_flag=0)
    lg = LightGCN(dict(n_users=n_users, n_items=n_items, norm_adj=A), largs, weights=wts)
    To = np.concatenate([wts["user_embedding"], wts["item_embedding"]]).astype(np.float32)
    wo, wuo = wts["w"].reshape(-1).copy(), wts["w_user"].reshape(-1).copy()
    lst = oracle.AdamState([To.shape, (d,), (d,)])
    for t in range(2):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = rs.randint(0, n_items, B).astype(np.int32); j = rs.randint(0, n_items, B).astype(np.int32)
        want = oracle.lgcn_train_step(kind, n_users, n_items, 2, A.indptr, A.indices, A.data, u, i, j, To, wo, wuo, lst,
                                      lr, decay, alpha, beta, B)
        got = lg.train_step(kind, lg.to_device_batch(u, i, j)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5)
    T = lg.T.cpu().numpy()
    np.testing.assert_allclose(T[:, :d], To, rtol=0, atol=0.02 * lr * 2)
    assert not T[:, d:].any()


@pytest.mark.parametrize("kind_name", ["SCORE_RUBI_BOTH", "SCORE_RUBI", "SCORE_DIRECT_MINUS_BOTH"])
def test_shared_listing_pass_sweep_equals_single_c(ops, kind_name, eval_filter):
    """f1: macr_score_topk_sweep ranks up to four values of c with ONE listing pass; every value must give exactly the
    lists macr_score_topk gives for that c (ids and score bits), and Evaluator.test_mf_sweep the metrics of separate
    test_mf calls -- 7 values: a group of four and a group of three."""
    from macr_amd.evaluator import Evaluator
    kind = getattr(ops, kind_name)
    rs = np.random.RandomState(9)
    d, n_users, n_items, U, K = 64, 3000, 5000, 1100, 20
    P = dev((rs.standard_normal((n_users, d)) * 0.4).astype(np.float32))
    Q = dev((rs.standard_normal((n_items, d)) * 0.4).astype(np.float32))
    w, wu = dev((rs.standard_normal(d) * 0.3).astype(np.float32)), dev((rs.standard_normal(d) * 0.3).astype(np.float32))
    users = np.sort(rs.choice(n_users, U, replace=False)).astype(np.int32)
    mask = [sorted(rs.choice(n_items, 25, replace=False).tolist()) for _ in range(U)]
    gt = [sorted(rs.choice(n_items, 5, replace=False).tolist()) for _ in range(U)]
    uid = dev(users)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    sig_i = ops.branch_sigmoid(Q, w)
    sig_u = ops.branch_sigmoid(P, wu, uid)
    cs = [-3.0, 0.0, 12.5, 40.0]
    sv, si = ops.score_topk_sweep(kind, P, uid, Q, K, sig_u, sig_i, dev(np.asarray(cs, np.float32)), mcsr, 0)
    for g, c in enumerate(cs):
        v1, i1 = ops.score_topk(kind, P, uid, Q, K, sig_u, sig_i, c, mcsr, 0)
        mv, mi, _ = ops.topk_merge(v1, i1)
        assert torch.equal(si[g], mi), (kind_name, c)
        assert torch.equal(sv[g].view(torch.int32), mv.view(torch.int32)), (kind_name, c)
    ev = Evaluator(mask, gt, n_items, torch.device("cuda"))
    cs7 = list(np.linspace(-5.0, 40.0, 7))
    for rep in range(2):                                   # second round: graph replays
        got = ev.test_mf_sweep(kind, P, uid, Q, [5, 20], w, wu, cs7)
    for c, res in zip(cs7, got):
        fresh = Evaluator(mask, gt, n_items, torch.device("cuda"))
        fresh.use_graph = False
        want = fresh.test_mf(kind, P, uid, Q, [5, 20], w, wu, float(c))
        for k in want:
            assert np.array_equal(res[k], want[k]), (kind_name, c, k)


def test_lightgcn_tune_cli(tmp_path):
    """LightGCN_tune.py (LightGCN_tune.py:852-870): the c values of a sweep share the listing pass; every line of the
    sweep must equal what LightGCN.py --pretrain 1 reports for that c on the saved weights."""
    common = ["--data_path", os.path.join(REPO, "data") + "/", "--dataset", "addressa", "--verbose", "1", "--layer_size",
              "[64,64]", "--Ks", "[20]", "--lr", "0.001", "--batch_size", "1024", "--gpu_id", "0", "--log_interval", "2",
              "--alpha", "1e-2", "--beta", "1e-3", "--weights_path", str(tmp_path) + "/", "--saveID", "s", "--loss", "bceboth",
              "--test", "rubiboth"]
    out = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN_tune.py")] + common + ["--epoch", "2", "--start", "0", "--end",
                   "40", "--step", "5"], str(tmp_path))
    lines = [l for l in out.splitlines() if l.startswith("c:")]
    assert [l.split()[0] for l in lines] == ["c:0.00", "c:10.00", "c:20.00", "c:30.00", "c:40.00"], out
    out2 = _run_cli([os.path.join(REPO, "macr_lightgcn", "LightGCN.py")] + common + ["--pretrain", "1", "--c", "40"], str(tmp_path))
    # --pretrain prints full-precision arrays; compare hit@20 of c=0 and c=40 with the sweep's lines at 5 digits
    def hit_of(line):
        return float(line.split("hit=[")[1].split(",")[0].rstrip("]"))
    pre = {l.split(":")[1]: l for l in out2.splitlines() if l.startswith("c:")}
    for c_key, sweep_line in (("0", lines[0]), ("40.0", lines[-1])):
        hr = float(pre[c_key].split("hit=[")[1].split("]")[0].split()[0])
        assert abs(hr - hit_of(sweep_line)) < 6e-6, (pre[c_key], sweep_line)


def test_evaluator_steps_down_from_the_fp16_filter_where_it_cannot_resolve_the_top(ops, monkeypatch):
    """Filter policy: (y - 30) sig_i sig_u on untrained rows of d = 128 packs every query's top closer than the fp16 filter's
    margin (tests/test_gpu_ops.py, the prologue test's third shape): the first, unseeded fp16 evaluation lists query blocks
    twice, so the next one runs under the bf16 filter (whose margin is 12x tighter), after which fp16 is tried again with
    the seeds the bf16 ranking left.  The metrics are the fp32-filter evaluator's throughout."""
    from macr_amd.evaluator import Evaluator
    monkeypatch.setenv("MACR_EVAL_FILTER", "f16")
    rs = np.random.RandomState(141 + 513)
    U, N, d = 513, 20011, 128
    P = dev((rs.standard_normal((U + 50, d)) * 0.4).astype(np.float32))
    Q = dev((rs.standard_normal((N, d)) * 0.4).astype(np.float32))
    w, wu = dev((rs.standard_normal(d) * 0.3).astype(np.float32)), dev((rs.standard_normal(d) * 0.3).astype(np.float32))
    uid = dev(rs.permutation(U + 50)[:U].astype(np.int32))
    mask = [sorted(rs.choice(N, 30, replace=False).tolist()) for _ in range(U)]
    gt = [sorted(rs.choice(N, 5, replace=False).tolist()) for _ in range(U)]
    ev = Evaluator(mask, gt, N, torch.device("cuda"))
    assert ev.filter == "f16"
    monkeypatch.setenv("MACR_EVAL_FILTER", "f32")
    plain = Evaluator(mask, gt, N, torch.device("cuda"))
    plain.use_seeds, plain.use_graph = False, False
    want = plain.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 30.0)
    used = []
    for step in range(4):
        got = ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 30.0)
        torch.cuda.synchronize()
        info = ev.last_eval_info()
        used.append((info.get("filter"), info["seeded"], info["query_blocks_relisted"], info["exact_fallback"]))
        for k in want:
            assert np.array_equal(got[k], want[k]), (step, k, used)
    assert used[0][0] == "f16" and not used[0][1]
    if used[0][2] or used[0][3]:                  # (what this shape does today; were the margin ever tightened, nothing to step down from)
        assert used[1][0] == "bf16", used
        assert used[2][0] == "f16", used          # back-off of one evaluation, then fp16 again -- seeded this time
    assert all(u[3] == 0 for u in used if u[0] != "f16"), used


def test_evaluator_seeds_follow_the_model_and_back_off_when_it_jumps(ops, eval_filter):
    """Evaluator.rank_local seeds every ranking's thresholds with the best candidates of the previous one.  Tables that
    drift a little between two evaluations keep the seeds good (no query block listed twice); a model that jumps
    (new item table) makes them stale: that evaluation pays a repair round, the next one goes back to the sampling pass
    (1, 2, 4 ... evaluations of back-off).  The metrics never depend on any of it: equal to a seedless evaluator's."""
    from macr_amd.evaluator import Evaluator
    rs = np.random.RandomState(11)
    d, n_users, n_items, U = 64, 5000, 9000, 3000
    P = dev((rs.standard_normal((n_users, d)) * 0.4).astype(np.float32))
    Q = dev((rs.standard_normal((n_items, d)) * 0.4).astype(np.float32))
    w, wu = dev((rs.standard_normal(d) * 0.3).astype(np.float32)), dev((rs.standard_normal(d) * 0.3).astype(np.float32))
    users = np.sort(rs.choice(n_users, U, replace=False)).astype(np.int32)
    mask = [sorted(rs.choice(n_items, 20, replace=False).tolist()) for _ in range(U)]
    gt = [sorted(rs.choice(n_items, 5, replace=False).tolist()) for _ in range(U)]
    uid = dev(users)
    ev = Evaluator(mask, gt, n_items, torch.device("cuda"))
    plain = Evaluator(mask, gt, n_items, torch.device("cuda"))
    plain.use_seeds, plain.use_graph = False, False
    gen = torch.Generator(device="cuda").manual_seed(3)
    modes = []
    for step in range(9):
        if step in (1, 2, 3, 6, 7, 8):        # drift: every row moves by ~1 % of its length
            Q.add_(torch.randn(Q.shape, generator=gen, device="cuda") * 0.004)
            P.add_(torch.randn(P.shape, generator=gen, device="cuda") * 0.004)
        if step in (4, 5):                    # jump: a new item table (in place: the captured graphs read the same buffers)
            Q.copy_(torch.randn(Q.shape, generator=gen, device="cuda") * 0.4)
        got = ev.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 30.0)
        torch.cuda.synchronize()
        info = ev.last_eval_info()
        modes.append((info["seeded"], [info["query_blocks_relisted"], info["exact_fallback"]]))
        want = plain.test_mf(ops.SCORE_RUBI_BOTH, P, uid, Q, [20], w, wu, 30.0)
        for k in want:
            assert np.array_equal(got[k], want[k]), (step, k, modes)
    seeded = [m[0] for m in modes]
    relisted = [m[1][0] for m in modes]
    assert seeded[0] is False                                    # nothing to seed the first ranking with
    assert seeded[1:5] == [True] * 4 and relisted[1:4] == [0, 0, 0]      # drift: seeds hold
    assert relisted[4] > 0                                       # jump: stale seeds, repaired
    assert seeded[5] is False                                    # ... so the next evaluation samples (back-off 1)
    assert seeded[6] is True and relisted[6] == 0                # and the one after tries seeds again: they hold
    assert all(m[1][1] == 0 for m in modes)                      # the exact fallback kernel never ran
    # graphs: the first round seeded and sampled, and the complete sampled sequence the stale evaluation fell back to
    assert len(ev._graphs) == (3 if ev.optimistic else 2)
    assert ev.fast_stats == ({"fast": 8, "redone": 1} if ev.optimistic else {"fast": 0, "redone": 0})
