"""CLI surface on CPU: flag names/defaults against the reference's parsers (golden G9) and the
early-stopping helpers."""
import importlib.util
import os

from helpers import REPO, golden


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_mf_flags_match_reference_defaults():
    mod = load(os.path.join(REPO, "macr_mf", "parse.py"), "mf_parse")
    ours = vars(mod.parse_args([]))
    ref = golden("mf", "tiny")["G9"]
    for k, v in ref.items():
        assert k in ours, k
        assert ours[k] == v and type(ours[k]) is type(v), (k, ours[k], v)
    assert set(ours) - set(ref) == {"seed", "sampler", "resume", "row_shard"}  # additive flags only


def test_lightgcn_flags_match_reference_defaults():
    mod = load(os.path.join(REPO, "macr_lightgcn", "utility", "parser.py"), "lg_parser")
    ours = vars(mod.parse_args([]))
    ref = golden("lgcn", "tiny")["G9"]
    for k, v in ref.items():
        assert k in ours, k
        assert ours[k] == v and type(ours[k]) is type(v), (k, ours[k], v)
    assert set(ours) - set(ref) == {"seed", "sampler", "resume"}
    args = mod.parse_args("--layer_size [64,64] --Ks [20] --loss bceboth --test rubiboth --gpu_id 0".split())
    assert args.layer_size == "[64,64]" and args.loss == "bceboth"


def test_early_stopping_helper():
    mod = load(os.path.join(REPO, "macr_lightgcn", "utility", "helper.py"), "lg_helper")
    best, step, stop = 0., 0, False
    for v in [0.1, 0.2, 0.2] + [0.15] * 10:
        best, step, stop = mod.early_stopping(v, best, step, expected_order='acc', flag_step=10)
    assert best == 0.2 and step == 10 and stop


def test_lightgcn_epoch_consumes_the_reference_number_of_sampler_draws():
    """macr_lightgcn/LightGCN.py:762-778 / :799-812 fetch one batch ahead: a training pass draws n_batch + 1 batches from
    Data.sample() and a test-loss pass n_batch + 1 from Data.sample_test(); the host RNG streams of a whole run only stay
    those of the reference if this CLI draws as many (in a subprocess: the module parses sys.argv at import)."""
    import subprocess
    import sys
    code = r'''
import sys, types
sys.argv = ["LightGCN.py", "--data_path", "%s/", "--dataset", "addressa", "--batch_size", "1024"]
sys.path.insert(0, "%s")
import LightGCN as cli
calls = {"sample": 0, "sample_test": 0}
dg = cli.data_generator
orig, orig_t = dg.sample, dg.sample_test
dg.sample = lambda: (calls.__setitem__("sample", calls["sample"] + 1), orig())[1]
dg.sample_test = lambda: (calls.__setitem__("sample_test", calls["sample_test"] + 1), orig_t())[1]
class FakeLog(list):
    def __getitem__(self, k):
        return self if isinstance(k, slice) else None
    def cpu(self): return self
    def numpy(self):
        import numpy as np
        return np.zeros((3, 3))
class FakeModel(object):
    def to_device_batch(self, u, i, j): return (u, i, j)
    def train_step(self, kind, batch, out, loss_only=False): pass
cli.train_epoch(FakeModel(), 0, 3, FakeLog())
cli.train_epoch(FakeModel(), 0, 3, FakeLog(), test_loss=True)
print(calls["sample"], calls["sample_test"])
''' % (os.path.join(REPO, "data"), os.path.join(REPO, "macr_lightgcn"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-2:] == ["4", "4"], out.stdout


def test_bench_gpus_n_becomes_its_own_launcher():
    """`python bench.py --gpus 2 ...` exactly as the driver types it (no torch.distributed.run in front): bench.py must start
    its two ranks itself.  Without a GPU every rank stops at "needs an MI355X" -- which proves both ranks were started with
    the rendezvous environment, and that the launcher hands their status back."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    import torch
    if torch.cuda.is_available():
        return                                           # (on a GPU box the -m gpu test runs the whole thing)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-3000:]
