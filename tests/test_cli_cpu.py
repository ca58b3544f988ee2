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
    assert set(ours) - set(ref) == {"seed", "sampler", "resume"}  # additive flags only


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
