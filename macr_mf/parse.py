"""Command-line flags of the MF CLI -- same names, types and defaults as the reference's
macr_mf/parse.py:3-92, so existing launch lines keep working.  Only --model mf with
--train {normalbce,rubibceboth} / --test {normal,rubi} runs on the MI355X hot path; flags
that belong to out-of-scope baselines are accepted for compatibility.
Additive flags (not in the reference): --seed, --sampler."""
import argparse

# (flag, type or None for nargs='?', default, help)
_FLAGS = [
    ("data_path", None, './data/', "directory that holds <dataset>/train.txt, test.txt"),
    ("dataset", None, 'movielens_ml_1m', "dataset name (addressa, gowalla, ml_10m, yelp2018, globe, ...)"),
    ("source", None, 'normal', "normal | dice (dice is out of scope)"),
    ("train", None, 'normalbce', "normalbce | rubibceboth"),
    ("test", None, 'normal', "normal | rubi"),
    ("valid_set", None, 'test', "test | valid"),
    ("alpha", float, 1e-3, "weight of the item-branch loss"),
    ("beta", float, 1e-3, "weight of the user-branch loss"),
    ("early_stop", int, 1, "1: stop after 10 evaluations without HR improvement"),
    ("verbose", int, 1, "print interval (epochs)"),
    ("epoch", int, 1000, "number of epochs"),
    ("embed_size", int, 64, "embedding size (32, 64, 128 or 256 on the HIP path)"),
    ("batch_size", int, 1024, "batch size (training triples and evaluation users)"),
    ("Ks", None, '[20]', "top-K cut-offs, python list literal"),
    ("epochs", None, '[]', "(compat) epochs at which c is tested"),
    ("regs", float, 1e-5, "l2 regularisation"),
    ("c", float, 40.0, "the constant c of counterfactual inference"),
    ("train_c", str, "val", "(compat) val | test"),
    ("lr", float, 1e-3, "learning rate"),
    ("wd", float, 1e-5, "(compat) weight decay, only used in file names"),
    ("model", None, 'mf', "mf (CausalE / IPSmf / biasmf are out of scope)"),
    ("skew", int, 0, "(compat)"),
    ("devide_ratio", float, 0.8, "(compat)"),
    ("save_flag", int, 1, "1: save a checkpoint at every evaluation"),
    ("cuda", str, '1', "HIP device index"),
    ("pretrain", int, 0, "0: train from scratch"),
    ("check_c", int, 1, "(compat)"),
    ("log_interval", int, 10, "evaluate every N epochs"),
    ("pop_wd", float, 0., "(compat)"),
    ("base", float, -1., "(compat)"),
    ("cf_pen", float, 1.0, "(compat)"),
    ("saveID", None, '', "suffix of the checkpoint directory"),
    ("user_min", int, 1, "(compat)"),
    ("user_max", int, 1000, "(compat)"),
    ("data_type", None, 'ori', "ori (imbalanced loaders are out of scope)"),
    ("imb_type", None, 'exp', "(compat)"),
    ("top_ratio", float, 0.1, "(compat)"),
    ("lam", float, 1., "(compat)"),
    ("check_epoch", None, 'all', "(compat)"),
    ("start", float, -1., "tune.py: first c of the sweep"),
    ("end", float, 1., "tune.py: last c of the sweep"),
    ("step", int, 20, "tune.py: number of c values"),
    ("out", int, 0, "(compat)"),
    # additive
    ("seed", int, 12345, "[new] seed of python/numpy/torch RNGs (the reference hard-codes 12345)"),
    ("resume", int, 0, "[new] 1: load the newest checkpoint of this run's checkpoint directory and continue training"),
    ("row_shard", int, 0, "[new] 1: ONE model whose table rows are sharded over the ranks of the process group (torchrun); "
                          "0: every rank trains a replica"),
    ("sampler", str, "reference", "[new] reference: the reference's python `random` stream (host); device: GPU sampler"),
]


def build_parser():
    p = argparse.ArgumentParser(description="Run pop_bias.")
    for name, typ, default, help_ in _FLAGS:
        if typ is None:
            p.add_argument('--' + name, nargs='?', default=default, help=help_)
        else:
            p.add_argument('--' + name, type=typ, default=default, help=help_)
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)
