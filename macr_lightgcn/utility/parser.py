"""Command-line flags of the LightGCN CLI -- same names, types and defaults as the reference's
macr_lightgcn/utility/parser.py:10-104.  On the MI355X hot path: --alg_type lightgcn, --adj_type pre,
--loss {bce,bceboth}, --test {normal,rubiboth}; the remaining flags are accepted for compatibility.
Additive: --seed, --sampler, --resume."""
import argparse

_FLAGS = [
    ("weights_path", None, '', "where checkpoints go"),
    ("data_path", None, '../data/', "directory that holds <dataset>/train.txt, test.txt"),
    ("proj_path", None, '', "(compat)"),
    ("dataset", None, 'gowalla', "dataset name"),
    ("valid_set", None, 'test', "valid | test"),
    ("pretrain", int, 0, "0: train from scratch; 1: load the best saved weights, evaluate c = 0 and the best c, exit"),
    ("verbose", int, 1, "print interval (epochs)"),
    ("is_norm", int, 1, "(compat)"),
    ("epoch", int, 1000, "number of epochs"),
    ("embed_size", int, 64, "embedding size (32, 64, 128 or 256 on the HIP path)"),
    ("layer_size", None, '[64, 64, 64, 64]', "one entry per propagation layer, python list literal"),
    ("batch_size", int, 1024, "batch size"),
    ("regs", None, '[1e-5,1e-5,1e-2]', "regularisation; only the first entry is used"),
    ("lr", float, 0.01, "learning rate"),
    ("c", float, 40.0, "the constant c of counterfactual inference"),
    ("model_type", None, 'lightgcn', "(compat)"),
    ("adj_type", None, 'pre', "adjacency normalisation {plain, norm, gcmc, mean, pre} (LightGCN.py:667-678); pre is the default of every "
                              "README command.  The row-normalised ones (norm, gcmc, mean: D^-1 A) are not symmetric: their backward "
                              "pass runs on the transposed matrix (a second CSR + SpMM plan)"),
    ("alg_type", None, 'lightgcn', "lightgcn (ngcf, gcn, gcmc are out of scope)"),
    ("gpu_id", int, 0, "HIP device index"),
    ("node_dropout_flag", int, 0, "0 (node dropout is out of scope)"),
    ("node_dropout", None, '[0.1]', "(compat)"),
    ("mess_dropout", None, '[0.1]', "(compat)"),
    ("Ks", None, '[1,5,10,15,20,30]', "top-K cut-offs, python list literal (up to 128)"),
    ("save_flag", int, 1, "1: save a checkpoint at every evaluation"),
    ("test_flag", None, 'part', "(compat)"),
    ("saveID", None, '', "suffix of the checkpoint files"),
    ("base", float, -1., "(compat)"),
    ("log_interval", int, 10, "evaluate every N epochs"),
    ("only_test", int, 0, "(compat)"),
    ("loss", None, 'bpr', "bce | bceboth  (bpr, bce1, bce2 are out of scope)"),
    ("alpha", float, 1e-3, "weight of the item-branch loss"),
    ("beta", float, 1e-3, "weight of the user-branch loss"),
    ("test", None, 'normal', "normal | rubiboth"),
    ("early_stop", int, 1, "1: stop after 10 evaluations without HR improvement"),
    ("start", float, -1., "LightGCN_tune.py: first c of the sweep"),
    ("end", float, 1., "LightGCN_tune.py: last c of the sweep"),
    ("step", int, 20, "LightGCN_tune.py: number of c values"),
    ("out", int, 0, "(compat)"),
    ("seed", int, 12345, "[new] seed of python/numpy/torch RNGs (the reference hard-codes 12345)"),
    ("resume", int, 0, "[new] 1: load the newest checkpoint of this run's checkpoint directory and continue training"),
    ("sampler", str, "reference", "[new] reference: the reference's random/numpy streams (host); device: GPU sampler"),
]


def build_parser():
    p = argparse.ArgumentParser(description="Run NGCF.")
    for name, typ, default, help_ in _FLAGS:
        if typ is None:
            p.add_argument('--' + name, nargs='?', default=default, help=help_)
        else:
            p.add_argument('--' + name, type=typ, default=default, help=help_)
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)
