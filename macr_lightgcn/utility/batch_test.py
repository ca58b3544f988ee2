"""Import-time harness glue + `test()` of the LightGCN CLI, as in the reference
(macr_lightgcn/utility/batch_test.py: globals :15-23, test :26-162)."""
import multiprocessing

import numpy as np
import torch

from utility.parser import parse_args
from utility.load_data import *          # noqa: F401,F403  (Data)
from evaluator import eval_score_matrix_foldout  # noqa: F401

from macr_amd import ops
from macr_amd.evaluator import Evaluator
from macr_amd.eval_cache import EvaluatorCache

cores = multiprocessing.cpu_count() // 2
args = parse_args()
data_generator = Data(path=args.data_path + args.dataset, batch_size=args.batch_size, args=args)
USR_NUM, ITEM_NUM = data_generator.n_users, data_generator.n_items
N_TRAIN, N_TEST = data_generator.n_train, data_generator.n_test
BATCH_SIZE = args.batch_size

_METHODS = {"normal": ops.SCORE_NORMAL, "rubiboth": ops.SCORE_RUBI_BOTH}
_evaluators = EvaluatorCache(max_cached=4)


def _evaluator_for(model, users_to_test):
    """(Evaluator, device user ids) of this user list; built once per list (macr_amd/eval_cache.py)"""
    def build(users):
        mask, gt = data_generator.eval_lists(users)
        return (Evaluator(mask, gt, ITEM_NUM, model.device),
                torch.tensor(list(users), dtype=torch.int32, device=model.device))
    return _evaluators.get("test", users_to_test, build)


def test(sess, model, users_to_test, drop_flag=False, train_set_flag=0, method="normal"):
    """Reference signature (batch_test.py:26).  Scores every user in `users_to_test` against all items on
    the PROPAGATED embeddings (computed once per call, not once per user batch), masks train items,
    ranks, and returns {'hr','recall','ndcg'} at model.Ks exactly as :134-161 post-processes the C++
    evaluator output.  sess / drop_flag are accepted for compatibility."""
    if method not in _METHODS:
        raise NotImplementedError("method %r is outside the MI355X hot path (normal | rubiboth)" % method)
    if train_set_flag != 0:
        raise NotImplementedError("train_set_flag != 0 is unused by the reference CLI")
    evaluator, uid = _evaluator_for(model, users_to_test)
    ua, ia = model.propagated()
    ret = evaluator.test_lgcn(_METHODS[method], ua, uid, ia.contiguous(), model.Ks, model.w, model.w_user,
                              model.rubi_c)
    # the reference indexes the result with the ORIGINAL order of Ks after sorting columns (:153-161)
    return {k: np.asarray(v) for k, v in ret.items()}


def test_sweep(sess, model, users_to_test, cs, method="rubiboth"):
    """test() for every c of `cs` (the loop of LightGCN_tune.py:852-870) -> list of result dicts; the values share the
    listing pass in groups of four (macr_score_topk_sweep)."""
    if _METHODS.get(method, ops.SCORE_NORMAL) == ops.SCORE_NORMAL:
        raise NotImplementedError("method %r has no c to sweep" % method)
    evaluator, uid = _evaluator_for(model, users_to_test)
    ua, ia = model.propagated()
    rets = evaluator.test_lgcn_sweep(_METHODS[method], ua, uid, ia.contiguous(), model.Ks, model.w, model.w_user, list(cs))
    return [{k: np.asarray(v) for k, v in r.items()} for r in rets]
