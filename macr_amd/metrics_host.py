"""Host-side metric helpers with the reference's names and semantics (macr_mf/train.py:32-117).

They exist for callers that imported these functions from the reference's train.py and for the
golden unit cases (G8); the evaluator itself computes the same quantities on the device
(macr_metrics_mf).  r = hit flags of a ranked list, float64 arithmetic like NumPy's."""
import numpy as np


def precision_at_k(r, k):
    assert k >= 1
    return np.mean(np.asarray(r)[:k])


def dcg_at_k(r, k, method=1):
    r = np.asarray(r, dtype=np.float64)[:k]
    if r.size:
        if method == 0:
            return r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1)))
        if method == 1:
            return np.sum(r / np.log2(np.arange(2, r.size + 2)))
        raise ValueError('method must be 0 or 1.')
    return 0.


def ndcg_at_k(r, k, maxlen, method=1):
    """DCG / ideal DCG of min(maxlen, k) hits (the reference normalises by the number of test items)."""
    ideal = (1. / np.log2(np.arange(2, k + 2)))[:min(maxlen, k)].sum()
    if not ideal:
        return 0.
    return dcg_at_k(r, k, method) / ideal


def recall_at_k(r, k, all_pos_num):
    return np.sum(np.asarray(r, dtype=np.float64)[:k]) / all_pos_num


def hit_at_k(r, k):
    return 1. if np.sum(np.asarray(r)[:k]) > 0 else 0.


def get_performance(user_pos_test, r, Ks):
    out = dict(precision=[], recall=[], ndcg=[], hit_ratio=[])
    for K in Ks:
        out['precision'].append(precision_at_k(r, K))
        out['recall'].append(recall_at_k(r, K, len(user_pos_test)))
        out['ndcg'].append(ndcg_at_k(r, K, len(user_pos_test)))
        out['hit_ratio'].append(hit_at_k(r, K))
    return {k: np.array(v) for k, v in out.items()}
