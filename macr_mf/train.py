"""MF command line of MACR on MI355X -- drop-in for the reference's macr_mf/train.py.

    python ./macr_mf/train.py --dataset addressa --batch_size 1024 --cuda 0 --saveID 1 \
        --log_interval 10 --lr 0.001 --train normalbce --test normal
    python ./macr_mf/train.py --dataset gowalla --batch_size 4096 --cuda 0 --saveID 0 --log_interval 10 \
        --lr 0.001 --check_c 1 --c 40 --train rubibceboth --test rubi --alpha 1e-2 --beta 1e-3

Same flags, stdout/log line formats, early stopping and checkpoint directory naming as the
reference (macr_mf/train.py:332-611); the training step and the evaluator run on the HIP
kernels.  `test(sess, model, users, ...)` keeps the reference signature (:162) and returns the
same dict of np.ndarray(len(Ks)).  Checkpoints are torch files (TF's format is not a goal).
"""
import logging
import os
import random
import sys
from time import time

import numpy as np
import torch

from batch_test import *          # noqa: F401,F403  (args, data, Ks, BATCH_SIZE, ITEM_NUM, USER_NUM)
from model import BPRMF, Session

from macr_amd import ops
from macr_amd.evaluator import Evaluator
from macr_amd.eval_cache import EvaluatorCache
from macr_amd.metrics_host import (precision_at_k, dcg_at_k, ndcg_at_k, recall_at_k, hit_at_k,   # noqa: F401
                                   get_performance)

logging.getLogger().setLevel(logging.INFO)

# model_type of the reference's test() (train.py:222-259) -> score kind
_MODEL_TYPES = {'o': ops.SCORE_NORMAL, 'rubi_both': ops.SCORE_RUBI_BOTH, 'rubi_c': ops.SCORE_RUBI,
                'direct_minus_c': ops.SCORE_DIRECT_MINUS}
_evaluators = EvaluatorCache(max_cached=4)


def _evaluator_for(model, test_users, valid_set):
    """(Evaluator, device user ids) of this user list; built once per list (macr_amd/eval_cache.py: keyed by the list object,
    by content the first time a list is seen -- the reference's test() is stateless, so any list may arrive)"""
    def build(users):
        mask, gt = data.eval_lists(users, valid_set)
        ev = (Evaluator(mask, gt, ITEM_NUM, model.device),
              torch.tensor(list(users), dtype=torch.int32, device=model.device))
        if getattr(model, "sharded", False):         # --row_shard 1: the item table this rank holds IS its item shard
            ev[0].set_local_items(model.own_i)
        return ev
    return _evaluators.get(valid_set, test_users, build)


def test(sess, model, test_users, batch_test_flag=False, model_type='o', valid_set="test",
         item_pop_test=None, pop_exp=0):
    """Reference signature (macr_mf/train.py:162).  Ranks every user in `test_users` against the whole
    catalogue minus its train items and returns the mean precision / recall / ndcg / hit_ratio at Ks.
    sess, batch_test_flag, item_pop_test, pop_exp are accepted for compatibility."""
    if model_type not in _MODEL_TYPES:
        raise NotImplementedError("model_type %r is outside the MI355X hot path (%s)" % (model_type, sorted(_MODEL_TYPES)))
    evaluator, uid = _evaluator_for(model, test_users, valid_set)
    model.sync()
    if getattr(model, "sharded", False):
        return evaluator.test_mf(_MODEL_TYPES[model_type], model.query_rows(uid), None, model.item_embedding, Ks,
                                 model.w, model.w_user, model.rubi_c)
    return evaluator.test_mf(_MODEL_TYPES[model_type], model.user_embedding, uid, model.item_embedding, Ks,
                             model.w, model.w_user, model.rubi_c)


def test_sweep(sess, model, test_users, cs, model_type='rubi_both', valid_set="test"):
    """test() for every c of `cs` (the loop of macr_mf/tune.py:545-578) -> list of result dicts, one per c.  c only
    enters the score epilogue, so the values share the listing pass in groups of four (macr_score_topk_sweep)."""
    if model_type not in _MODEL_TYPES or _MODEL_TYPES[model_type] == ops.SCORE_NORMAL:
        raise NotImplementedError("model_type %r has no c to sweep" % model_type)
    evaluator, uid = _evaluator_for(model, test_users, valid_set)
    model.sync()
    if getattr(model, "sharded", False):
        return evaluator.test_mf_sweep(_MODEL_TYPES[model_type], model.query_rows(uid), None, model.item_embedding, Ks,
                                       model.w, model.w_user, list(cs))
    return evaluator.test_mf_sweep(_MODEL_TYPES[model_type], model.user_embedding, uid, model.item_embedding, Ks,
                                   model.w, model.w_user, list(cs))


def early_stop(hr, ndcg, recall, precision, cur_epoch, config, stopping_step, flag_step=10):
    """Patience-10 early stopping on HR with `>=` (macr_mf/train.py:313-330)."""
    if hr >= config['best_hr']:
        stopping_step = 0
        config.update(best_hr=hr, best_ndcg=ndcg, best_recall=recall, best_pre=precision, best_epoch=cur_epoch)
    else:
        stopping_step += 1
    should_stop = stopping_step >= flag_step
    if should_stop:
        print("Early stopping is trigger")
    return config, stopping_step, should_stop


def _ckpt_dir():
    return '{}_{}_checkpoint/wd_{}_lr_{}_{}/'.format(args.model, args.dataset, args.wd, args.lr, args.saveID)


def train_epoch(model, kind, n_batch, loss_log, device_sampler=None):
    """n_batch = n_train // batch_size + 1 steps (train.py:467-470).  --sampler reference follows the
    reference's python `random` stream (host-bound); --sampler device draws the batches on the GPU.
    Per-step losses stay on the device and come back once per epoch."""
    for idx in range(n_batch):
        if device_sampler is not None:
            batch = device_sampler.sample()
        else:
            users, pos_items, neg_items = data.sample()
            batch = model.to_device_batch(users, pos_items, neg_items)
        model.train_step(kind, batch, loss_log[idx], defer=True)    # the Adam pass rides under the next step
    model.sync()
    per_step = loss_log[:n_batch].cpu().numpy()
    loss = mf_loss = reg_loss = 0.
    for row in per_step:                                     # same accumulation order as train.py:497-499
        loss += row[0] / n_batch
        mf_loss += row[1] / n_batch
        reg_loss += row[2] / n_batch
    return loss, mf_loss, reg_loss


def _saved_epochs():
    """epochs of the <epoch>_ckpt.pt files of this run's checkpoint directory (train.py:588-591 naming), ascending"""
    d, out = _ckpt_dir(), []
    for f in os.listdir(d) if os.path.isdir(d) else []:
        if f.endswith('_ckpt.pt') and f[:-8].isdigit():
            out.append(int(f[:-8]))
    return sorted(out)


def main(sweep=False):
    """sweep=True is macr_mf/tune.py: evaluate np.linspace(--start, --end, --step) values of c instead of --c."""
    from macr_amd import sharding
    seed = args.seed
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    logging.basicConfig(filename="{}_{}_{}_{}".format(args.model, args.dataset, args.train, args.wd))
    if args.model != 'mf':
        raise NotImplementedError("--model %s is out of scope (mf only)" % args.model)
    if not torch.cuda.is_available():
        raise SystemExit("macr_mf/train.py needs an MI355X: the HIP path has no CPU fallback")
    dev_index = int(os.environ.get("LOCAL_RANK", args.cuda)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        # item-sharded evaluation across the node's GPUs (MACR_DIST_BACKEND=gloo: test rig, several ranks on one GPU)
        torch.distributed.init_process_group(os.environ.get("MACR_DIST_BACKEND", "nccl"))
    main_rank = sharding.is_main()           # several ranks: replicas train, ONE prints / logs / writes (rank 0)

    def say(text):
        if main_rank:
            print(text)
            logging.info(text)

    config = dict(n_users=data.n_users, n_items=data.n_items)
    if args.row_shard == 1:
        from macr_amd.mf import ShardedBPRMF
        model = ShardedBPRMF(args, config, seed=seed)
    else:
        model = BPRMF(args, config, seed=seed)
    if main_rank:
        print('MF model.')
    sess = Session(model)
    kind = model.kind_of(args.train)
    model.default_kind = kind                    # (a row-sharded model is created lazily: with THIS run's loss, whoever asks first)
    rubi_type = "rubi_both" if args.train == 'rubibceboth' else "rubi_c"        # train.py:548-556

    if args.pretrain != 0:
        # train.py:605-611: restore a saved model and evaluate it.  The reference hard-codes the file
        # (mf_<dataset>_checkpoint/wd_1e-05_lr_0.001_0/299_ckpt.ckpt); here: this run's checkpoint directory, the epoch
        # in best_epoch.txt (else the newest file), c from best_c.txt (else --c).
        epochs = _saved_epochs()
        if not epochs:
            raise SystemExit("--pretrain 1: no <epoch>_ckpt.pt under %s" % _ckpt_dir())
        pick, c = epochs[-1], args.c
        if os.path.exists(_ckpt_dir() + 'best_epoch.txt'):
            e = int(open(_ckpt_dir() + 'best_epoch.txt').read().strip() or -1)
            pick = e if e in epochs else pick
        if os.path.exists(_ckpt_dir() + 'best_c.txt'):
            c = float(open(_ckpt_dir() + 'best_c.txt').read().strip())
        print('#load existing models.')
        model.load_state_dict(torch.load(_ckpt_dir() + '{}_ckpt.pt'.format(pick), map_location=model.device))
        users_to_test = list((data.test_user_list if args.valid_set == "test" else data.valid_user_list).keys())
        if args.test == "rubi":
            model.update_c(sess, c)
            ret = test(sess, model, users_to_test, model_type=rubi_type, valid_set=args.valid_set)
        else:
            ret = test(sess, model, users_to_test, valid_set=args.valid_set)
        say('epoch %d c:%.2f recall=[%.5f, %.5f], precision=[%.5f, %.5f], hit=[%.5f, %.5f], ndcg=[%.5f, %.5f]' % (
            pick, c, ret['recall'][0], ret['recall'][-1], ret['precision'][0], ret['precision'][-1],
            ret['hit_ratio'][0], ret['hit_ratio'][-1], ret['ndcg'][0], ret['ndcg'][-1]))
        return ret

    config["best_hr"], config["best_ndcg"], config['best_recall'], config['best_pre'], config["best_epoch"] = 0, 0, 0, 0, 0
    config['best_c_hr'], config['best_c_epoch'], config['best_c'] = 0, 0, 0.0
    stopping_step = 0
    start_epoch = 0
    resumed = None                                   # the bookkeeping of the interrupted run (saved next to its checkpoint)
    if args.resume == 1:
        from macr_amd import train_state

        def latest():
            epochs = _saved_epochs()
            return (epochs[-1], _ckpt_dir() + '{}_ckpt.pt'.format(epochs[-1])) if epochs else None
        last, resumed = train_state.resume(model, latest, lambda e: _ckpt_dir() + '{}_train_state.json'.format(e))
        if last is not None:
            start_epoch = last + 1
            say('resumed from epoch %d' % last)
            if resumed:
                config.update(resumed['config'])
                stopping_step = resumed['stopping_step']
    n_batch = data.n_train // args.batch_size + 1
    loss_log = torch.zeros((n_batch, 3), dtype=torch.float32, device=model.device)
    device_sampler = None
    if args.sampler == "device":
        from macr_amd.sampler import DeviceSampler
        device_sampler = DeviceSampler(data.train_user_list, data.n_users, data.n_items, args.batch_size,
                                       model.device, seed=seed)
    elif args.sampler != "reference":
        raise SystemExit("--sampler must be reference or device")
    if device_sampler is not None and start_epoch:
        # the batch of step k is a function of (seed, k): continue the sequence instead of replaying epoch 0's batches
        device_sampler.step = resumed['sampler_step'] if resumed else start_epoch * n_batch
    # (the reference rebuilds this list at every evaluation, train.py:505-516; one object per run keeps test()'s evaluator
    # lookup at an identity check)
    users_to_test = list((data.test_user_list if args.valid_set == "test" else data.valid_user_list).keys())
    for epoch in range(start_epoch, args.epoch):
        t1 = time()
        loss, mf_loss, reg_loss = train_epoch(model, kind, n_batch, loss_log, device_sampler)
        if np.isnan(loss):
            print('ERROR: loss is nan.')
            sys.exit()
        if (epoch + 1) % args.log_interval != 0:
            if args.verbose > 0 and epoch % args.verbose == 0:
                say('Epoch %d [%.1fs]: train==[%.5f=%.5f + %.5f]' % (epoch, time() - t1, loss, mf_loss, reg_loss))
            continue

        t2 = time()
        sharding.broadcast_params(model.parameters())       # item-sharded evaluation scores ONE model (rank 0's)
        tail = ('train==[%.8f=%.8f + %.8f], recall=[%.5f, %.5f], precision=[%.5f, %.5f], hit=[%.5f, %.5f], '
                'ndcg=[%.5f, %.5f]')
        def report(head, ret):
            if args.verbose > 0:
                say(head + tail % (loss, mf_loss, reg_loss, ret['recall'][0], ret['recall'][-1],
                                   ret['precision'][0], ret['precision'][-1], ret['hit_ratio'][0],
                                   ret['hit_ratio'][-1], ret['ndcg'][0], ret['ndcg'][-1]))

        if args.test in ("normal", "rubi_user_wise"):
            ret = test(sess, model, users_to_test, valid_set=args.valid_set)
            report('Epoch %d [%.1fs + %.1fs]: ' % (epoch, t2 - t1, time() - t2), ret)
        elif args.test == "rubi":
            if main_rank:
                print('Epoch %d' % epoch)
            if kind == ops.LOSS_NORMALBCE:
                raise NotImplementedError("--test rubi needs a branch loss (--train rubibceboth | rubibce)")
            c_values = np.linspace(args.start, args.end, args.step) if sweep else [args.c]
            best = (0, 0, 0, 0, 0.0)               # train.py:540-544: bests start at 0
            # tune.py:545-578: one test() per c; here the values of a sweep share the listing pass in groups of four
            rets = test_sweep(sess, model, users_to_test, c_values, model_type=rubi_type, valid_set=args.valid_set) if sweep else None
            for k, c in enumerate(c_values):        # the best c of the sweep drives early stopping
                model.update_c(sess, c)
                ret = rets[k] if sweep else test(sess, model, users_to_test, model_type=rubi_type, valid_set=args.valid_set)
                report('c:%.2f [%.1fs + %.1fs]: ' % (c, t2 - t1, time() - t2), ret)
                if ret['hit_ratio'][0] > best[0]:
                    best = (ret['hit_ratio'][0], ret['recall'][0], ret['precision'][0], ret['ndcg'][0], c)
            ret['hit_ratio'][0], ret['recall'][0], ret['precision'][0], ret['ndcg'][0] = best[:4]
            if best[0] > config['best_c_hr']:
                config['best_c_hr'], config['best_c'], config['best_c_epoch'] = best[0], best[4], epoch
        else:
            raise NotImplementedError("--test %s" % args.test)

        config, stopping_step, should_stop = early_stop(ret['hit_ratio'][0], ret['ndcg'][0], ret['recall'][0],
                                                        ret['precision'][0], epoch, config, stopping_step)
        state = model.state_dict() if args.save_flag == 1 and (main_rank or getattr(model, "sharded", False)) else None   # (sharded: collective)
        if args.save_flag == 1 and main_rank:
            os.makedirs(_ckpt_dir(), exist_ok=True)
            torch.save(state, _ckpt_dir() + '{}_ckpt.pt'.format(epoch))
            # what --resume 1 needs besides the model: best-so-far / early-stopping state and where the samplers stand
            from macr_amd import train_state
            train_state.save(_ckpt_dir() + '{}_train_state.json'.format(epoch),
                             {'config': {k: (v.item() if hasattr(v, 'item') else v) for k, v in config.items() if k.startswith('best_')},
                              'stopping_step': int(stopping_step),
                              'sampler_step': int(device_sampler.step) if device_sampler is not None else 0})
        if should_stop and args.early_stop == 1:
            say("{} dataset best epoch{}: hr:{} ndcg:{} recall:{} precision:{}".format(
                args.dataset, config['best_epoch'], config['best_hr'], config['best_ndcg'], config['best_recall'],
                config['best_pre']))
            if main_rank:
                os.makedirs(_ckpt_dir(), exist_ok=True)
                with open(_ckpt_dir() + 'best_epoch.txt', 'w') as f:
                    print(config['best_epoch'], file=f)
                if args.test == 'rubi':
                    with open(_ckpt_dir() + 'best_c.txt', 'w') as f:
                        print(config['best_c'], file=f)
            break


if __name__ == '__main__':
    main()
