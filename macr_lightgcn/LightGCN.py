"""LightGCN command line of MACR on MI355X -- drop-in for the reference's macr_lightgcn/LightGCN.py.

    python macr_lightgcn/LightGCN.py --data_path data/ --dataset addressa --verbose 1 --layer_size [64,64] \
        --Ks [20] --loss bceboth --test rubiboth --c 40 --epoch 2000 --early_stop 1 --lr 0.001 \
        --batch_size 1024 --gpu_id 0 --log_interval 10 --alpha 1e-2 --beta 1e-3

Same flags, log line formats, early stopping and weight-file naming as the reference
(LightGCN.py:649-904).  The propagation (SpMM), the training step and the evaluator run on the HIP
kernels; the sampler is the reference's host sampler (python `random` + numpy.random streams),
overlapped with the device step because kernel launches are asynchronous (the reference needs a
helper thread for that, :567-647).  The per-log-interval "test loss" pass (:799-819) runs as
loss-only steps (MACR_STEP_LOSS_ONLY) on sample_test() batches, so the host RNG streams stay aligned
with the reference for a whole run.  Differences: TensorBoard graph dumps are dropped; checkpoints are
torch files (same directory and file naming), which --pretrain 1 and the additive --resume 1 read back;
under torch.distributed.run every rank trains a replica, rank 0 prints / logs / saves, and the
evaluation is item-sharded over the ranks after a broadcast of rank 0's parameters.
"""
import logging
import os
import random
import sys
from time import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from utility.helper import early_stopping, ensureDir          # noqa: E402
from utility.batch_test import *                              # noqa: E402,F401,F403  (args, data_generator, test ...)
from macr_amd.lightgcn import LightGCN as _LightGCN           # noqa: E402
from macr_amd.mf import Session                               # noqa: E402

logging.getLogger().setLevel(logging.INFO)


class LightGCN(_LightGCN):
    """LightGCN(data_config, pretrain_data) as in the reference (:32-33); flags come from `args`."""

    def __init__(self, data_config, pretrain_data=None, **kw):
        super(LightGCN, self).__init__(data_config, args, pretrain_data=pretrain_data, **kw)


def pick_adjacency(adj_type):
    plain_adj, norm_adj, mean_adj, pre_adj = data_generator.get_adj_mat()
    if adj_type == 'plain':
        print('use the plain adjacency matrix')
        return plain_adj
    if adj_type == 'norm':
        print('use the normalized adjacency matrix')
        return norm_adj
    if adj_type == 'gcmc':
        print('use the gcmc adjacency matrix')
        return mean_adj
    if adj_type == 'pre':
        print('use the pre adjcency matrix')
        return pre_adj
    print('use the mean adjacency matrix')
    return mean_adj + sp.eye(mean_adj.shape[0])


def train_epoch(model, kind, n_batch, loss_log, device_sampler=None, test_loss=False):
    """n_batch steps (LightGCN.py:765-790).  test_loss=True is the reference's second pass of n_batch loss-only
    runs on data_generator.sample_test() batches (:799-819), no parameter update (with a device_sampler: that
    sampler's batches -- `--sampler device` builds one over the test lists).
    RNG consumption with the host sampler: the reference fetches one batch ahead (:762-764 / :799-801, then one
    `sample_thread` per iteration :767 / :804), so a pass draws n_batch + 1 batches and throws the last one away; the
    extra draw below keeps the python `random` / numpy streams -- and so every later epoch's batches -- the same."""
    for idx in range(n_batch):
        if test_loss and device_sampler is not None:
            batch = device_sampler.sample()
        elif test_loss:
            users, pos_items, neg_items = data_generator.sample_test()
            batch = model.to_device_batch(users, pos_items, neg_items)
        elif device_sampler is not None:
            batch = device_sampler.sample()
        else:
            users, pos_items, neg_items = data_generator.sample()
            batch = model.to_device_batch(users, pos_items, neg_items)
        model.train_step(kind, batch, loss_log[idx], loss_only=test_loss)
    if device_sampler is None:                                  # the reference's discarded look-ahead batch
        data_generator.sample_test() if test_loss else data_generator.sample()
    per_step = loss_log[:n_batch].cpu().numpy()
    loss = mf_loss = emb_loss = 0.
    for row in per_step:
        loss += row[0] / n_batch
        mf_loss += row[1] / n_batch
        emb_loss += row[2] / n_batch
    return loss, mf_loss, emb_loss


def _weights_dir(model):
    layer = '-'.join(str(l) for l in model.weight_size)
    return '%sweights/%s/%s/%s/l%s_r%s' % (args.weights_path, args.dataset, model.model_type, layer, str(args.lr),
                                          '-'.join(str(r) for r in model.regs))


def _saved_epochs(path):
    """epochs of the weights_<saveID>-<epoch>.pt files in `path` (LightGCN.py:892 naming), ascending"""
    pre, out = 'weights_{}-'.format(args.saveID), []
    for f in os.listdir(path) if os.path.isdir(path) else []:
        if f.startswith(pre) and f.endswith('.pt') and f[len(pre):-3].isdigit():
            out.append(int(f[len(pre):-3]))
    return sorted(out)


def main(sweep=False):
    """sweep=True is LightGCN_tune.py: evaluate np.linspace(--start, --end, --step) values of c."""
    from macr_amd import sharding
    seed = args.seed
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    logging.basicConfig(filename="LightGCN_{}_{}_{}_{}".format(args.dataset, args.loss, args.test, args.alpha))
    if not torch.cuda.is_available():
        raise SystemExit("macr_lightgcn/LightGCN.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", args.gpu_id)) % max(torch.cuda.device_count(), 1))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(os.environ.get("MACR_DIST_BACKEND", "nccl"))
    main_rank = sharding.is_main()           # several ranks: replicas train, ONE prints / logs / writes (rank 0)

    def say(text, end='\n'):
        if main_rank:
            print(text, end=end)
            logging.info(text)

    config = dict(n_users=data_generator.n_users, n_items=data_generator.n_items)
    config['norm_adj'] = pick_adjacency(args.adj_type)
    model = LightGCN(data_config=config, pretrain_data=None, seed=seed)
    sess = Session(model)
    kind = model.kind_of(args.loss)
    weights_save_path = _weights_dir(model)

    if args.pretrain == 1:
        # LightGCN.py:707-719: restore a saved model, evaluate c = 0 and the best c, exit.  The reference hard-codes
        # the file ("Your Path") and best_c = 45; here: the epoch in best_epoch_<saveID>.txt (else the newest file) and --c.
        epochs = _saved_epochs(weights_save_path)
        if not epochs:
            raise SystemExit("--pretrain 1: no weights_%s-<epoch>.pt under %s" % (args.saveID, weights_save_path))
        best_file = weights_save_path + '/best_epoch_{}.txt'.format(args.saveID)
        pick = epochs[-1]
        if os.path.exists(best_file):
            e = int(open(best_file).read().strip() or -1)
            pick = e if e in epochs else pick
        model.load_state_dict(torch.load(weights_save_path + '/weights_{}-{}.pt'.format(args.saveID, pick),
                                         map_location=model.device))
        users_to_test = list(data_generator.test_set.keys())
        for c in [0, args.c]:
            model.update_c(sess, c)
            ret = test(sess, model, users_to_test, method="rubiboth")
            say('c:{}: recall={}, hit={}, ndcg={}'.format(c, str(ret["recall"]), str(ret['hr']), str(ret['ndcg'])))
        return
    if args.pretrain != 0:
        # the reference's -1 / -2 modes load embeddings of another model (LightGCN.py:665-693): outside this path
        raise SystemExit("--pretrain %d is not supported (0: train, 1: evaluate saved weights)" % args.pretrain)
    print('using xavier initialization')
    print('without pretraining.')
    start_epoch = 1
    cur_best_pre_0, stopping_step, best_epoch, best_hr_norm, best_c_epoch, best_c_hr = 0., 0, 0, 0, 0, 0.
    resumed = None                                   # the bookkeeping of the interrupted run (saved next to its weights)
    if args.resume == 1:
        from macr_amd import train_state

        def latest():
            epochs = _saved_epochs(weights_save_path)
            return (epochs[-1], weights_save_path + '/weights_{}-{}.pt'.format(args.saveID, epochs[-1])) if epochs else None
        last, resumed = train_state.resume(model, latest,
                                           lambda e: weights_save_path + '/train_state_{}-{}.json'.format(args.saveID, e))
        if last is not None:
            start_epoch = last + 1
            say('resumed from epoch %d' % last)
            if resumed:
                cur_best_pre_0, stopping_step, best_epoch, best_hr_norm, best_c_epoch, best_c_hr = resumed['bests']

    n_batch = data_generator.n_train // args.batch_size + 1
    loss_log = torch.zeros((n_batch, 3), dtype=torch.float32, device=model.device)
    device_sampler = test_sampler = None
    if args.sampler == "device":
        from macr_amd.sampler import DeviceSampler
        device_sampler = DeviceSampler(data_generator.train_items, data_generator.n_users, data_generator.n_items,
                                       args.batch_size, model.device, seed=seed, pool=data_generator.exist_users)
        # sample_test (utility/load_data.py:214-254): users with test items, a positive from the user's TEST list, a
        # negative outside her test and train lists
        test_users = sorted(data_generator.test_set.keys())
        both = {u: list(data_generator.test_set[u]) + list(data_generator.train_items.get(u, [])) for u in test_users}
        test_sampler = DeviceSampler({u: sorted(data_generator.test_set[u]) for u in test_users}, data_generator.n_users,
                                     data_generator.n_items, args.batch_size, model.device, seed=seed + 1, pool=test_users,
                                     exclude=both)
    elif args.sampler != "reference":
        raise SystemExit("--sampler must be reference or device")
    if device_sampler is not None and start_epoch > 1:
        # the batch of step k is a function of (seed, k): continue the sequences instead of replaying epoch 1's batches
        device_sampler.step = resumed['sampler_step'] if resumed else (start_epoch - 1) * n_batch
        test_sampler.step = resumed['test_sampler_step'] if resumed else ((start_epoch - 1) // args.log_interval) * n_batch
    users_to_test = list(data_generator.test_set.keys())      # one list object per run: test() finds its evaluator by identity
    for epoch in range(start_epoch, args.epoch + 1):
        t1 = time()
        loss, mf_loss, emb_loss = train_epoch(model, kind, n_batch, loss_log, device_sampler)
        if np.isnan(loss):
            print('ERROR: loss is nan.')
            sys.exit()
        if (epoch % args.log_interval) != 0:
            if args.verbose > 0 and epoch % args.verbose == 0:
                say('Epoch %d [%.1fs]: train==[%.5f=%.5f + %.5f]' % (epoch, time() - t1, loss, mf_loss, emb_loss))
            continue

        # the reference's "test loss" pass (:799-819): n_batch loss-only runs on sample_test() batches
        loss_test, mf_loss_test, emb_loss_test = train_epoch(model, kind, n_batch, loss_log, test_sampler, test_loss=True)
        t2 = time()
        sharding.broadcast_params(model.parameters())       # item-sharded evaluation scores ONE model (rank 0's)
        perf_str = ''
        if args.test == 'normal':
            ret = test(sess, model, users_to_test, drop_flag=True)
            t3 = time()
            if args.verbose > 0:
                perf_str = 'Epoch %d [%.1fs + %.1fs]: test==[%.5f=%.5f + %.5f + %.5f], recall=[%s], hr=[%s], ndcg=[%s]\n' % (
                    epoch, t2 - t1, t3 - t2, loss_test, mf_loss_test, emb_loss_test, 0.0,
                    ', '.join('%.5f' % r for r in ret['recall']), ', '.join('%.5f' % r for r in ret['hr']),
                    ', '.join('%.5f' % r for r in ret['ndcg']))
                say(perf_str, end='')
            if ret['hr'][0] > best_hr_norm:
                best_hr_norm, best_epoch = ret['hr'][0], epoch
        elif args.test == 'rubiboth':
            if main_rank:
                print('Epoch %d' % epoch)
            best_hr = 0
            c_values = np.linspace(args.start, args.end, args.step) if sweep else [args.c]
            rets = test_sweep(sess, model, users_to_test, c_values, method=args.test) if sweep else None
            for k, c in enumerate(c_values):
                model.update_c(sess, c)
                ret = rets[k] if sweep else test(sess, model, users_to_test, method=args.test)
                if ret['hr'][0] > best_hr:
                    best_hr = ret['hr'][0]
                if args.verbose > 0:
                    perf_str += 'c:%.2f recall=[%.5f, %.5f], hit=[%.5f, %.5f], ndcg=[%.5f, %.5f]\n' % (
                        c, ret['recall'][0], ret['recall'][-1], ret['hr'][0], ret['hr'][-1], ret['ndcg'][0],
                        ret['ndcg'][-1])
            ret['hr'][0] = best_hr
            if best_hr > best_c_hr:                      # config['best_c_epoch'] of the reference (:880)
                best_c_hr, best_c_epoch = best_hr, epoch
            say(perf_str, end='')
        else:
            raise NotImplementedError("--test %s is outside the MI355X hot path (normal | rubiboth)" % args.test)

        cur_best_pre_0, stopping_step, should_stop = early_stopping(ret['hr'][0], cur_best_pre_0, stopping_step,
                                                                    expected_order='acc', flag_step=10)
        if ret['hr'][0] == cur_best_pre_0:
            best_epoch = epoch
        if args.save_flag == 1 and main_rank:
            ensureDir(weights_save_path)
            os.makedirs(weights_save_path, exist_ok=True)      # tf.train.Saver created this level itself
            torch.save(model.state_dict(), weights_save_path + '/weights_{}-{}.pt'.format(args.saveID, epoch))
            # what --resume 1 needs besides the model: best-so-far / early-stopping state and where the samplers stand
            from macr_amd import train_state
            train_state.save(weights_save_path + '/train_state_{}-{}.json'.format(args.saveID, epoch),
                             {'bests': [float(cur_best_pre_0), int(stopping_step), int(best_epoch), float(best_hr_norm),
                                        int(best_c_epoch), float(best_c_hr)],
                              'sampler_step': int(device_sampler.step) if device_sampler is not None else 0,
                              'test_sampler_step': int(test_sampler.step) if test_sampler is not None else 0})
            print('save the weights in path: ', weights_save_path)
        if should_stop and args.early_stop == 1:
            if main_rank:
                os.makedirs(weights_save_path, exist_ok=True)
                with open(weights_save_path + '/best_epoch_{}.txt'.format(args.saveID), 'w') as f:
                    f.write(str(best_c_epoch if args.test != 'normal' else best_epoch))
            break


if __name__ == '__main__':
    main()
