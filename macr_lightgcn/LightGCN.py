"""LightGCN command line of MACR on MI355X -- drop-in for the reference's macr_lightgcn/LightGCN.py.

    python macr_lightgcn/LightGCN.py --data_path data/ --dataset addressa --verbose 1 --layer_size [64,64] \
        --Ks [20] --loss bceboth --test rubiboth --c 40 --epoch 2000 --early_stop 1 --lr 0.001 \
        --batch_size 1024 --gpu_id 0 --log_interval 10 --alpha 1e-2 --beta 1e-3

Same flags, log line formats, early stopping and weight-file naming as the reference
(LightGCN.py:649-904).  The propagation (SpMM), the training step and the evaluator run on the HIP
kernels; the sampler is the reference's host sampler (python `random` + numpy.random streams),
overlapped with the device step because kernel launches are asynchronous (the reference needs a
helper thread for that, :567-647).  Differences: the per-log-interval "test loss" pass (:799-819)
is not run (pure logging; prints nan in the `--test normal` line), TensorBoard graph dumps are
dropped, checkpoints are torch files.
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


def train_epoch(model, kind, n_batch, loss_log, device_sampler=None):
    for idx in range(n_batch):
        if device_sampler is not None:
            batch = device_sampler.sample()
        else:
            users, pos_items, neg_items = data_generator.sample()
            batch = model.to_device_batch(users, pos_items, neg_items)
        model.train_step(kind, batch, loss_log[idx])
    per_step = loss_log[:n_batch].cpu().numpy()
    loss = mf_loss = emb_loss = 0.
    for row in per_step:
        loss += row[0] / n_batch
        mf_loss += row[1] / n_batch
        emb_loss += row[2] / n_batch
    return loss, mf_loss, emb_loss


def main(sweep=False):
    """sweep=True is LightGCN_tune.py: evaluate np.linspace(--start, --end, --step) values of c."""
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
        torch.distributed.init_process_group("nccl")
    config = dict(n_users=data_generator.n_users, n_items=data_generator.n_items)
    config['norm_adj'] = pick_adjacency(args.adj_type)
    if args.pretrain != 0:
        raise NotImplementedError("--pretrain %d restores TF weights in the reference; out of scope" % args.pretrain)
    model = LightGCN(data_config=config, pretrain_data=None, seed=seed)
    print('using xavier initialization')
    print('without pretraining.')
    sess = Session(model)
    kind = model.kind_of(args.loss)
    weights_save_path = None
    if args.save_flag == 1:
        layer = '-'.join(str(l) for l in model.weight_size)
        weights_save_path = '%sweights/%s/%s/%s/l%s_r%s' % (args.weights_path, args.dataset, model.model_type, layer,
                                                            str(args.lr), '-'.join(str(r) for r in model.regs))
        ensureDir(weights_save_path)
        os.makedirs(weights_save_path, exist_ok=True)      # tf.train.Saver created this level itself

    cur_best_pre_0, stopping_step, best_epoch, best_hr_norm, best_c_epoch = 0., 0, 0, 0, 0
    n_batch = data_generator.n_train // args.batch_size + 1
    loss_log = torch.zeros((n_batch, 3), dtype=torch.float32, device=model.device)
    device_sampler = None
    if args.sampler == "device":
        from macr_amd.sampler import DeviceSampler
        device_sampler = DeviceSampler(data_generator.train_items, data_generator.n_users, data_generator.n_items,
                                       args.batch_size, model.device, seed=seed, pool=data_generator.exist_users)
    elif args.sampler != "reference":
        raise SystemExit("--sampler must be reference or device")
    for epoch in range(1, args.epoch + 1):
        t1 = time()
        loss, mf_loss, emb_loss = train_epoch(model, kind, n_batch, loss_log, device_sampler)
        if np.isnan(loss):
            print('ERROR: loss is nan.')
            sys.exit()
        if (epoch % args.log_interval) != 0:
            if args.verbose > 0 and epoch % args.verbose == 0:
                perf_str = 'Epoch %d [%.1fs]: train==[%.5f=%.5f + %.5f]' % (epoch, time() - t1, loss, mf_loss, emb_loss)
                print(perf_str)
                logging.info(perf_str)
            continue

        t2 = time()
        users_to_test = list(data_generator.test_set.keys())
        perf_str = ''
        if args.test == 'normal':
            ret = test(sess, model, users_to_test, drop_flag=True)
            t3 = time()
            nan = float('nan')           # the reference's test-loss pass (:799-819) is not run
            if args.verbose > 0:
                perf_str = 'Epoch %d [%.1fs + %.1fs]: test==[%.5f=%.5f + %.5f + %.5f], recall=[%s], hr=[%s], ndcg=[%s]\n' % (
                    epoch, t2 - t1, t3 - t2, nan, nan, nan, 0.0,
                    ', '.join('%.5f' % r for r in ret['recall']), ', '.join('%.5f' % r for r in ret['hr']),
                    ', '.join('%.5f' % r for r in ret['ndcg']))
                print(perf_str, end='')
                logging.info(perf_str)
            if ret['hr'][0] > best_hr_norm:
                best_hr_norm, best_epoch = ret['hr'][0], epoch
        elif args.test == 'rubiboth':
            print('Epoch %d' % epoch)
            best_hr = 0
            c_values = np.linspace(args.start, args.end, args.step) if sweep else [args.c]
            for c in c_values:
                model.update_c(sess, c)
                ret = test(sess, model, users_to_test, method=args.test)
                if ret['hr'][0] > best_hr:
                    best_hr = ret['hr'][0]
                if args.verbose > 0:
                    perf_str += 'c:%.2f recall=[%.5f, %.5f], hit=[%.5f, %.5f], ndcg=[%.5f, %.5f]\n' % (
                        c, ret['recall'][0], ret['recall'][-1], ret['hr'][0], ret['hr'][-1], ret['ndcg'][0],
                        ret['ndcg'][-1])
            ret['hr'][0] = best_hr
            print(perf_str, end='')
            logging.info(perf_str)
        else:
            raise NotImplementedError("--test %s is outside the MI355X hot path (normal | rubiboth)" % args.test)

        cur_best_pre_0, stopping_step, should_stop = early_stopping(ret['hr'][0], cur_best_pre_0, stopping_step,
                                                                    expected_order='acc', flag_step=10)
        if ret['hr'][0] == cur_best_pre_0:
            best_epoch = epoch
        if args.save_flag == 1:
            torch.save(model.state_dict(), weights_save_path + '/weights_{}-{}.pt'.format(args.saveID, epoch))
            print('save the weights in path: ', weights_save_path)
        if should_stop and args.early_stop == 1:
            with open(weights_save_path + '/best_epoch_{}.txt'.format(args.saveID), 'w') as f:
                f.write(str(best_c_epoch if args.test != 'normal' else best_epoch))
            break


if __name__ == '__main__':
    main()
