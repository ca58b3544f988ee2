"""What one call of the MF CLI's `test(sess, model, users, ...)` (macr_mf/train.py, reference signature train.py:162) costs,
measured through the CLI module itself -- not through the Evaluator object bench.py drives.

    python tools/cli_test_cost.py [--workload gowalla] [--evals 20] [--epochs_between 1]

Writes a synthetic dataset with the workload's shapes (train.txt / test.txt in the reference's format, macr_amd/synth.py's
laws) into a temporary directory, imports macr_mf/train.py with the README's command line for it, trains `epochs_between`
epochs between two timed evaluations with the device sampler (so the tables move as they do in a run), and times
`train.test(...)` end to end: wall time per call, users/s, and the share that is host work in front of the device
(evaluator lookup, argument marshalling, the graph replay call).  One JSON line on stdout.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def write_dataset(root, name, cfg, seed=4242):
    from macr_amd import synth
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    users, mask_lists, gt_lists = synth.eval_problem(cfg, seed=777)
    train = synth.interaction_lists(cfg["n_users"], cfg["n_items"], cfg["n_train"] / cfg["n_users"], seed=seed)
    for q, u in enumerate(users):                       # the query users' train lists are the evaluation problem's masks
        train[int(u)] = list(mask_lists[q])
    with open(os.path.join(d, "train.txt"), "w") as f:
        for u, items in enumerate(train):
            if u == cfg["n_users"] - 1 and (cfg["n_items"] - 1) not in items:
                items = list(items) + [cfg["n_items"] - 1]          # n_items = max id + 1 (load_data.py:104-105)
            f.write(" ".join([str(u)] + [str(int(x)) for x in items]) + "\n")
    with open(os.path.join(d, "test.txt"), "w") as f:
        for q, u in enumerate(users):
            f.write(" ".join([str(int(u))] + [str(int(x)) for x in gt_lists[q]]) + "\n")
    return len(users)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--evals", type=int, default=20)
    ap.add_argument("--epochs_between", type=int, default=1)
    a = ap.parse_args()
    import torch
    from macr_amd import synth
    cfg = synth.WORKLOADS[a.workload]
    tmp = tempfile.mkdtemp(prefix="macr_cli_")
    n_test = write_dataset(tmp, a.workload, cfg)
    os.chdir(tmp)
    sys.argv = ["train.py", "--dataset", a.workload, "--data_path", tmp + "/", "--batch_size", str(cfg["batch"]), "--cuda", "0",
                "--lr", "0.001", "--check_c", "1", "--c", str(cfg["c"]), "--train", "rubibceboth", "--test", "rubi",
                "--alpha", str(cfg["alpha"]), "--beta", str(cfg["beta"]), "--sampler", "device", "--save_flag", "0"]
    sys.path.insert(0, os.path.join(REPO, "macr_mf"))
    import train as cli                                   # parses sys.argv, loads the dataset
    from model import BPRMF, Session
    from macr_amd.sampler import DeviceSampler
    model = BPRMF(cli.args, dict(n_users=cli.data.n_users, n_items=cli.data.n_items), seed=12345)
    sess = Session(model)
    kind = model.kind_of("rubibceboth")
    model.update_c(sess, cfg["c"])
    n_batch = cli.data.n_train // cfg["batch"] + 1
    loss_log = torch.zeros((n_batch, 3), dtype=torch.float32, device=model.device)
    smp = DeviceSampler(cli.data.train_user_list, cli.data.n_users, cli.data.n_items, cfg["batch"], model.device, seed=1)
    users_to_test = list(cli.data.test_user_list.keys())

    def evaluate():
        return cli.test(sess, model, users_to_test, model_type="rubi_both")
    for _ in range(4):                                    # settle: evaluator build, graph capture, seeding policy
        cli.train_epoch(model, kind, n_batch, loss_log, smp)
        ret = evaluate()
    wall, host = [], []
    for _ in range(a.evals):
        for _e in range(a.epochs_between):
            cli.train_epoch(model, kind, n_batch, loss_log, smp)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        g0.record()
        ret = evaluate()
        g1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        wall.append(t1 - t0)
        host.append((t1 - t0) - 1e-3 * g0.elapsed_time(g1))
    ws = sorted(wall)
    out = {"what": "macr_mf/train.py::test() called as the CLI calls it (same list object every evaluation), timed end to end",
           "workload": "%s-shape synthetic dataset through the MF loader: %d users x %d items, %d test users, n_train %d"
                       % (a.workload, cli.data.n_users, cli.data.n_items, n_test, cli.data.n_train),
           "evaluations": a.evals, "train_steps_between": a.epochs_between * n_batch,
           "ms_per_test_call": 1e3 * float(np.mean(wall)), "ms_median": 1e3 * ws[len(ws) // 2], "ms_min": 1e3 * ws[0], "ms_max": 1e3 * ws[-1],
           "eval_users_per_s": n_test / float(np.mean(wall)),
           "host_gap_us": 1e6 * float(np.mean(host)),
           "evaluator_content_lookups": cli._evaluators.content_lookups,
           "metrics": {k: float(v[0]) for k, v in ret.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
