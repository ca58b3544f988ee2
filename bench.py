#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X:
   "train interactions/sec + eval users/sec (full-catalog top-K@20), 1/2/4/8 GPU".

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the training hot path (gather + dots + counterfactual
branches + (B,B) BCE + gradients + TF-style dense Adam) over one batch of B
synthetic (u, i+, i-) triples that already live in HBM.  Default workload =
BASELINE.json configs[1]: Gowalla shapes, MACR-MF `rubibceboth`, d=64, B=4096,
c=40 (the other configs are parity-test cases, selectable with --workload).
`value` = training interactions/s of the whole job; the evaluator (full
catalogue, train-masked, top-20, metrics) is timed in the same run and reported
as `eval_users_per_s` with its own MFMA roofline.

Multi-GPU: the training step of these configs fits one GPU and the
(B,B) loss couples every pair of a batch, so N GPUs run N independent replicas
("replicas only", no data-path collective, weak scaling); the evaluator shards
the item catalogue across the ranks and exchanges per-shard top-K with one RCCL
all-gather.

Batches arrive as a sampler emits them (NOT ordered by item): the step groups its
batch on the device, inside the timed region.  The K-step region is repeated
(each repetition bracketed by barrier + synchronize, exactly K steps) until about
60 ms have been timed and the MEDIAN repetition is reported: 20 steps are < 1 ms.

Rank 0 prints ONE JSON line.  Extra objects: `roofline` = the launch with the
largest time per step inside the timed region (HIP-event timed on the launch
stream; for the default workload the (B,B) launch that carries the dense Adam
pass), `roofline_step` (whole step, SURVEY.md 8d bytes / ms_per_step),
`roofline_eval`, `end_to_end` (device sampler feeding the step), `kernels`,
`cpu_baseline` (the CPU oracle = "port": the un-tuned checker, timed on this
box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="gowalla", choices=["addressa", "gowalla", "ml10m", "yelp2018"])
    ap.add_argument("--train", default="rubibceboth", choices=["normalbce", "rubibceboth"])
    ap.add_argument("--eval-reps", type=int, default=20, help="timed evaluations (more than one period of the seeding policy's back-off)")
    ap.add_argument("--eval-settle", type=int, default=6, help="untimed evaluations (with the training steps between them) before the timed ones")
    ap.add_argument("--eval-train-steps", type=int, default=20,
                    help="untimed training steps between two timed evaluations (the evaluator seeds its thresholds with the "
                         "previous evaluation's ranking, so the tables must move as they do in a training run)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pos", default="zipf", choices=["zipf", "uniform"], help="positive-item distribution (diagnostic)")
    ap.add_argument("--no-eval", action="store_true", help="skip the evaluator leg (diagnostic)")
    ap.add_argument("--regions", type=int, default=0, help="number of timed K-step regions (0 = as many as fit in ~60 ms); the median is reported")
    ap.add_argument("--no-e2e", action="store_true", help="skip the sampler-inclusive end-to-end leg (diagnostic)")
    ap.add_argument("--no-defer", action="store_true", help="complete every step's Adam pass inside the step instead of under the next step's (B,B) kernel (diagnostic)")
    ap.add_argument("--presorted", action="store_true", help="feed batches already ordered by positive item (diagnostic; the step orders its batch on the device either way)")
    return ap.parse_args()


def algorithmic_bytes(kernel, cfg, B):
    """per-launch algorithmic HBM bytes of each kernel (fp32, int32)."""
    d, rows = cfg["d"], cfg["n_users"] + cfg["n_items"]
    if kernel in ("adam_dense", "bxb+adam"):
        # the dense Adam pass reads+writes theta,m,v of EVERY row, stand-alone or as the Adam blocks of the (B,B)
        # launch (the (B,B) part itself moves no HBM bytes)
        return 24 * d * rows
    if kernel == "pair_fwd":
        return B * (12 * d + 12)                     # 3 rows + 3 indices read
    if kernel in ("pair_bwd", "pair_normal"):
        return B * (24 * d + 12)                     # 3 rows read + 3 gradient rows written + indices
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    # one process per GPU over RCCL; MACR_DIST_BACKEND=gloo lets several ranks share one GPU (test rig for the N>1 path)
    backend = os.environ.get("MACR_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    from macr_amd import ops, sharding, synth
    from macr_amd.evaluator import Evaluator

    cfg = synth.WORKLOADS[args.workload]
    B, d = cfg["batch"], cfg["d"]
    kind = ops.LOSS_RUBIBCEBOTH if args.train == "rubibceboth" else ops.LOSS_NORMALBCE
    gen = torch.Generator(device=dev).manual_seed(12345 + rank)
    P = synth.xavier_table(cfg["n_users"], d, gen, dev)
    Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
    w = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    hyper = ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
    state = ops.MFState(P, Q, w, wu, hyper, B)
    n_batches = min(args.steps + args.warmup, 256)
    batches = synth.train_batches(n_batches, cfg["n_users"], cfg["n_items"], B, gen, dev, zipf=args.pos == "zipf",
                                  sort_by_pos=args.presorted)
    loss_log = torch.zeros((n_batches, 3), dtype=torch.float32, device=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def run_steps(n, first):
        for s in range(n):
            k = (first + s) % n_batches
            state.step(kind, batches[k, 0], batches[k, 1], batches[k, 2], loss_log[k], defer=not args.no_defer)
        state.flush()          # inside every timed region: all parameter updates are complete when the clock stops

    # ------------------------------------------------------------- training: W warmup + exactly K timed steps
    run_steps(args.warmup, 0)

    def timed_region():
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, args.warmup)
        torch.cuda.synchronize(); barrier()
        return sharding.max_over_ranks(time.perf_counter() - t0, dev)

    regions = [timed_region()]
    # a 20-step region is under a millisecond: repeat the SAME K-step region until ~60 ms are on the clock, report the median
    n_rep = args.regions if args.regions > 0 else int(min(200, max(1, round(0.06 / max(regions[0], 1e-6)))))
    n_rep = int(sharding.max_over_ranks(float(n_rep), dev))
    regions += [timed_region() for _ in range(n_rep - 1)]
    elapsed = float(np.median(regions))
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    losses = loss_log.cpu().numpy()
    if not np.isfinite(losses[: min(n_batches, args.steps)]).all():
        raise SystemExit("ERROR: loss is nan.")

    # ------------------------------------------------------------- per-kernel HIP-event timing (same stream)
    # Recording an event after every launch adds a fixed cost to each event-to-event interval (the un-instrumented
    # K-step region above runs the same kernels back to back).  It is calibrated from the two measurements bench.py
    # already has -- sum of event intervals per step vs. wall time per step -- and removed uniformly per launch:
    # "event_us" is the raw interval, "avg_us" the calibrated duration (what rocprofv3 --kernel-trace reports).
    n_prof = 20
    ops.timing_begin()
    run_steps(n_prof, 0)
    marks = ops.timing_end(max_n=n_prof * 8 + 8)
    # The Adam pass on its own (complete steps, flags=0): in deferred mode it only runs stand-alone at the flush.
    alone = []
    if kind == ops.LOSS_RUBIBCEBOTH and not args.no_defer:
        ops.timing_begin()
        for s_ in range(10):
            k_ = s_ % n_batches
            state.step(kind, batches[k_, 0], batches[k_, 1], batches[k_, 2], loss_log[k_], defer=False)
        alone = [ms for name, ms in ops.timing_end(max_n=64) if name == "adam_dense"]
    kernels = {}
    for name, ms in marks:
        k = kernels.setdefault(name, [0, 0.0])
        k[0] += 1; k[1] += ms
    kern_avg = {n: {"launches_per_step": c / n_prof, "event_us": 1e3 * t / c} for n, (c, t) in kernels.items()}
    event_sum_us = sum(v["event_us"] * v["launches_per_step"] for v in kern_avg.values())
    launches = sum(v["launches_per_step"] for v in kern_avg.values())
    event_overhead_us = max(0.0, (event_sum_us - 1e3 * ms_per_step) / launches)
    for v in kern_avg.values():
        v["avg_us"] = max(v["event_us"] - event_overhead_us, 0.1)
    step_kernel_us = sum(v["avg_us"] * v["launches_per_step"] for v in kern_avg.values())
    pmc = {}
    pmc_path = os.path.join(REPO, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
        except Exception:
            pmc = {}
    for n, v in kern_avg.items():
        ab = algorithmic_bytes(n, cfg, B)
        if ab:
            v["algorithmic_bytes"] = ab
            v["GBps"] = ab / (v["avg_us"] * 1e-6) / 1e9
        if n == "bxb":
            v["gevals_per_s"] = 2.0 * B * B / (v["avg_us"] * 1e-6) / 1e9   # fused-BCE element evaluations
    # `roofline`: the launch with the largest time per step INSIDE the timed region, whatever it is
    per_step_us = {n: v["avg_us"] * v["launches_per_step"] for n, v in kern_avg.items()}
    dom = max(per_step_us, key=per_step_us.get)

    def roof(name, avg_us, ab):
        gbps = ab / (avg_us * 1e-6) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbps / HBM_PEAK_GBS, "traffic": pmc.get(args.workload, {}).get(name), "avg_us": avg_us,
                "us_per_step": per_step_us.get(name), "algorithmic_bytes": ab}
    if "algorithmic_bytes" in kern_avg[dom]:
        roofline = roof(dom, kern_avg[dom]["avg_us"], kern_avg[dom]["algorithmic_bytes"])
        if dom == "bxb+adam":
            roofline["note"] = ("the (B,B) launch with the dense Adam pass of the previous step riding in it: HBM bytes = "
                                "that pass (24*d*rows); the launch also does the 2*B^2 fused-BCE evaluations, which move "
                                "no HBM bytes (VALU-bound: kernels['bxb'] is the same launch without the Adam blocks)")
    else:   # a launch without HBM work dominates (e.g. non-deferred bxb): report it as such, no bandwidth claim
        roofline = {"kernel": dom, "bound": "valu", "achieved": kern_avg[dom].get("gevals_per_s"), "peak": None,
                    "unit": "G fused-BCE evaluations/s", "frac": None, "traffic": pmc.get(args.workload, {}).get(dom),
                    "avg_us": kern_avg[dom]["avg_us"], "us_per_step": per_step_us[dom]}
    # the whole step against the HBM roofline: SURVEY.md 8(d) B*(24d+12) + 24d*(n_users+n_items) bytes per step
    step_bytes = B * (24 * d + 12) + 24 * d * (cfg["n_users"] + cfg["n_items"])
    roofline_step = {"bound": "hbm", "algorithmic_bytes": step_bytes, "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
    aux = {}
    if alone:   # the Adam pass where it runs alone (complete steps), for comparison with the riding pass
        us = max(1e3 * sum(alone) / len(alone) - event_overhead_us, 0.1)
        aux["adam_dense_alone"] = roof("adam_dense", us, algorithmic_bytes("adam_dense", cfg, B))
        aux["adam_dense_alone"]["launches_sampled"] = len(alone)

    # ------------------------------------------------------------- end to end: the device sampler feeds the step
    end_to_end = None
    if not args.no_e2e:
        from macr_amd.sampler import DeviceSampler
        lists = synth.interaction_lists(cfg["n_users"], cfg["n_items"], cfg["n_train"] / cfg["n_users"], seed=4242 + rank)
        smp = DeviceSampler(lists, cfg["n_users"], cfg["n_items"], B, dev, seed=99 + rank)

        def run_e2e(n):
            for s_ in range(n):
                buf = smp.sample()                       # same stream; one launch draws the next 32 batches (macr_sample_triples_many)
                state.step(kind, buf[0], buf[1], buf[2], loss_log[s_ % n_batches], defer=not args.no_defer)
            state.flush()
        run_e2e(args.warmup)
        e2e = []
        for _ in range(max(1, min(n_rep, 50))):
            barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_e2e(args.steps)
            torch.cuda.synchronize(); barrier()
            e2e.append(sharding.max_over_ranks(time.perf_counter() - t0, dev))
        t_e2e = float(np.median(e2e))
        end_to_end = {"interactions_per_s": world * B * args.steps / t_e2e, "ms_per_step": 1e3 * t_e2e / args.steps,
                      "sampler": "device (macr_sample_triples_many: users without replacement, uniform positive of the user's "
                                 "train list, rejection-sampled negative), one launch per 32 steps on the step's stream",
                      "host_sampler_note": "--sampler reference keeps the reference's python stream at ~1 M triples/s (host bound)"}

    # ------------------------------------------------------------- evaluator: full catalogue, masked, top-20 + metrics
    users, mask_lists, gt_lists = synth.eval_problem(cfg, seed=777)       # same on every rank
    Ks = [20]
    eval_users_per_s = ev_elapsed = ev_unseeded_ms = ev_modes = None
    ret, roofline_eval = {}, None
    if not args.no_eval:
        def one_model():
            if world > 1:      # replicas trained on different batches: evaluate ONE model (rank 0's), item-sharded
                for t in (state.P, state.Q, state.w, state.wu):
                    torch.distributed.broadcast(t, 0)
        one_model()
        ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
        uid = torch.from_numpy(users).to(dev)

        def run_eval():
            return ev.test_mf(ops.SCORE_RUBI_BOTH, state.P, uid, state.Q, Ks, state.w, state.wu, cfg["c"])
        ret = run_eval()                 # first evaluation: thresholds from a sampling pass (no previous ranking to seed from)
        # a training run evaluates hundreds of times: let the evaluator's seeding policy see a few evaluations of THIS model
        # (tables moving as below) before the clock starts -- it backs off from seeds that keep going stale
        for r_ in range(args.eval_settle):
            if args.eval_train_steps > 0:
                run_steps(args.eval_train_steps, args.eval_train_steps * r_)
                one_model()
            ret = run_eval()
            torch.cuda.synchronize()
        # Timed evaluations: as in a training run, the tables MOVE between two evaluations (20 untimed training steps
        # here), and an evaluation seeds its thresholds with the ids the previous one returned (Evaluator.rank_local).
        ev_elapsed, ev_modes = 0.0, []
        for r_ in range(args.eval_reps):
            if args.eval_train_steps > 0:
                run_steps(args.eval_train_steps, args.eval_train_steps * r_)
                one_model()
            barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = run_eval()
            torch.cuda.synchronize(); barrier()
            ev_elapsed += sharding.max_over_ranks(time.perf_counter() - t0, dev)
            st_ = ev._stats.tolist()
            ev_modes.append({"seeded": bool(ev._last_seeded), "query_blocks_relisted": st_[0], "exact_fallback": st_[1]})
        eval_users_per_s = len(users) * args.eval_reps / ev_elapsed
        # the same evaluation without seeds (what a first evaluation costs: sampling pass + k_tau instead of k_tau_seed),
        # also a graph replay
        ev.use_seeds = False
        run_eval(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run_eval(); torch.cuda.synchronize()
        ev_unseeded_ms = 1e3 * (time.perf_counter() - t0)
        if args.eval_train_steps > 0:
            run_steps(args.eval_train_steps, 0)
            one_model()
        # per-kernel events need the launches themselves, not the graph replay.  The SAMPLED sequence (what a first
        # evaluation runs, and what the policy falls back to) is the one `roofline_eval` prices; a seeded attempt on
        # the same tables is reported beside it with what happened to it (roofline_eval.seeded).
        ev.use_graph = False
        ops.timing_begin()
        run_eval()
        emarks = ops.timing_end()
        ev.use_seeds = True
        ev._seed_skip = 0
        ops.timing_begin()
        run_eval()
        smarks = ops.timing_end()
        seeded_run = {"seeded": bool(ev._last_seeded), "query_blocks_relisted": ev._stats.tolist()[0],
                      "kernels_us": {}}
        for name, ms in smarks:
            seeded_run["kernels_us"][name] = seeded_run["kernels_us"].get(name, 0.0) + 1e3 * max(ms - 1e-3 * event_overhead_us, 0.0)
        ev.use_graph = True
        ev_kernel_mode = {"seeded": False}
        ek = {}
        for name, ms in emarks:
            ek[name] = ek.get(name, 0.0) + max(ms - 1e-3 * event_overhead_us, 0.0)
        lo, hi = sharding.item_shard_range(cfg["n_items"], rank, world)
        flops = 2.0 * len(users) * (hi - lo) * d
        # The ranking = sample pass + tau + listing pass + select (+ the fallback launch that returns at once):
        # `achieved` counts the catalogue's U*N*d multiply-adds ONCE over the time of all of them (the sample pass
        # re-multiplies 1/8 of the tiles; that is overhead, not work).  "stream" is the listing pass alone.
        rank_kernels = ("score_sample", "tau", "tau_seed", "score_stream", "select", "repair_plan", "score_sample2", "tau2",
                        "score_stream2", "select2", "score_topk")
        st_us = 1e3 * sum(ek.get(k, 0.0) for k in rank_kernels)
        stream_us = 1e3 * ek.get("score_stream", float("nan"))
        roofline_eval = {"kernel": "+".join(k for k in rank_kernels if k in ek), "bound": "mfma",
                         "achieved": flops / (st_us * 1e-6) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "avg_us": st_us, "flops": flops, "traffic": pmc.get(args.workload, {}).get("score_stream"),
                         "stream": {"avg_us": stream_us, "achieved": flops / (stream_us * 1e-6) / 1e12,
                                    "frac": flops / (stream_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS},
                         "kernels_us": {k: 1e3 * v for k, v in ek.items()}, "mode": ev_kernel_mode}
        roofline_eval["frac"] = roofline_eval["achieved"] / MFMA_F32_PEAK_TFLOPS
        s_us = sum(seeded_run["kernels_us"].get(k, 0.0) for k in rank_kernels)
        seeded_run.update({"avg_us": s_us, "frac": flops / (s_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                           "note": "one seeded ranking of the same tables (seeds = best candidates of the previous evaluation, "
                                   "--eval-train-steps older); query_blocks_relisted > 0: the seeds were stale and the repair "
                                   "round ran"})
        roofline_eval["seeded"] = seeded_run

    # ------------------------------------------------------------- CPU baseline: the oracle ("port") on the host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        threads = os.cpu_count() or 1
        Pc, Qc = P.cpu().numpy().copy(), Q.cpu().numpy().copy()
        wc, wuc = w.cpu().numpy().copy(), wu.cpu().numpy().copy()
        st = oracle.AdamState([Pc.shape, Qc.shape, (d,), (d,)])
        hb = batches[:8].cpu().numpy()
        oracle.mf_train_step(kind, hb[0, 0], hb[0, 1], hb[0, 2], Pc, Qc, wc, wuc, st, cfg["lr"], cfg["regs"],
                             cfg["alpha"], cfg["beta"], B)
        t0, n_cpu = time.perf_counter(), 0
        while time.perf_counter() - t0 < args.cpu_seconds * 0.6 and n_cpu < 64:
            k = (n_cpu + 1) % 8
            oracle.mf_train_step(kind, hb[k, 0], hb[k, 1], hb[k, 2], Pc, Qc, wc, wuc, st, cfg["lr"], cfg["regs"],
                                 cfg["alpha"], cfg["beta"], B)
            n_cpu += 1
        cpu_train = n_cpu * B / (time.perf_counter() - t0)
        n_eval_cpu = min(256, len(users))
        mptr, midx = oracle.csr_from_lists(mask_lists[:n_eval_cpu])
        gptr, gidx = oracle.csr_from_lists(gt_lists[:n_eval_cpu])
        t0 = time.perf_counter()
        sig_i = oracle.branch_sigmoid(Qc, wc)
        sig_u = oracle.branch_sigmoid(Pc[users[:n_eval_cpu]], wuc)
        _, oi, oc = oracle.score_topk(oracle.SCORE_RUBI_BOTH, Pc[users[:n_eval_cpu]], Qc, 20, sig_u, sig_i, cfg["c"],
                                      (mptr, midx))
        oracle.metrics_mf(oi, oc, (gptr, gidx), Ks)
        cpu_eval = n_eval_cpu / (time.perf_counter() - t0)
        cpu = {"value": cpu_train, "unit": "interactions/s", "cores": threads, "kind": "port",
               "what": "oracle (un-tuned checker: serial gather/scatter, a calloc of the full gradient tables per step); "
                       "a statement about the checker, not about what a tuned CPU implementation could do",
               "sample": "%d training steps of the same workload (B=%d) on the CPU restatement of the reference "
                         "path (oracle/macr_oracle.c, OpenMP); eval: %d of the %d query users"
                         % (n_cpu, B, n_eval_cpu, len(users)),
               "eval_users_per_s": cpu_eval}

    if rank == 0:
        out = {
            "metric": "train interactions/sec + eval users/sec (full-catalog top-K@20)",
            "value": value, "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shape MACR-MF %s d=%d batch=%d c=%g (n_users=%d, n_items=%d); synthetic "
                                   "Xavier tables, Zipf positives, batches as sampled (grouped on the device inside the step)" % (args.workload, args.train, d, B, cfg["c"],
                                                                      cfg["n_users"], cfg["n_items"]),
                       "parallelism": "replicas x%d (train) / item-sharded x%d + RCCL all-gather (eval)" % (world, world),
                       "global_batch": B * world},
            "eval_users_per_s": eval_users_per_s,
            "eval_ms_per_pass": None if ev_elapsed is None else 1e3 * ev_elapsed / args.eval_reps,
            "eval_note": None if ev_elapsed is None else "graph replays; the tables move by --eval-train-steps (%d) untimed training steps "
                         "between two timed evaluations; the evaluator seeds its thresholds with the previous evaluation's best "
                         "candidates unless those went stale last time (eval_modes: what each timed evaluation did); "
                         "eval_ms_unseeded = the same evaluation with the sampling pass instead" % args.eval_train_steps,
            "eval_modes": None if ev_elapsed is None else ev_modes,
            "eval_ms_unseeded": None if ev_elapsed is None else ev_unseeded_ms,
            "eval_users": len(users), "eval_metrics": {k: float(v[0]) for k, v in ret.items()},
            "timed_regions": {"n": len(regions), "each": "exactly %d steps, barrier+synchronize on both sides" % args.steps,
                              "reported": "median", "min_ms_per_step": 1e3 * min(regions) / args.steps,
                              "max_ms_per_step": 1e3 * max(regions) / args.steps},
            "step_kernel_us": step_kernel_us, "event_overhead_us_per_launch": event_overhead_us, "kernels": kern_avg,
            "roofline": roofline, "roofline_step": roofline_step, "roofline_aux": aux, "end_to_end": end_to_end,
            "roofline_eval": roofline_eval, "cpu_baseline": cpu,
            "last_losses": [float(x) for x in losses[(args.warmup + args.steps - 1) % n_batches]],
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
