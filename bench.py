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
catalogue, train-masked, top-20, metrics) is timed in the same run, under both
candidate filters of its listing pass, from the same model state:
`eval_users_per_s` = the evaluator as the CLIs run it (fp16 candidate filter +
fp32 re-scoring: the fp32 ranking bit for bit; MACR_EVAL_FILTER=bf16|f32 for the
others), `roofline_eval` = the same evaluations with fp32 products throughout,
priced against the fp32 MFMA peak (`roofline_eval.eval_users_per_s` is that
configuration's rate), `roofline_eval_f16` (`_bf16`) = what the fp16 (bf16)
matrix cores execute in the default one.

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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 / _f16, dense (MI355X_MICROARCH.md)
# reduced-precision candidate filters of the evaluator (same ranking as "f32", bit for bit); the Evaluator's default
LP_FILTERS = ("bf16", "f16")
DEFAULT_EVAL_FILTER = "f16"


def lp_filter_name():
    return os.environ.get("MACR_EVAL_FILTER", DEFAULT_EVAL_FILTER).strip().lower()
N_BATCHES = 256                # pre-generated sampler batches every leg of the default workloads cycles through


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="gowalla", choices=["addressa", "gowalla", "ml10m", "yelp2018", "config4"],
                    help="config4 = BASELINE configs[4]: 10 M users x 1 M items, d=128, ONE model row-sharded over the ranks "
                         "(training) and item-sharded (evaluation); every other workload trains replicas")
    ap.add_argument("--c4-users", type=int, default=10_000_000)
    ap.add_argument("--c4-items", type=int, default=1_000_000)
    ap.add_argument("--c4-eval-users", type=int, default=100_000)
    ap.add_argument("--c4-lazy", type=int, default=0,
                    help="config4: period K of the lazy dense Adam pass (1 = the dense pass every step; 0 = by shard size)")
    ap.add_argument("--no-config4", action="store_true",
                    help="N >= 2 only: skip the configs[4] leg (ONE row-sharded model over the ranks) that the default workload's "
                         "line carries as `config4`")
    ap.add_argument("--train", default="rubibceboth", choices=["normalbce", "rubibceboth"])
    ap.add_argument("--mf", action="store_true", help="--workload yelp2018 runs MACR-LightGCN (BASELINE configs[3]); --mf runs "
                                                      "MACR-MF at the same shapes instead (diagnostic)")
    ap.add_argument("--eval-reps", type=int, default=20, help="timed evaluations (more than one period of the seeding policy's back-off)")
    ap.add_argument("--eval-settle", type=int, default=6, help="untimed evaluations (with the training steps between them) before the timed ones")
    ap.add_argument("--eval-train-steps", type=int, default=-1,
                    help="untimed training steps between two timed evaluations (the evaluator seeds its thresholds with the "
                         "previous evaluation's ranking, so the tables must move as they do in a training run).  Default -1 = "
                         "what the CLIs do between two evaluations: --log_interval 10 epochs of n_train // batch + 1 steps")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pos", default="zipf", choices=["zipf", "uniform"], help="positive-item distribution (diagnostic)")
    ap.add_argument("--no-eval", action="store_true", help="skip the evaluator leg (diagnostic)")
    ap.add_argument("--regions", type=int, default=0, help="number of timed K-step regions (0 = as many as fit in ~60 ms); the median is reported")
    ap.add_argument("--no-e2e", action="store_true", help="skip the sampler-inclusive end-to-end leg (diagnostic)")
    ap.add_argument("--no-defer", action="store_true", help="complete every step's Adam pass inside the step instead of under the next step's (B,B) kernel (diagnostic)")
    ap.add_argument("--through-cli-test", action="store_true",
                    help="also time macr_mf/train.py::test() itself (tools/cli_test_cost.py in a subprocess: a synthetic dataset of the "
                         "workload's shapes through the MF loader, the CLI's own call) -> `cli_test` in the line")
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


def bench_config4(args, rank, world, dev, emit=True):
    """BASELINE configs[4]: synthetic 10 M users x 1 M items, d = 128, B = 8192, `rubibceboth`, ONE model whose rows are
    sharded over the ranks, row r on rank r % W (macr_amd/sharded_train.py: three batch-sized collectives per step, dense Adam on the
    rank's shard) and whose evaluation is item-sharded (one all-gather of per-shard top-K).  `value` = interactions/s of
    that one model = B * steps / time: the work is fixed, more ranks divide it -- strong scaling.  At --gpus 1 the whole
    22.5 GB of tables, Adam slots and gradient scratch sit on the one GPU (the > 2^32-byte case of every kernel)."""
    import math
    from macr_amd import ops, sharding, synth, sharded_train
    from macr_amd.evaluator import Evaluator
    n_users, n_items, d, B = args.c4_users, args.c4_items, 128, 8192
    lr, regs, alpha, beta, c = 1e-3, 1e-5, 1e-3, 1e-3, 40.0
    own_u = sharded_train.Owned(n_users, rank, world)         # interleaved: row r on rank r % world
    own_i = sharded_train.Owned(n_items, rank, world)
    gen = torch.Generator(device=dev).manual_seed(4000 + rank)

    def xavier_rows(rows_local, rows_global):
        lim = math.sqrt(6.0 / (rows_global + d))
        return ((torch.rand((rows_local, d), generator=gen, device=dev, dtype=torch.float32) * 2 - 1) * lim).contiguous()
    gen_all = torch.Generator(device=dev).manual_seed(12345)          # identical on every rank: branch vectors and batches
    w = synth.xavier_table(d, 1, gen_all, dev).reshape(-1)
    wu = synth.xavier_table(d, 1, gen_all, dev).reshape(-1)
    hyper = ops.make_hyper(lr, regs, alpha, beta, B)
    model = sharded_train.RowShardedMF(None, None, w, wu, sharded_train.HipBackend(ops.LOSS_RUBIBCEBOTH, d, hyper, dev),
                                       rank=rank, world=world,
                                       shards=(xavier_rows(own_u.n, n_users), xavier_rows(own_i.n, n_items), n_users, n_items),
                                       lazy_period=args.c4_lazy if args.c4_lazy > 0 else None)
    # (independent of --steps: same batches, same model, whatever the call.  256 since round 5: regions long enough for the lazy
    # optimizer pass's steady lag -- hundreds of steps -- revisited a pool of 32 batches twenty times, and the model memorised
    # them to the point of users whose branch factor underflows to 0, i.e. whose scores all tie: the evaluator's exact kernel)
    n_batches = 256
    batches = synth.train_batches(n_batches, n_users, n_items, B, gen_all, dev)
    sharding.broadcast_params([batches])      # ONE batch stream for the one model (the draws are seeded; this makes it a fact)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # the split step's routing table per batch, as a caller that knows its batches on the host has it (the CLI's sampler does):
    # W*W integers per batch, computed once here, so that no step synchronises with the host
    counts_pool = [None] * n_batches
    if world > 1 and model.split:
        counts_pool = [model.route(batches[k, 0], batches[k, 1], batches[k, 2])[0].tolist() for k in range(n_batches)]

    def run_steps(n, first):
        out = None
        for s in range(n):
            k = (first + s) % n_batches
            out = model.step(batches[k, 0], batches[k, 1], batches[k, 2], counts=counts_pool[k])
        return out
    run_steps(args.warmup, 0)

    def timed_region():
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, args.warmup)
        model.flush()               # lazy dense Adam: the region ends with every row at its last step (the tables a reader would see)
        torch.cuda.synchronize(); barrier()
        return sharding.max_over_ranks(time.perf_counter() - t0, dev)
    regions = [timed_region()]
    n_rep = args.regions if args.regions > 0 else int(min(50, max(1, round(0.25 / max(regions[0], 1e-6)))))
    n_rep = int(sharding.max_over_ranks(float(n_rep), dev))
    regions += [timed_region() for _ in range(n_rep - 1)]
    elapsed = float(np.median(regions))
    ms_per_step = 1e3 * elapsed / args.steps
    value = B * args.steps / elapsed
    last = run_steps(1, 0).cpu().numpy()
    if not np.isfinite(last).all():
        raise SystemExit("ERROR: loss is nan.")
    # per-kernel events (rank-local launches) and per-collective events
    n_prof = 8
    run_steps(2 * model.lazy_period if model.lazy_period > 1 else 0, 1)      # lazy pass: every sweep at its full lag again (the regions end in a flush)
    model.collective_ms = {}
    ops.timing_begin()
    run_steps(n_prof, 0)
    marks = ops.timing_end(max_n=n_prof * 40)
    torch.cuda.synchronize()
    coll = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in model.collective_ms.items()}
    model.collective_ms = None
    ku = {}
    for name, ms in marks:
        e = ku.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += ms
    kern = {n_: {"launches_per_step": c_ / n_prof, "event_us": 1e3 * t / c_} for n_, (c_, t) in ku.items()}
    rows_local = own_u.n + own_i.n
    adam_bytes = 24.0 * d * rows_local
    adam_name = "adam_lazy" if "adam_lazy" in kern else "adam_indexed" if "adam_indexed" in kern else "adam_dense"    # (indexed: the pass sums the staged gradient rows itself)
    adam_us = kern.get(adam_name, {}).get("event_us")
    roofline = None
    if adam_us:
        gbps = adam_bytes / (adam_us * 1e-6) / 1e9
        roofline = {"kernel": adam_name, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBS, "traffic": None, "avg_us": adam_us, "algorithmic_bytes": adam_bytes,
                    "note": "TF-style dense Adam over this rank's %d rows (24*d bytes per row and step); event-timed, "
                            "includes ~3 us of event overhead" % rows_local}
        try:            # PMC HBM bytes per launch of the same command (tools/profile.sh c4 -> profiles/pmc_latest.json), when it was profiled
            roofline["traffic"] = json.load(open(os.path.join(REPO, "profiles", "pmc_latest.json"))).get("config4", {}).get(adam_name)
            roofline["traffic_source"] = REPLAYED % "pmc_latest.json[config4]" if roofline["traffic"] else None
        except Exception:
            pass
        if adam_name == "adam_lazy":
            # The lazy pass does K steps of a row's arithmetic per trip to HBM (SURVEY section 7 sanctions it): its launch does NOT move
            # the 24*d bytes per row and step, so an HBM fraction of it would exceed 1 and mean nothing.  Its bound is VALU issue:
            # `frac` = issue-time limit / measured time (<= 1); the bytes it really moves and the algorithmic-bytes rate are
            # reported beside it under names of their own.
            K = model.lazy_period
            moved = adam_bytes / K + 3.0 * B / world * (24 * d + 4)
            elems = 1.0 * d * rows_local
            valu_us = elems * 13 / 64 * 4.5 / (1024 * 2.4e9) * 1e6
            lim_us, src, detail = valu_us, "model: 13 VALU instructions per element and step, 4.5 cycles each per wave64, 1024 SIMDs at 2.4 GHz", None
            try:        # instruction counts of the same command under rocprofv3 --pmc (tools/profile.sh c4), issue cycles per class measured on this chip
                cnt = json.load(open(os.path.join(REPO, "profiles", "pmc_sq_latest.json"))).get("config4", {}).get("adam_lazy")
                rates = json.load(open(os.path.join(REPO, "profiles", "valu_rates.json"))).get("classes", {}).get("waves_per_simd_2", {})
                cyc_full = rates.get("v_fma_f32", {}).get("cycles_at_2p4GHz", 4.0)
                cyc_tr = rates.get("v_exp_f32", {}).get("cycles_at_2p4GHz", 16.0)
                insts, tr = cnt["SQ_INSTS_VALU"], cnt.get("SQ_INSTS_VALU_TRANS_F32", 2.0 / 13.0 * cnt["SQ_INSTS_VALU"])
                lim_us = ((insts - tr) * cyc_full + tr * cyc_tr) / 1024.0 / 2.4e3
                src = REPLAYED % "pmc_sq_latest.json[config4][adam_lazy] + valu_rates.json"
                detail = {"valu_wave_insts_per_launch": insts, "transcendental_share": tr / insts,
                          "issue_cycles_per_wave_instr": {"full_rate": cyc_full, "transcendental": cyc_tr},
                          "avg_us_under_pmc": cnt.get("avg_ns_under_pmc", 0) / 1e3}
            except Exception:
                pass
            roofline = {"kernel": adam_name, "bound": "valu-issue", "achieved": lim_us, "peak": adam_us, "unit": "us (issue-time limit / measured)",
                        "frac": min(1.0, lim_us / adam_us), "traffic": roofline["traffic"],
                        "traffic_source": REPLAYED % "pmc_latest.json[config4]" if roofline["traffic"] else None,
                        "avg_us": adam_us, "period": K, "issue_limit_us": lim_us, "issue_limit_source": src, "valu_issue": detail,
                        "valu_floor_us_model": valu_us,
                        "hbm_moved": {"bytes_per_launch_model": moved, "GBps": moved / (adam_us * 1e-6) / 1e9,
                                      "frac_of_hbm_peak": moved / (adam_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                      "note": "a K-th of the shard's theta/m/v + the batch's rows: what one launch really moves"},
                        "algorithmic_bytes_rate": {"bytes_per_launch": adam_bytes, "GBps": gbps,
                                                   "note": "SURVEY 8(d)'s 24*d bytes per row and step divided by the launch time: every row still "
                                                           "receives every step's update, but K steps share one trip to HBM -- NOT a bandwidth and "
                                                           "not comparable with the HBM peak (it exceeds it by design)"},
                        "note": "lazy dense Adam (period %d): bound by the VALU time of the K-step arithmetic per row; frac = issue_limit_us / "
                                "avg_us" % K}
    step_bytes = B * (24 * d + 12) + 24.0 * d * (n_users + n_items)
    if model.lazy_period > 1:
        # a lazy step moves a K-th of the tables, not all of them: the whole-step HBM fraction is taken on the bytes a step moves
        # (<= 1); SURVEY 8(d)'s algorithmic bytes per second are reported under a name of their own
        moved_step = B * (24 * d + 12) + 24.0 * d * (n_users + n_items) / model.lazy_period + 3.0 * B * (24 * d + 4)
        g = moved_step / (ms_per_step * 1e-3) / 1e9 / max(world, 1)
        roofline_step = {"bound": "hbm", "bytes_moved_model": moved_step, "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                         "frac": g / HBM_PEAK_GBS, "algorithmic_bytes": step_bytes,
                         "algorithmic_bytes_rate_GBps": step_bytes / (ms_per_step * 1e-3) / 1e9 / max(world, 1),
                         "note": "lazy dense Adam (period %d): `achieved` / `frac` = the bytes a step really moves (pair part + a K-th of "
                                 "theta/m/v + the batch's rows) over the step time; algorithmic_bytes_rate_GBps = SURVEY 8(d)'s bytes per "
                                 "step over the same time, which exceeds the HBM peak by design and is not a bandwidth" % model.lazy_period}
    else:
        g = step_bytes / (ms_per_step * 1e-3) / 1e9 / max(world, 1)
        roofline_step = {"bound": "hbm", "algorithmic_bytes": step_bytes, "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                         "frac": g / HBM_PEAK_GBS}
    # ------------------------------------------------------------- evaluation: item shards + one all-gather
    eval_out = {}
    if not args.no_eval:
        rs = np.random.RandomState(777)
        U = min(args.c4_eval_users, n_users)
        users = np.sort(rs.choice(n_users, U, replace=False)).astype(np.int64)
        mask_lists = synth.interaction_lists(U, n_items, 30.0, seed=778)
        gt_lists = [sorted(set(rs.randint(0, n_items, 50).tolist()) - set(m)) for m in mask_lists]
        ev = Evaluator(mask_lists, gt_lists, n_items, dev)
        ev.set_local_items(own_i)
        ud = torch.from_numpy(users).to(dev)
        Pq = torch.zeros((U, d), dtype=torch.float32, device=dev)       # the query users' rows: every rank adds the ones it owns

        def query_rows():
            own, loc = own_u.local_of(ud)
            Pq.zero_()
            Pq[own] = model.rows("P", loc[own])      # (the query users' rows as of now: no pass over the 10 M-row user table)
            if world > 1:
                torch.distributed.all_reduce(Pq)

        def run_eval():
            query_rows()
            return ev.test_mf(ops.SCORE_RUBI_BOTH, Pq, None, model.Q, [20], model.w, model.wu, c)
        def timed(filt):
            ev.filter = filt
            ret = run_eval(); ret = run_eval()
            times = []
            for r_ in range(max(2, min(args.eval_reps, 5))):
                run_steps(2, 2 * r_)
                barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                ret = run_eval()
                torch.cuda.synchronize(); barrier()
                times.append(sharding.max_over_ranks(time.perf_counter() - t0, dev))
            return float(np.median(times)), ret
        # fp32 products throughout (priced against the fp32 MFMA peak), then the evaluator's default: the bf16 candidate
        # filter with fp32 re-scoring -- the same ranking, bit for bit
        t_f32, _ = timed("f32")
        t_ev, ret = timed(lp_filter_name())
        flops_rank = 2.0 * U * own_i.n * d
        eval_out = {"eval_users_per_s": U / t_ev, "eval_ms_per_pass": 1e3 * t_ev, "eval_users": U, "eval_filter": ev.filter,
                    "eval_info": ev.last_eval_info(), "eval_fast_stats": dict(getattr(ev, "fast_stats", {})),
                    "eval_metrics": {k: float(v[0]) for k, v in ret.items()},
                    "roofline_eval": {"filter": "f32", "bound": "mfma", "flops_per_rank": flops_rank, "eval_users_per_s": U / t_f32,
                                      "achieved": flops_rank / t_f32 / 1e12,
                                      "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops_rank / t_f32 / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                      "note": "whole evaluation incl. the query-row exchange, the all-gather and the metrics, per rank"}}
    # ------------------------------------------------------------- CPU baseline (N = 1): the oracle on a 1/64 row sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        nu, ni = n_users // 64, n_items // 64
        rs = np.random.RandomState(5)
        Pc = (rs.standard_normal((nu, d)) * 0.02).astype(np.float32); Qc = (rs.standard_normal((ni, d)) * 0.02).astype(np.float32)
        wc, wuc = w.cpu().numpy().copy(), wu.cpu().numpy().copy()
        st = oracle.AdamState([Pc.shape, Qc.shape, (d,), (d,)])
        hb = batches[:4].cpu().numpy()
        omp_set_threads(min(32, os.cpu_count() or 1))
        with oracle.fast():                       # the speed build of the C port (oracle/Makefile `fast`)
            t0, n_cpu = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.cpu_seconds and n_cpu < 64:
                k = n_cpu % 4
                oracle.mf_train_step(ops.LOSS_RUBIBCEBOTH, (hb[k, 0] % nu).astype(np.int32), (hb[k, 1] % ni).astype(np.int32),
                                     (hb[k, 2] % ni).astype(np.int32), Pc, Qc, wc, wuc, st, lr, regs, alpha, beta, B)
                n_cpu += 1
        cpu = {"value": n_cpu * B / (time.perf_counter() - t0), "unit": "interactions/s", "cores": min(32, os.cpu_count() or 1),
               "host_cpus": os.cpu_count() or 1, "cpu_model": cpu_model(), "kind": "port",
               "sample": "%d steps of the C port (fast build; B=%d, d=%d) on tables of 1/64 of the rows (%d + %d): the dense Adam "
                         "pass of the full tables would be 64x that part of a step" % (n_cpu, B, d, nu, ni)}
    if rank == 0:
        out = {"metric": "train interactions/sec + eval users/sec (full-catalog top-K@20)",
               "value": value, "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "eval_users_per_s": eval_out.get("eval_users_per_s"),
               "config": {"workload": "configs[4]: synthetic %d users x %d items, MACR-MF rubibceboth d=%d batch=%d c=%g; ONE model, "
                                      "rows sharded over %d rank(s), row r on rank r %% W (training), item-sharded evaluation of %d query users"
                                      % (n_users, n_items, d, B, c, world, args.c4_eval_users),
                          "parallelism": "row-sharded x%d: %s; evaluation: all-reduce of the query rows + one all-gather of per-shard "
                                         "top-K" % (world, ("split step -- all-to-all of the 3B/W rows a rank's slice needs, all-reduces of the "
                                                            "forward scalars, the (B,B) partials and the branch-vector partials, all-to-all of the "
                                                            "gradient rows back to their owners") if (world > 1 and model.split) else
                                                    ("all-reduce of the batch's 3B rows (%.1f MB) + all-reduce of the (B,B) partials + broadcast "
                                                     "of the branch-vector partials per step" % (3 * B * d * 4 / 1e6))),
                          "global_batch": B},
               "rows_per_rank": rows_local, "bytes_per_rank": 4.0 * d * rows_local * 4 + 4.0 * rows_local, "lazy_adam_period": model.lazy_period,
               "lazy_adam_note": None if model.lazy_period <= 1 else
               "every timed region ends with a flush (all rows at the last step); a region shorter than ~4 periods (%d steps) never reaches the "
               "pass's steady lag and reads faster than training does -- quote --steps >= %d" % (4 * model.lazy_period, 4 * model.lazy_period),
               "wire_bytes_per_step": None if model.wire_rows is None else {
                   "rows_crossing_ranks_each_way": model.wire_rows, "row_bytes": model.wire_rows * d * 4,
                   "replicated_step_all_reduce_buffer_bytes": 3 * B * d * 4,
                   "note": "split step: rows to the slices + gradient rows back (two all-to-alls, this rank, last step) against the "
                           "(3,B,d) buffer the replicated step all-reduces (a ring moves ~2x that per rank)"},
               "collectives_ms": coll, "kernels": kern, "roofline": roofline,
               "roofline_step": roofline_step,
               "timed_regions": {"n": len(regions), "reported": "median", "min_ms_per_step": 1e3 * min(regions) / args.steps,
                                 "max_ms_per_step": 1e3 * max(regions) / args.steps},
               "cpu_baseline": cpu, "last_losses": [float(x) for x in last]}
        out.update(eval_out)
        if not emit:
            return out
        print(json.dumps(out))
    if not emit:
        return None
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_lgcn(args, rank, world, dev):
    """BASELINE configs[3]: Yelp2018-shape MACR-LightGCN, 2 layers [64,64], `bceboth`, B = 4096, c = 40
    (macr_lightgcn/LightGCN.py:288-309 propagation, :495-532 loss, :201 Adam; utility/batch_test.py:26-162 evaluation).
    A step = dense SpMM layer + batch-row-sparse layer forward, the pair kernels on propagated rows, the same two layers
    backward with the ego regulariser and dense Adam in the last one's epilogue.  `value` = training interactions/s
    (N ranks: N replicas, weak); the evaluation (propagate once + full-catalogue ranking + fold-out metrics) is item-sharded
    over the ranks.  `roofline` = the dominant launch, the dense SpMM layer, against HBM with SURVEY 8(d)'s compulsory bytes
    nnz*8 + (N+1)*4 + 2*N*d*4."""
    from macr_amd import ops, sharding, synth
    from macr_amd.evaluator import Evaluator
    cfg = synth.WORKLOADS[args.workload]
    n_u, n_i, d, B, L = cfg["n_users"], cfg["n_items"], cfg["d"], cfg["batch"], 2
    kind = ops.LOSS_RUBIBCEBOTH if args.train == "rubibceboth" else ops.LOSS_NORMALBCE
    lists, A = synth.lgcn_graph(cfg, seed=9)                       # same graph on every rank
    N, nnz = A.shape[0], A.nnz
    gen = torch.Generator(device=dev).manual_seed(12345 + rank)
    T = synth.xavier_table(N, d, gen, dev)
    w = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    adj = ops.CSR.from_scipy(A, dev)
    hyper = ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
    state = ops.LGCNState(T, n_u, n_i, w, wu, adj, L, hyper, B)
    n_batches = 64
    batches = synth.train_batches(n_batches, n_u, n_i, B, gen, dev)
    loss_log = torch.zeros((n_batches, 3), dtype=torch.float32, device=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def run_steps(n, first):
        for s in range(n):
            k = (first + s) % n_batches
            state.step(kind, batches[k, 0], batches[k, 1], batches[k, 2], loss_log[k])
    run_steps(args.warmup, 0)

    def timed_region():
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, args.warmup)
        torch.cuda.synchronize(); barrier()
        return sharding.max_over_ranks(time.perf_counter() - t0, dev)
    regions = [timed_region()]
    n_rep = args.regions if args.regions > 0 else int(min(100, max(1, round(0.1 / max(regions[0], 1e-6)))))
    n_rep = int(sharding.max_over_ranks(float(n_rep), dev))
    regions += [timed_region() for _ in range(n_rep - 1)]
    elapsed = float(np.median(regions))
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    losses = loss_log.cpu().numpy()
    if not np.isfinite(losses[: min(n_batches, args.steps)]).all():
        raise SystemExit("ERROR: loss is nan.")
    # per-kernel HIP events on the launch stream, calibrated against the un-instrumented region as in main()
    n_prof = 10
    ops.timing_begin()
    run_steps(n_prof, 0)
    marks = ops.timing_end(max_n=n_prof * 16 + 16)
    agg = {}
    for name, ms in marks:
        a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
    kern = {n_: {"launches_per_step": c_ / n_prof, "event_us": 1e3 * t / c_} for n_, (c_, t) in agg.items()}
    ev_sum = sum(v["event_us"] * v["launches_per_step"] for v in kern.values())
    launches = sum(v["launches_per_step"] for v in kern.values())
    overhead = max(0.0, (ev_sum - 1e3 * ms_per_step) / launches)
    for v in kern.values():
        v["avg_us"] = max(v["event_us"] - overhead, 0.1)
    pmc = {}
    pmc_path = os.path.join(REPO, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path)).get(args.workload, {})
        except Exception:
            pmc = {}
    layer_bytes = nnz * 8 + (N + 1) * 4 + 2 * N * d * 4                 # SURVEY 8(d) K5, one dense layer
    adam_bytes = 24 * d * N
    alg = {"spmm_csr": layer_bytes, "spmm_stream": layer_bytes,
           # the last backward layer: a dense layer whose epilogue is the ego regulariser + dense Adam on T (read T, m, v
           # and write them back instead of writing the layer's output rows)
           "spmm_csr+adam": nnz * 8 + (N + 1) * 4 + N * d * 4 + adam_bytes, "spmm_stream+adam": nnz * 8 + (N + 1) * 4 + N * d * 4 + adam_bytes}
    for n_, v in kern.items():
        if n_ in alg:
            v["algorithmic_bytes"] = alg[n_]
            v["GBps"] = alg[n_] / (v["avg_us"] * 1e-6) / 1e9
        if n_ == "bxb":
            v["gevals_per_s"] = 2.0 * B * B / (v["avg_us"] * 1e-6) / 1e9
    dense = "spmm_stream" if "spmm_stream" in kern else "spmm_csr"
    dk = kern[dense]
    roofline = {"kernel": dense, "bound": "hbm", "achieved": dk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dk["GBps"] / HBM_PEAK_GBS, "traffic": pmc.get(dense),
                "traffic_source": (REPLAYED % ("pmc_latest.json[%s][%s]" % (args.workload, dense))) if pmc.get(dense) else None,
                "avg_us": dk["avg_us"], "avg_us_source": "HIP events on the launch stream, this run",
                "working_set_bytes": 4 * N * d * 4 + nnz * 8, "resident": "infinity-cache: tables, layer buffers and the adjacency (%.0f MB) stay in the "
                "256 MiB Infinity Cache; the gather's cost is L2 misses served from it, not HBM" % ((4 * N * d * 4 + nnz * 8) / 1e6),
                "us_per_step": dk["avg_us"] * dk["launches_per_step"], "algorithmic_bytes": layer_bytes,
                "note": "one dense propagation layer E' = A E (LightGCN.py:297-305): nnz*8 + (N+1)*4 + 2*N*d*4 compulsory bytes, "
                        "N = %d, nnz = %d; `traffic` = PMC HBM bytes per launch (profiles/pmc_latest.json): the gather misses "
                        "the L2 of the XCD it runs on, which is what it pays above the compulsory bytes" % (N, nnz)}
    # whole step: SURVEY 8(d) "a LightGCN step = 2 layers fwd + 2 bwd" + the pair part + dense Adam over T
    step_bytes = 2 * L * layer_bytes + B * (24 * d + 12) + adam_bytes
    roofline_step = {"bound": "hbm", "algorithmic_bytes": step_bytes, "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "2*L dense layers + pair part + dense Adam, as SURVEY 8(d) counts a step; the step here runs the last "
                             "forward and the first backward layer for the batch's rows only, so it moves fewer bytes than this"}
    # ------------------------------------------------------------- evaluation: propagate once, rank, fold-out metrics
    rs = np.random.RandomState(777)
    U = cfg["n_test_users"]
    users = np.sort(rs.choice(n_u, U, replace=False)).astype(np.int32)
    mask_lists = [lists[u_] for u_ in users]                      # the query users' train items (batch_test.py:124-129)
    gt_lists = []
    for row in mask_lists:
        seen, t_ = set(row), []
        while len(t_) < cfg["test_per_user"]:
            x = int(rs.randint(0, n_i))
            if x not in seen:
                t_.append(x); seen.add(x)
        gt_lists.append(sorted(t_))
    Ks = [20]
    eval_out = {}
    if not args.no_eval:
        steps_between = args.eval_train_steps if args.eval_train_steps >= 0 else 10 * (cfg["n_train"] // B + 1)
        steps_between = min(steps_between, 400)                   # (a 0.2 ms step: 400 steps = what the model moves in ~1 epoch)
        uid = torch.from_numpy(users).to(dev)

        def one_model():
            if world > 1:
                for t in (state.T, state.w, state.wu):
                    torch.distributed.broadcast(t, 0)
                state._E = None

        def suite(filt):
            ev = Evaluator(mask_lists, gt_lists, n_i, dev)
            ev.filter = filt

            def run_eval():
                E = state.propagated()                            # E = mean(E0, A E0, A^2 E0): once per evaluation
                return ev.test_lgcn(ops.SCORE_RUBI_BOTH, E[:n_u], uid, E[n_u:], Ks, state.w, state.wu, cfg["c"])
            one_model(); run_eval()
            for r_ in range(min(args.eval_settle, 3)):
                run_steps(steps_between, steps_between * r_); one_model(); run_eval()
            times, modes = [], []
            for r_ in range(args.eval_reps):
                run_steps(steps_between, steps_between * r_); one_model()
                barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                ret = run_eval()
                torch.cuda.synchronize(); barrier()
                times.append(sharding.max_over_ranks(time.perf_counter() - t0, dev))
                modes.append(ev.last_eval_info())
            # kernels of one evaluation (direct launches; the propagation included)
            ev.use_graph = False
            state._E = None
            ops.timing_begin()
            run_eval()
            ek = {}
            for name, ms in ops.timing_end():
                ek[name] = ek.get(name, 0.0) + 1e3 * max(ms - 1e-3 * overhead, 0.0)
            ev.use_graph = True
            ts = sorted(times)
            return {"users_per_s": U * len(times) / sum(times), "ms_per_eval": 1e3 * sum(times) / len(times),
                    "ms_per_eval_median": 1e3 * ts[len(ts) // 2], "ms_per_eval_min": 1e3 * ts[0], "ms_per_eval_max": 1e3 * ts[-1],
                    "evaluations": len(times), "seeded": sum(1 for m in modes if m["seeded"]),
                    "repaired": sum(1 for m in modes if m["query_blocks_relisted"] > 0 or m["exact_fallback"]),
                    "kernels_us": ek, "metrics": {k: float(v[0]) for k, v in ret.items()}}
        s32 = suite("f32")
        sdef = suite(lp_filter_name())
        lo, hi = sharding.item_shard_range(n_i, rank, world)
        flops = 2.0 * U * (hi - lo) * d
        k32 = s32["kernels_us"]
        rank_us = sum(v for k_, v in k32.items() if k_.startswith(("score_", "tau", "select", "bf16_prep", "repair_plan")))
        eval_out = {"eval_users_per_s": sdef["users_per_s"], "eval_ms_per_pass": sdef["ms_per_eval"], "eval_users": U,
                    "eval_metrics": sdef["metrics"],
                    "eval": {"default_filter": lp_filter_name(), "f32": s32, lp_filter_name(): sdef, "train_steps_between_evaluations": steps_between,
                             "note": "one evaluation = propagation (L dense layers + mean) + branch sigmoids + ranking + fold-out "
                                     "metrics (batch_test.py:26-162), graph replays except the propagation"},
                    "roofline_eval": {"filter": "f32", "bound": "mfma", "flops": flops, "avg_us": rank_us,
                                      "achieved": flops / (rank_us * 1e-6) / 1e12 if rank_us else None, "peak": MFMA_F32_PEAK_TFLOPS,
                                      "unit": "TFLOP/s", "frac": flops / (rank_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS if rank_us else None,
                                      "eval_users_per_s": s32["users_per_s"],
                                      "note": "U*N*d multiply-adds counted once over the ranking kernels of one sampled evaluation"}}
    # ------------------------------------------------------------- CPU baseline: the oracle's LightGCN step and evaluation
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        from oracle import torch_port
        T0, w0, wu0 = state.T.cpu().numpy().copy(), state.w.cpu().numpy().copy(), state.wu.cpu().numpy().copy()
        hb = batches[:4].cpu().numpy()

        def c_steps(seconds, cap):
            Tc, wc, wuc = T0.copy(), w0.copy(), wu0.copy()
            st = oracle.AdamState([Tc.shape, (d,), (d,)])
            step = lambda k: oracle.lgcn_train_step(kind, n_u, n_i, L, A.indptr, A.indices, A.data, hb[k % 4, 0], hb[k % 4, 1],
                                                    hb[k % 4, 2], Tc, wc, wuc, st, cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
            step(0)
            t0, n = time.perf_counter(), 0
            while n < 2 or (time.perf_counter() - t0 < seconds and n < cap):
                step(n + 1); n += 1
            return n * B / (time.perf_counter() - t0), n
        with oracle.fast():
            Tw, ww, wuw = T0.copy(), w0.copy(), wu0.copy()
            stw = oracle.AdamState([Tw.shape, (d,), (d,)])
            omp_n, omp_seen = best_omp_threads(lambda: oracle.lgcn_train_step(
                kind, n_u, n_i, L, A.indptr, A.indices, A.data, hb[0, 0], hb[0, 1], hb[0, 2], Tw, ww, wuw, stw, cfg["lr"],
                cfg["regs"], cfg["alpha"], cfg["beta"], B))
            fast_rate, n_fast = c_steps(0.2 * args.cpu_seconds, 1024)
        strict_rate, n_strict = c_steps(0.15 * args.cpu_seconds, 256)
        torch_rate = n_torch = None
        torch_threads0 = torch.get_num_threads()
        try:
            torch.set_num_threads(min(torch_threads0, max(omp_n, 32)))
            pk = torch_port.LOSS_RUBIBCEBOTH if kind == ops.LOSS_RUBIBCEBOTH else torch_port.LOSS_NORMALBCE
            port = torch_port.LGCNPort(T0, n_u, n_i, w0, wu0, torch_port.csr_to_torch(A.indptr, A.indices, A.data, N), L,
                                       cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
            port.train_step(pk, hb[0, 0], hb[0, 1], hb[0, 2])
            t0, n_torch = time.perf_counter(), 0
            while n_torch < 1 or (time.perf_counter() - t0 < 0.2 * args.cpu_seconds and n_torch < 64):
                port.train_step(pk, hb[(n_torch + 1) % 4, 0], hb[(n_torch + 1) % 4, 1], hb[(n_torch + 1) % 4, 2]); n_torch += 1
            torch_rate = n_torch * B / (time.perf_counter() - t0)
        except Exception as e:
            n_torch = repr(e)
        n_ev = min(1024, U)
        mptr, midx = oracle.csr_from_lists(mask_lists[:n_ev]); gptr, gidx = oracle.csr_from_lists(gt_lists[:n_ev])
        with oracle.fast():
            t0 = time.perf_counter()
            E = oracle.lgcn_propagate(A.indptr, A.indices, A.data, T0, L)
            t_prop = time.perf_counter() - t0
            t0 = time.perf_counter()
            Eu, Ei = np.ascontiguousarray(E[users[:n_ev]]), np.ascontiguousarray(E[n_u:])
            sig_i = oracle.branch_sigmoid(Ei, w0); sig_u = oracle.branch_sigmoid(Eu, wu0)
            _, oi, _ = oracle.score_topk(oracle.SCORE_RUBI_BOTH, Eu, Ei, 20, sig_u, sig_i, cfg["c"], (mptr, midx), fill_masked=True)
            oracle.metrics_foldout(oi, (gptr, gidx))
            t_rank = time.perf_counter() - t0
        ev_impl = {"c_port_fast": U / (t_prop + t_rank * U / n_ev)}
        if oracle.have_ref():
            # the reference's own C++ evaluator on the dense score rows batch_test.py:134 hands it (train items at -inf)
            S = oracle.score_matrix(oracle.SCORE_RUBI_BOTH, Eu, Ei, sig_u, sig_i, cfg["c"])
            for q in range(n_ev):
                S[q, mask_lists[q]] = -np.inf
            t0 = time.perf_counter()
            oracle.ref_eval_score_matrix_foldout(S, gt_lists[:n_ev], top_k=20, thread_num=os.cpu_count() or 4)
            ev_impl["reference_cpp_ranking_only"] = n_ev / (time.perf_counter() - t0)
        used_torch_threads = torch.get_num_threads()
        torch.set_num_threads(torch_threads0)
        cpu = {"value": fast_rate, "unit": "interactions/s", "cores": omp_n, "host_cpus": os.cpu_count() or 1, "cpu_model": cpu_model(), "kind": "port",
               "omp_threads_s_per_step": omp_seen,
               "what": "c_port_fast = oracle/macr_oracle.c (orc_lgcn_train_step: CSR SpMM layers forward and backward, pair loss "
                       "and gradients, dense Adam on T), OpenMP, compiled -O3 -march=x86-64-v3 -ffast-math; c_checker = the strict "
                       "build the parity tests use; torch_graph = oracle/torch_port.py, the reference's graph on torch-CPU "
                       "(sparse CSR @ dense, dense (B,B) tensors, autograd, TF-form Adam)",
               "implementations": {"c_port_fast": fast_rate, "c_checker": strict_rate, "torch_graph": torch_rate},
               "torch_threads": used_torch_threads,
               "sample": "%d / %d / %s LightGCN training steps (fast C / checker / torch graph; B=%d, N=%d, nnz=%d, L=%d); eval: one "
                         "propagation (%.3f s) + ranking and metrics of %d of the %d query users, extrapolated"
                         % (n_fast, n_strict, n_torch, B, N, nnz, L, t_prop, n_ev, U),
               "eval_users_per_s": ev_impl["c_port_fast"], "eval_implementations": ev_impl}
    if rank == 0:
        out = {"metric": "train interactions/sec + eval users/sec (full-catalog top-K@20)",
               "value": value, "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "nranks": world, "eval_users_per_s": eval_out.get("eval_users_per_s"),
               "config": {"workload": "%s-shape MACR-LightGCN %s, %d layers [64,64], d=%d batch=%d c=%g (n_users=%d, n_items=%d, "
                                      "N=%d nodes, nnz=%d = 2*n_train, `pre` adjacency); synthetic graph (lognormal user degrees, Zipf "
                                      "items), Xavier tables, batches as sampled"
                                      % (args.workload, "bceboth" if kind == ops.LOSS_RUBIBCEBOTH else "bce", L, d, B, cfg["c"], n_u, n_i, N, nnz),
                          "parallelism": "replicas x%d (train) / item-sharded x%d + RCCL all-gather (eval)" % (world, world),
                          "global_batch": B * world},
               "timed_regions": {"n": len(regions), "each": "exactly %d steps, barrier+synchronize on both sides" % args.steps,
                                 "reported": "median", "min_ms_per_step": 1e3 * min(regions) / args.steps,
                                 "max_ms_per_step": 1e3 * max(regions) / args.steps},
               "event_overhead_us_per_launch": overhead, "kernels": kern, "roofline": roofline, "roofline_step": roofline_step,
               "cpu_baseline": cpu, "last_losses": [float(x) for x in losses[(args.warmup + args.steps - 1) % n_batches]]}
        out.update(eval_out)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


REPLAYED = "profiles/%s -- rocprofv3 passes of the same command committed earlier (tools/profile.sh), NOT collected by this run"


def cpu_model():
    """model string of the host CPU the baseline legs run on"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def omp_set_threads(n):
    """thread count of the C port's OpenMP regions (libgomp, the runtime oracle/_build/*.so link): the default -- every
    hardware thread of a 2-socket SMT host -- is its worst setting (measured on the GPU box: 701 ms per Gowalla step at 256
    threads, 23 ms at 32)"""
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def best_omp_threads(step, candidates=(16, 32, 64, 128)):
    """time `step()` under each thread count (one warm call, two timed) and leave the fastest set; -> (threads, {n: s/step})"""
    ncpu = os.cpu_count() or 1
    seen, best = {}, None
    for n in [c for c in candidates if c <= ncpu] or [ncpu]:
        if not omp_set_threads(n):
            return ncpu, {}
        step()
        t0 = time.perf_counter(); step(); step()
        seen[n] = (time.perf_counter() - t0) / 2
        if best is None or seen[n] < seen[best]:
            best = n
    omp_set_threads(best)
    return best, seen


def cpu_baseline_mf(args, kind, cfg, P, Q, w, wu, hb, users, mask_lists, gt_lists, Ks):
    """The reference's CPU path, restated, timed on this box's host cores on a bounded sample of the same workload
    (--cpu-seconds).  Three implementations of the same step, all held to each other by tests/test_torch_port_cpu.py:
      c_port_fast    oracle/macr_oracle.c compiled for speed (-O3, AVX2, -ffast-math: vectorised expf/logf), OpenMP -- `value`
      c_checker      the same file as the parity tests use it (-O2, strict IEEE, libm scalar calls)
      torch_graph    oracle/torch_port.py: the reference's TF1 graph op for op on torch-CPU (dense (B,B) tensors, autograd,
                     TF-form dense Adam) -- the closest thing to the TF1 CPU path that runs here
    Evaluation: the C port's fused scoring + ranking, the torch port's per-batch score matrix + topk, and the reference's
    own C++ evaluator (oracle/_ref, built from the reference's headers) on dense score rows."""
    import oracle
    from oracle import torch_port
    B, d = cfg["batch"], cfg["d"]
    Pc, Qc = P.cpu().numpy().copy(), Q.cpu().numpy().copy()
    wc, wuc = w.cpu().numpy().copy(), wu.cpu().numpy().copy()
    budget = args.cpu_seconds

    def c_steps(seconds, cap):
        Pw, Qw, ww, wuw = Pc.copy(), Qc.copy(), wc.copy(), wuc.copy()
        st = oracle.AdamState([Pw.shape, Qw.shape, (d,), (d,)])
        step = lambda k: oracle.mf_train_step(kind, hb[k % 8, 0], hb[k % 8, 1], hb[k % 8, 2], Pw, Qw, ww, wuw, st, cfg["lr"],
                                              cfg["regs"], cfg["alpha"], cfg["beta"], B)
        step(0)
        t0, n = time.perf_counter(), 0
        while n < 2 or (time.perf_counter() - t0 < seconds and n < cap):
            step(n + 1); n += 1
        return n * B / (time.perf_counter() - t0), n
    with oracle.fast():
        Pw, Qw, ww, wuw = Pc.copy(), Qc.copy(), wc.copy(), wuc.copy()
        stw = oracle.AdamState([Pw.shape, Qw.shape, (d,), (d,)])
        omp_n, omp_seen = best_omp_threads(lambda: oracle.mf_train_step(kind, hb[0, 0], hb[0, 1], hb[0, 2], Pw, Qw, ww, wuw, stw,
                                                                         cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B))
        fast_rate, n_fast = c_steps(0.2 * budget, 2048)
    strict_rate, n_strict = c_steps(0.15 * budget, 256)
    torch_rate = n_torch = None
    import torch as _t
    torch_threads0 = _t.get_num_threads()
    try:
        _t.set_num_threads(min(torch_threads0, max(omp_n, 32)))
        port = torch_port.MFPort(Pc, Qc, wc, wuc, cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
        pk = torch_port.LOSS_RUBIBCEBOTH if kind == 1 else torch_port.LOSS_NORMALBCE
        port.train_step(pk, hb[0, 0], hb[0, 1], hb[0, 2])
        t0, n_torch = time.perf_counter(), 0
        while n_torch < 1 or (time.perf_counter() - t0 < 0.2 * budget and n_torch < 64):
            port.train_step(pk, hb[(n_torch + 1) % 8, 0], hb[(n_torch + 1) % 8, 1], hb[(n_torch + 1) % 8, 2]); n_torch += 1
        torch_rate = n_torch * B / (time.perf_counter() - t0)
    except Exception as e:                                   # (a baseline leg must not take the bench line with it)
        torch_rate, n_torch = None, repr(e)
    # evaluation
    ev = {}
    n_c = min(1024, len(users))
    mptr, midx = oracle.csr_from_lists(mask_lists[:n_c]); gptr, gidx = oracle.csr_from_lists(gt_lists[:n_c])
    with oracle.fast():
        t0 = time.perf_counter()
        sig_i = oracle.branch_sigmoid(Qc, wc); sig_u = oracle.branch_sigmoid(Pc[users[:n_c]], wuc)
        _, oi, oc = oracle.score_topk(oracle.SCORE_RUBI_BOTH, Pc[users[:n_c]], Qc, Ks[0], sig_u, sig_i, cfg["c"], (mptr, midx))
        oracle.metrics_mf(oi, oc, (gptr, gidx), Ks)
        ev["c_port_fast"] = n_c / (time.perf_counter() - t0)
    try:
        port = torch_port.MFPort(Pc, Qc, wc, wuc, cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
        n_t = min(4096, len(users))
        port.evaluate(users[:256], mask_lists[:256], cfg["c"], K=Ks[0])
        t0 = time.perf_counter()
        ids = port.evaluate(users[:n_t], mask_lists[:n_t], cfg["c"], K=Ks[0], batch=B)
        ev["torch_graph"] = n_t / (time.perf_counter() - t0)
        if oracle.have_ref():
            # the reference's own evaluator (tools.h + evaluate_foldout.h, compiled from the reference's headers) on the score rows
            # the reference would hand it (batch_test.py:134): dense (U,N) fp32 with the train items at -inf
            n_r = min(1024, len(users))
            S = port.score_matrix(users[:n_r], cfg["c"]).numpy()
            for q in range(n_r):
                S[q, mask_lists[q]] = -np.inf
            t0 = time.perf_counter()
            oracle.ref_eval_score_matrix_foldout(S, gt_lists[:n_r], top_k=Ks[0], thread_num=os.cpu_count() or 4)
            ev["reference_cpp_ranking_only"] = n_r / (time.perf_counter() - t0)
    except Exception as e:
        ev["torch_graph_error"] = repr(e)
    used_torch_threads = _t.get_num_threads()
    _t.set_num_threads(torch_threads0)
    return {"value": fast_rate, "unit": "interactions/s", "cores": omp_n, "host_cpus": os.cpu_count() or 1, "cpu_model": cpu_model(), "kind": "port",
            "what": "c_port_fast = oracle/macr_oracle.c, OpenMP, compiled -O3 -march=x86-64-v3 -ffast-math (vectorised "
                    "expf/logf); gradient tables persistent (no per-step calloc); `cores` = the OpenMP thread count that was "
                    "fastest on this host (omp_threads_s_per_step: what each candidate took)",
            "omp_threads_s_per_step": omp_seen,
            "implementations": {"c_port_fast": fast_rate, "c_checker": strict_rate, "torch_graph": torch_rate},
            "torch_threads": used_torch_threads,
            "sample": "%d / %d / %s training steps (fast C / checker / torch graph) of the same workload (B=%d) on full-size tables; "
                      "eval: %d (C port) and up to 4096 (torch graph) of the %d query users"
                      % (n_fast, n_strict, n_torch, B, n_c, len(users)),
            "eval_users_per_s": max(v for k_, v in ev.items() if isinstance(v, float)), "eval_implementations": ev}


def roofline_bxb(workload, B, kern_avg):
    """The (B,B) launch against ITS bound: VALU issue ("transcendental VALU -- not HBM, not MFMA").
    Instruction counts per launch come from the PMC passes of the same command (profiles/pmc_sq_latest.json: SQ_INSTS_VALU,
    SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, separate rocprofv3 --pmc runs), the share of transcendentals from the static mix of the
    kernel's two pair loops (profiles/bxb_isa_mix.json, tools/isa_mix.py on the shipped code object), the issue cost of each class
    from tools/valu_rate_bench.hip on this chip (profiles/valu_rates.json; defaults: 4 cycles a full-rate or packed wave64
    instruction, 16 a transcendental).  issue_limit_us = sum over classes of count x cycles / 1024 SIMDs / 2.4 GHz peak clock;
    frac = issue_limit_us / measured time: 1.0 = the launch takes exactly what issuing its own instructions costs."""
    def load(name):
        path = os.path.join(REPO, "profiles", name)
        try:
            return json.load(open(path))
        except Exception:
            return None
    sq = (load("pmc_sq_latest.json") or {}).get(workload, {})
    mix = load("bxb_isa_mix.json") or {}
    rates = load("valu_rates.json") or {}
    out = {"bound": "valu-issue", "pairs": B * B, "fused_bce_evaluations": 2 * B * B, "simds": 1024, "peak_clock_ghz": 2.4,
           "sources": {"avg_us / gevals_per_s": "HIP events, this run",
                       "pmc, valu_wave_insts_per_launch, transcendental_share_measured": REPLAYED % ("pmc_sq_latest.json[%s]" % workload),
                       "static_mix": "profiles/bxb_isa_mix.json -- tools/isa_mix.py on the code object, committed; not this run",
                       "issue_cycles_per_wave_instr": "profiles/valu_rates.json -- tools/valu_rate_bench.hip on an MI355X, committed; not this run"}}
    # static mix: the kernel's two pair loops (6- and 4-transcendental forms), R = 4 rows per lane
    loops = []
    for name, k in mix.items():
        if "k_bxbILi4ELb1ELi0E" in name or "k_bxbILi4ELb1ELb0E" in name:      # (R = 4, FULL, no Adam blocks; the ADAM parameter is an int since abi 11)
            loops = [l for l in k["loops"] if l["mix"].get("valu_trans", 0) >= 8][:2]
    if loops:
        share = [l["trans_share_of_valu"] for l in loops]
        out["static_mix"] = {"loops": [{"valu": l["valu_total"], "trans": l["mix"].get("valu_trans", 0), "packed": l["mix"].get("valu_packed", 0),
                                        "pairs_per_lane_trip": l["mix"].get("valu_trans", 0) // (6 if l is loops[0] else 4) or None} for l in loops],
                             "trans_share_of_valu": {"six_transcendental_form": share[0], "four_transcendental_form": share[-1]}}
    cls = (rates.get("classes") or {}).get("waves_per_simd_2", {})
    cyc_full = cls.get("v_fma_f32", {}).get("cycles_at_2p4GHz", 4.0)
    cyc_pk = cls.get("v_pk_fma_f32", {}).get("cycles_at_2p4GHz", 4.0)
    cyc_tr = cls.get("v_exp_f32", {}).get("cycles_at_2p4GHz", 16.0)
    out["issue_cycles_per_wave_instr"] = {"full_rate": cyc_full, "packed_f32": cyc_pk, "transcendental": cyc_tr,
                                          "source": "profiles/valu_rates.json" if cls else "defaults (no measurement committed)"}
    for variant in ("bxb", "bxb+adam"):
        if variant not in kern_avg:
            continue
        e = {"avg_us": kern_avg[variant]["avg_us"]}
        c = sq.get(variant)
        if c and loops:
            insts = c["SQ_INSTS_VALU"]
            e["valu_wave_insts_per_launch"] = insts
            e["valu_thread_insts_per_pair"] = insts * 64.0 / (B * B)
            # bounds of the launch's issue time: every pair through the 4-transcendental form ... through the 6-transcendental form
            lim = {}
            for form, l in (("six", loops[0]), ("four", loops[-1])):
                v, t, pk = l["valu_total"], l["mix"].get("valu_trans", 0), l["mix"].get("valu_packed", 0)
                per_inst = ((v - t - pk) * cyc_full + pk * cyc_pk + t * cyc_tr) / v
                lim[form] = insts * per_inst / 1024.0 / 2.4e3
            e["issue_limit_us"] = {"all_pairs_six_transcendental_form": lim["six"], "all_pairs_four_transcendental_form": lim["four"]}
            e["frac"] = {"vs_six_form": lim["six"] / e["avg_us"], "vs_four_form": lim["four"] / e["avg_us"]}
            if c.get("SQ_INSTS_VALU_TRANS_F32"):
                # the transcendental count itself, where this rocprofv3 has the counter: the launch's own mix of the two forms
                t_ = c["SQ_INSTS_VALU_TRANS_F32"]
                pk_share = sum(l["mix"].get("valu_packed", 0) for l in loops) / float(sum(l["valu_total"] for l in loops))
                pk_ = pk_share * insts
                lim_m = ((insts - t_ - pk_) * cyc_full + pk_ * cyc_pk + t_ * cyc_tr) / 1024.0 / 2.4e3
                e["transcendental_share_measured"] = t_ / insts
                e["transcendentals_per_pair"] = t_ * 64.0 / (B * B)
                e["issue_limit_us"]["measured_mix"] = lim_m
                e["frac"]["vs_measured_mix"] = lim_m / e["avg_us"]
            if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
                e["pmc"] = {k_: c[k_] for k_ in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES",
                                                 "SQ_INSTS_VALU_TRANS_F32", "avg_ns_under_pmc") if k_ in c}
        e["gevals_per_s"] = 2.0 * B * B / (e["avg_us"] * 1e-6) / 1e9
        out[variant] = e
    return out


def self_launch(n):
    """Re-run this script as n ranks of one node: python -m torch.distributed.run --nnodes=1 --nproc-per-node n
    --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>.  Exits with the launcher's status."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N ...` as typed: become the launcher.  One process per GPU under torch.distributed.run
        # (rendezvous on 127.0.0.1, a free port), same arguments; rank 0's JSON line is this process's stdout.
        return self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or without a launcher: bench.py "
                         "starts its own ranks)" % (args.gpus, world, args.gpus))
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

    if args.workload == "config4":
        return bench_config4(args, rank, world, dev)
    if args.workload == "yelp2018" and not args.mf:
        return bench_lgcn(args, rank, world, dev)
    cfg = synth.WORKLOADS[args.workload]
    B, d = cfg["batch"], cfg["d"]
    if args.eval_train_steps < 0:
        args.eval_train_steps = 10 * (cfg["n_train"] // B + 1)
    kind = ops.LOSS_RUBIBCEBOTH if args.train == "rubibceboth" else ops.LOSS_NORMALBCE
    gen = torch.Generator(device=dev).manual_seed(12345 + rank)
    P = synth.xavier_table(cfg["n_users"], d, gen, dev)
    Q = synth.xavier_table(cfg["n_items"], d, gen, dev)
    w = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
    hyper = ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
    state = ops.MFState(P, Q, w, wu, hyper, B)
    # The batch pool does NOT depend on --steps / --warmup: the evaluator leg trains the model through it between its
    # timed evaluations, and `--steps 20` must evaluate the same model as `--steps 200` (round 3: 25 batches in the
    # driver's call, 220 in the README's -- every seeded evaluation of the 25-batch model was repaired).
    n_batches = N_BATCHES
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
    alone, bxb_warm = [], []
    if kind == ops.LOSS_RUBIBCEBOTH and not args.no_defer:
        ops.timing_begin()
        for s_ in range(10):
            k_ = s_ % n_batches
            state.step(kind, batches[k_, 0], batches[k_, 1], batches[k_, 2], loss_log[k_], defer=False)
        marks_alone = ops.timing_end(max_n=64)
        alone = [ms for name, ms in marks_alone if name == "adam_dense"]
        bxb_warm = [ms for name, ms in marks_alone if name == "bxb"][2:]    # the (B,B) launch WITHOUT Adam blocks, warm (complete steps)
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
    if bxb_warm and "bxb" in kern_avg:
        # in deferred mode the (B,B) launch without Adam blocks runs once per region, cold (the first step's): its warm time
        # comes from the complete steps above (same kernel, same batches), which is what rocprofv3 averages
        kern_avg["bxb"]["cold_first_launch_us"] = kern_avg["bxb"]["avg_us"]
        kern_avg["bxb"]["avg_us"] = max(1e3 * sum(bxb_warm) / len(bxb_warm) - event_overhead_us, 0.1)
        kern_avg["bxb"]["avg_us_source"] = "%d warm launches of complete (non-deferred) steps, this run" % len(bxb_warm)
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

    # theta, m, v of both tables + the gradient tables: what a step touches, step after step
    working_set = 4 * d * (cfg["n_users"] + cfg["n_items"]) * 4
    resident = ("infinity-cache: the %.0f MB working set of a step stays in the 256 MiB Infinity Cache from step to step, so `achieved` is "
                "algorithmic bytes / time against the HBM peak, NOT evidence of HBM traffic at that rate; the same pass on tables that "
                "cannot stay on chip is roofline_aux.adam_dense_out_of_cache (measured in this run)" % (working_set / 1e6)
                ) if working_set < 256 * 2 ** 20 else "hbm: the working set (%.0f MB) exceeds the 256 MiB Infinity Cache" % (working_set / 1e6)

    def roof(name, avg_us, ab):
        gbps = ab / (avg_us * 1e-6) / 1e9
        tr = pmc.get(args.workload, {}).get(name)
        return {"kernel": name, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbps / HBM_PEAK_GBS, "traffic": tr,
                "traffic_source": (REPLAYED % ("pmc_latest.json[%s][%s]" % (args.workload, name))) if tr else None,
                "avg_us": avg_us, "avg_us_source": "HIP events on the launch stream, this run",
                "us_per_step": per_step_us.get(name), "algorithmic_bytes": ab,
                "working_set_bytes": working_set, "resident": resident}
    if "algorithmic_bytes" in kern_avg[dom]:
        roofline = roof(dom, kern_avg[dom]["avg_us"], kern_avg[dom]["algorithmic_bytes"])
        if dom == "bxb+adam":
            roofline["note"] = ("the (B,B) launch with the dense Adam pass of the previous step riding in it: HBM bytes = "
                                "that pass (24*d*rows); the launch also does the 2*B^2 fused-BCE evaluations, which move "
                                "no HBM bytes (VALU-bound: kernels['bxb'] is the same launch without the Adam blocks)")
    else:   # a launch without HBM work dominates (e.g. non-deferred bxb): report it as such, no bandwidth claim
        roofline = {"kernel": dom, "bound": "valu", "achieved": kern_avg[dom].get("gevals_per_s"), "peak": None,
                    "unit": "G fused-BCE evaluations/s", "frac": None, "traffic": pmc.get(args.workload, {}).get(dom),
                    "traffic_source": (REPLAYED % "pmc_latest.json") if pmc.get(args.workload, {}).get(dom) else None,
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

    # The same dense Adam pass where the tables CANNOT stay in the Infinity Cache: 2.4 M rows of d floats (theta, m, v, gradient
    # tables; 3.7 GB of algorithmic bytes per step), complete `normalbce` steps, event-timed here.  The out-of-cache companion of `roofline`.
    if not args.no_defer:
        try:
            big_u, big_i = 1_900_000, 500_000
            gb = torch.Generator(device=dev).manual_seed(7)
            big = ops.MFState(synth.xavier_table(big_u, d, gb, dev), synth.xavier_table(big_i, d, gb, dev), w.clone(), wu.clone(),
                              hyper, B, lazy_period=1)
            bb = synth.train_batches(4, big_u, big_i, B, gb, dev)
            for k_ in range(3):
                big.step(ops.LOSS_NORMALBCE, bb[k_ % 4, 0], bb[k_ % 4, 1], bb[k_ % 4, 2])
            ops.timing_begin()
            for k_ in range(8):
                big.step(ops.LOSS_NORMALBCE, bb[k_ % 4, 0], bb[k_ % 4, 1], bb[k_ % 4, 2])
            ms_big = [ms for name, ms in ops.timing_end(max_n=64) if name == "adam_dense"]
            ab_big = 24 * d * (big_u + big_i)
            us_big = max(1e3 * sum(ms_big) / len(ms_big) - event_overhead_us, 0.1)
            aux["adam_dense_out_of_cache"] = {
                "kernel": "adam_dense", "bound": "hbm", "achieved": ab_big / (us_big * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ab_big / (us_big * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_us": us_big, "algorithmic_bytes": ab_big,
                "working_set_bytes": 4 * d * (big_u + big_i) * 4, "resident": "hbm (working set 10x the Infinity Cache)",
                "launches_sampled": len(ms_big), "source": "measured in this run",
                "note": "the pass `roofline` prices, on %d + %d rows of d=%d: every byte comes from and goes to HBM" % (big_u, big_i, d)}
            del big, bb
            torch.cuda.empty_cache()
        except Exception as e:                                   # (a diagnostic leg must not take the line with it)
            aux["adam_dense_out_of_cache"] = {"error": repr(e)[:200]}

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
    ret, roofline_eval, roofline_eval_bf16 = {}, None, None
    if not args.no_eval:
        def one_model():
            if world > 1:      # replicas trained on different batches: evaluate ONE model (rank 0's), item-sharded
                for t in (state.P, state.Q, state.w, state.wu):
                    torch.distributed.broadcast(t, 0)
        def eval_suite(filt):
            """the evaluator measurements under one candidate filter of the listing pass (Evaluator.filter): timed
            evaluations with training in between, the unseeded time, per-kernel events of a sampled and a seeded ranking"""
            one_model()
            ev = Evaluator(mask_lists, gt_lists, cfg["n_items"], dev)
            ev.filter = filt
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
                if os.environ.get("MACR_BENCH_DEBUG"):
                    print("settle", r_, ev.last_eval_info(), "skip", ev._seed_skip, "backoff", ev._seed_backoff, file=sys.stderr)
            # Timed evaluations: as in a training run, the tables MOVE between two evaluations (20 untimed training steps
            # here), and an evaluation seeds its thresholds with the ids the previous one returned (Evaluator.rank_local).
            ev_elapsed, ev_modes, ev_times, ev_dev_us = 0.0, [], [], []
            for r_ in range(args.eval_reps):
                if args.eval_train_steps > 0:
                    run_steps(args.eval_train_steps, args.eval_train_steps * r_)
                    one_model()
                barrier(); torch.cuda.synchronize()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                g0.record()
                ret = run_eval()
                g1.record()
                torch.cuda.synchronize(); barrier()
                ev_times.append(sharding.max_over_ranks(time.perf_counter() - t0, dev))
                ev_dev_us.append(1e3 * g0.elapsed_time(g1))      # what the stream was busy with between the two records
                ev_elapsed += ev_times[-1]
                ev_modes.append(ev.last_eval_info())
            if os.environ.get("MACR_BENCH_DEBUG"):
                print("eval times (us)", filt, [int(1e6 * t_) for t_ in ev_times], file=sys.stderr)
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
            seeded_run = {"seeded": ev.last_eval_info()["seeded"], "query_blocks_relisted": ev.last_eval_info()["query_blocks_relisted"],
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
            rank_kernels = ("score_sample", "score_sample_b", "tau", "tau_seed", "bf16_prep", "bf16_prep_c", "bf16_prep+tau_seed", "score_stream", "score_stream_b", "select", "select_b",
                            "repair_plan", "score_sample2", "tau2", "score_stream2", "select2", "score_topk")
            st_us = 1e3 * sum(ek.get(k, 0.0) for k in rank_kernels)
            stream_us = 1e3 * ek.get("score_stream_b" if filt in LP_FILTERS else "score_stream", float("nan"))
            roofline_eval = {"kernel": "+".join(k for k in rank_kernels if k in ek), "bound": "mfma",
                             "achieved": flops / (st_us * 1e-6) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "avg_us": st_us, "flops": flops, "traffic": pmc.get(args.workload, {}).get("score_stream"),
                             "traffic_source": (REPLAYED % ("pmc_latest.json[%s][score_stream]" % args.workload)) if pmc.get(args.workload, {}).get("score_stream") else None,
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

            return {"ret": ret, "eval_users_per_s": eval_users_per_s, "ev_elapsed": ev_elapsed, "ev_unseeded_ms": ev_unseeded_ms,
                    "ev_modes": ev_modes, "ev_times": ev_times, "ev_dev_us": ev_dev_us, "roofline_eval": roofline_eval}

        # "f32": the (U, N) product on the fp32 matrix cores -- `roofline_eval`, priced against the fp32 MFMA peak as in
        # the earlier rounds.  "bf16": the Evaluator's default, a bf16 candidate filter with fp32 re-scoring (the same
        # ranking bit for bit, tests/): the headline `eval_users_per_s`.
        # Both suites start from the SAME model and train it through the same batches between their evaluations (the
        # parameters and optimizer slots are put back in place -- captured graphs keep their pointers): how far a model has
        # come decides how often seeded thresholds go stale, and the second suite must not meet an older model.
        snap_names = ("P", "Q", "w", "wu", "mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu", "adam_pow")
        state.flush()
        snapshot = {n_: getattr(state, n_).clone() for n_ in snap_names}
        suite_f32 = eval_suite("f32")
        suite = suite_f32
        if lp_filter_name() in LP_FILTERS:
            state.flush()
            for n_, t_ in snapshot.items():
                getattr(state, n_).copy_(t_)
            suite = eval_suite(lp_filter_name())
        del snapshot
        ret, eval_users_per_s, ev_elapsed = suite["ret"], suite["eval_users_per_s"], suite["ev_elapsed"]
        ev_unseeded_ms, ev_modes = suite["ev_unseeded_ms"], suite["ev_modes"]
        roofline_eval = suite_f32["roofline_eval"]
        roofline_eval["filter"] = "f32"
        roofline_eval["eval_users_per_s"] = suite_f32["eval_users_per_s"]
        roofline_eval["eval_ms_unseeded"] = suite_f32["ev_unseeded_ms"]
        roofline_eval["eval_modes"] = suite_f32["ev_modes"]
        roofline_eval_bf16 = None
        if suite is not suite_f32:
            rb = suite["roofline_eval"]
            # what the bf16 matrix cores execute: three products per fp32 multiply-add (hi*hi + hi*lo + lo*hi) and one more
            # MFMA per tile for the bias slab that carries the score epilogue (k_score_stream_c): 3 + 16/d
            # (the fp16 filter: one product and the slab, 1 + 16/d)
            sb = rb["stream"]["avg_us"]
            mult = (1.0 if lp_filter_name() == "f16" else 3.0) + 16.0 / d
            roofline_eval_bf16 = {"filter": lp_filter_name(), "bound": "mfma-" + lp_filter_name(), "kernel": rb["kernel"], "avg_us": rb["avg_us"],
                                  "kernels_us": rb["kernels_us"], "seeded": rb["seeded"], "mode": rb["mode"],
                                  "stream": {"avg_us": sb, "executed_flops": mult * rb["flops"],
                                             "achieved": mult * rb["flops"] / (sb * 1e-6) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                                             "unit": "TFLOP/s", "frac": mult * rb["flops"] / (sb * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
                                  "eval_users_per_s": suite["eval_users_per_s"],
                                  "speedup_vs_f32_filter": suite["eval_users_per_s"] / suite_f32["eval_users_per_s"],
                                  "note": "same ranking as the f32 filter, bit for bit: reduced-precision products only pick candidates, "
                                          "the best 64 per query are re-scored in fp32 (DESIGN.md, ranking note)"}
            for key in ("frac",):
                rb["seeded"].pop(key, None)

    # both filters side by side, in one short object near the top of the line
    eval_summary = None
    if not args.no_eval:
        def _row(su):
            modes = su["ev_modes"]
            ts = sorted(su["ev_times"])
            return {"users_per_s": su["eval_users_per_s"], "ms_per_eval": 1e3 * su["ev_elapsed"] / args.eval_reps,
                    # every timed evaluation is a host round trip (replay, synchronise) behind 2 010 training steps: the mean is
                    # what users_per_s is computed from, median / min / max show what host hiccups did to it
                    "ms_per_eval_median": 1e3 * ts[len(ts) // 2], "ms_per_eval_min": 1e3 * ts[0], "ms_per_eval_max": 1e3 * ts[-1],
                    # host_gap_us: wall time of an evaluation (replay call -> means on the host) minus the time between two events
                    # recorded on the stream around it -- launch of the replay, the wait for the pinned-memory results, the
                    # python between them.  A property of the box's host as much as of the code: what separates boxes.
                    "device_us_per_eval": float(np.mean(su["ev_dev_us"])),
                    "host_gap_us": float(np.mean([1e6 * t_ - d_ for t_, d_ in zip(su["ev_times"], su["ev_dev_us"])])),
                    "ms_unseeded": su["ev_unseeded_ms"], "evaluations": len(modes),
                    "seeded": sum(1 for m in modes if m["seeded"]),
                    "repaired": sum(1 for m in modes if m["query_blocks_relisted"] > 0 or m["exact_fallback"]),
                    "query_blocks_relisted": sum(m["query_blocks_relisted"] for m in modes)}
        eval_summary = {"default_filter": lp_filter_name() if suite is not suite_f32 else "f32", "f32": _row(suite_f32),
                        (lp_filter_name() if suite is not suite_f32 else "bf16"): _row(suite) if suite is not suite_f32 else None,
                        "train_steps_between_evaluations": args.eval_train_steps, "batch_pool": n_batches,
                        "roofline_eval_frac_f32_sampled": roofline_eval["frac"]}

    # ------------------------------------------------------------- CPU baseline: ports of the reference path on the host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_mf(args, kind, cfg, P, Q, w, wu, batches[:8].cpu().numpy(), users, mask_lists, gt_lists, Ks)

    # ------------------------------------------------------------- N > 1: what the process group reports, the evaluation's
    # one collective timed on its own, and the configs[4] leg (the training path that SHARDS: one model, rows over the ranks)
    multi, c4_line = None, None
    if world > 1:
        Uq, Kq = len(users), Ks[0]
        lv = torch.zeros((Uq, Kq), dtype=torch.float32, device=dev)
        li = torch.zeros((Uq, Kq), dtype=torch.int32, device=dev)
        for _ in range(3):
            sharding.gather_topk(lv, li)
        torch.cuda.synchronize(); barrier()
        ag = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sharding.gather_topk(lv, li); e1.record()
            ag.append((e0, e1))
        torch.cuda.synchronize()
        ag_us = float(np.median([1e3 * a.elapsed_time(b) for a, b in ag]))
        multi = {"nranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                 "devices": torch.cuda.device_count(),
                 "collectives_us": {"eval_all_gather_topk": sharding.max_over_ranks(ag_us, dev)},
                 "eval_all_gather_bytes_per_rank": Uq * Kq * 8,
                 "note": "event time of the evaluation's one collective (pack + all_gather_into_tensor + unpack of the (U,K) "
                         "(score, id) lists), median of 20, max over ranks"}
        if not args.no_config4:
            # (a failure here -- this part has run on one GPU and under gloo only -- must not take the line with it)
            try:
                c4_line = bench_config4(args, rank, world, dev, emit=False)
            except Exception as e:                                    # noqa: BLE001
                c4_line = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                print("[bench] config4 part failed on rank %d: %s" % (rank, c4_line["error"]), file=sys.stderr)

    cli_test = None
    if rank == 0 and world == 1 and args.through_cli_test:
        import subprocess
        try:
            r_ = subprocess.run([sys.executable, os.path.join(REPO, "tools", "cli_test_cost.py"), "--workload", args.workload],
                                capture_output=True, text=True, timeout=600)
            cli_test = json.loads(r_.stdout.strip().splitlines()[-1])
        except Exception as e:                                     # noqa: BLE001
            cli_test = {"error": repr(e)[:200]}
    if rank == 0:
        out = {
            "metric": "train interactions/sec + eval users/sec (full-catalog top-K@20)",
            "value": value, "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "eval_users_per_s": eval_users_per_s,
            "nranks": 1 if multi is None else multi["nranks"], "multi_gpu": multi,
            "config": {"workload": "%s-shape MACR-MF %s d=%d batch=%d c=%g (n_users=%d, n_items=%d); synthetic "
                                   "Xavier tables, Zipf positives, batches as sampled (grouped on the device inside the step)" % (args.workload, args.train, d, B, cfg["c"],
                                                                      cfg["n_users"], cfg["n_items"]),
                       "parallelism": "replicas x%d (train) / item-sharded x%d + RCCL all-gather (eval)" % (world, world),
                       "global_batch": B * world},
            "eval": eval_summary,
            "sharded_figure": {"name": "eval_users_per_s", "value": eval_users_per_s, "scaling": "strong",
                               "note": "`value` counts N independent training replicas (these configs' step fits one GPU: "
                                       "replicas only, SURVEY.md 8e); the path of this workload that SHARDS over the ranks is the "
                                       "item-sharded evaluation -- its users/s is the figure a scaling curve should be read from. "
                                       "--workload config4 trains ONE row-sharded model (strong scaling)"},
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
            "roofline": roofline, "roofline_step": roofline_step, "roofline_aux": aux,
            "roofline_bxb": roofline_bxb(args.workload, B, kern_avg) if kind == ops.LOSS_RUBIBCEBOTH else None,
            "end_to_end": end_to_end, "cli_test": cli_test,
            "roofline_eval": roofline_eval, ("roofline_eval_" + (roofline_eval_bf16 or {}).get("filter", "bf16")): roofline_eval_bf16, "cpu_baseline": cpu,
            "last_losses": [float(x) for x in losses[(args.warmup + args.steps - 1) % n_batches]],
        }
        if c4_line is not None:
            # configs[4] in the same line: interactions/s of ONE model row-sharded over the ranks (strong scaling), its
            # per-step collectives and the item-sharded evaluation of 100 000 query users against 1 M items
            out["config4"] = c4_line if "error" in c4_line else {k: c4_line.get(k) for k in ("value", "unit", "ms_per_step", "scaling", "config", "rows_per_rank",
                                                          "collectives_ms", "wire_bytes_per_step", "kernels", "roofline",
                                                          "roofline_step", "eval_users_per_s", "eval_ms_per_pass", "eval_users",
                                                          "roofline_eval", "timed_regions")}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
