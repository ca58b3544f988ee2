"""Seed sweep of the reference README's Addressa commands (README.md:36-52,:66-82) through the drop-in CLIs.

    python tools/seed_sweep.py --out gpurun_out/seed_sweep.json [--seeds 12] [--jobs 4] [--lgcn_ref_seeds 3]

For every seed: run the CLI, read its evaluation lines, take the evaluation with the best HR@20 (the reference's
early-stopping criterion, `>=` for MF train.py:313-330, `>` for LightGCN helper.py:35-50) and record HR / recall / NDCG
there.  Output: per-run records + mean / sd per (model, sampler) and where the published rows of README.md:89,:94 fall
(z = (published - mean) / sd).  Extra `--ablate` runs repeat the MF sweep with one candidate changed at a time.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PUBLISHED = {  # /root/reference/README.md:89 (LightGCN_Adressa), :94 (MF Adressa)
    "mf": {"hr": 0.13561, "recall": 0.10612, "ndcg": 0.04667},
    "lgcn": {"hr": 0.16356, "recall": 0.12967, "ndcg": 0.06071},
}
MF_CMD = ("python ./macr_mf/train.py --dataset addressa --batch_size 1024 --cuda 0 --saveID {sid} --log_interval 10 "
          "--lr 0.001 --check_c 1 --c 40 --train rubibceboth --test rubi --alpha 1e-3 --beta 1e-3 --save_flag 0 "
          "--seed {seed} --sampler {sampler} {extra}")
LGCN_CMD = ("python macr_lightgcn/LightGCN.py --data_path data/ --dataset addressa --verbose 1 --layer_size [64,64] "
            "--Ks [20] --loss bceboth --test rubiboth --c 40 --epoch 2000 --early_stop 1 --lr 0.001 --batch_size 1024 "
            "--gpu_id 0 --log_interval 10 --alpha 1e-2 --beta 1e-3 --save_flag 0 --saveID {sid} --seed {seed} "
            "--sampler {sampler} {extra}")
MF_LINE = re.compile(r"c:40\.00 \[.*?recall=\[([\d.]+), [\d.]+\], precision=\[([\d.]+), [\d.]+\], hit=\[([\d.]+), "
                     r"[\d.]+\], ndcg=\[([\d.]+), ")
LGCN_LINE = re.compile(r"c:40\.00 recall=\[([\d.]+), [\d.]+\], hit=\[([\d.]+), [\d.]+\], ndcg=\[([\d.]+), ")


def run(model, seed, sampler, extra=""):
    cmd = (MF_CMD if model == "mf" else LGCN_CMD).format(sid=1000 + seed, seed=seed, sampler=sampler, extra=extra)
    t0 = time.time()
    p = subprocess.run(cmd, shell=True, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=1800)
    evals = []
    for line in p.stdout.splitlines():
        if model == "mf":
            m = MF_LINE.search(line)
            if m:
                evals.append(dict(recall=float(m.group(1)), hr=float(m.group(3)), ndcg=float(m.group(4))))
        else:
            m = LGCN_LINE.search(line)
            if m:
                evals.append(dict(recall=float(m.group(1)), hr=float(m.group(2)), ndcg=float(m.group(3))))
    if p.returncode != 0 or not evals:
        return dict(model=model, seed=seed, sampler=sampler, extra=extra, rc=p.returncode, tail=p.stdout[-800:])
    best = 0
    for k, e in enumerate(evals):               # MF keeps the LAST of equal bests (>=), LightGCN the first (>)
        if (e["hr"] >= evals[best]["hr"]) if model == "mf" else (e["hr"] > evals[best]["hr"]):
            best = k
    rec = dict(model=model, seed=seed, sampler=sampler, extra=extra, rc=0, n_evals=len(evals), best_eval=best,
               wall_s=round(time.time() - t0, 1), **evals[best])
    rec["final"] = evals[-1]
    rec["hr_curve"] = [e["hr"] for e in evals]
    rec["ndcg_curve"] = [e["ndcg"] for e in evals]
    return rec


def summarise(recs, model):
    out = {}
    ok = [r for r in recs if r["rc"] == 0]
    for key in ("hr", "recall", "ndcg"):
        x = np.array([r[key] for r in ok], dtype=np.float64)
        mean, sd = float(x.mean()), float(x.std(ddof=1)) if len(x) > 1 else 0.0
        pub = PUBLISHED[model][key]
        out[key] = dict(mean=round(mean, 6), sd=round(sd, 6), min=float(x.min()), max=float(x.max()), published=pub,
                        z=round((pub - mean) / sd, 2) if sd > 0 else None,
                        inside_2sd=bool(abs(pub - mean) <= 2 * sd), rel_diff=round(mean / pub - 1, 4))
    out["n"] = len(ok)
    out["failed"] = len(recs) - len(ok)
    out["best_eval_index"] = [r["best_eval"] for r in ok]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/seed_sweep.json")
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--lgcn_ref_seeds", type=int, default=3)
    ap.add_argument("--mf_sampler", default="reference")
    ap.add_argument("--skip_lgcn", action="store_true")
    ap.add_argument("--ablate", nargs="*", default=[], help="name=extra-flags groups for additional MF sweeps")
    args = ap.parse_args()
    seeds = [12345] + [1000 + 37 * k for k in range(1, args.seeds)]
    groups = [("mf/%s" % args.mf_sampler, "mf", args.mf_sampler, "", seeds)]
    if args.mf_sampler != "device":
        groups.append(("mf/device", "mf", "device", "", seeds))
    if not args.skip_lgcn:
        groups.append(("lgcn/device", "lgcn", "device", "", seeds))
        if args.lgcn_ref_seeds:
            groups.append(("lgcn/reference", "lgcn", "reference", "", seeds[:args.lgcn_ref_seeds]))
    for a in args.ablate:
        name, extra = a.split("=", 1)
        groups.append(("mf/%s/%s" % (args.mf_sampler, name), "mf", args.mf_sampler, extra, seeds))
    result = {"published_source": "reference README.md:89 (LightGCN_Adressa), :94 (MF Adressa)", "seeds": seeds,
              "commands": {"mf": MF_CMD, "lgcn": LGCN_CMD}, "groups": {}}
    with ThreadPoolExecutor(args.jobs) as pool:
        futs = {name: [pool.submit(run, model, s, sampler, extra) for s in sd] for name, model, sampler, extra, sd in groups}
        for name, model, sampler, extra, sd in groups:
            recs = [f.result() for f in futs[name]]
            result["groups"][name] = {"summary": summarise(recs, model) if any(r["rc"] == 0 for r in recs) else None,
                                      "runs": recs}
            print(name, json.dumps(result["groups"][name]["summary"]), flush=True)
            os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
            with open(args.out, "w") as f:
                json.dump(result, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
