#!/usr/bin/env python3
"""Condense gpurun_out/prof_<run> (tools/profile.sh) into the small, committed files under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (all kernels of the run)
  profiles/<tag>_pmc.json           per-kernel averages of the PMC passes
  profiles/pmc_latest.json          {workload: {kernel: HBM bytes per launch}} read by bench.py (`traffic`)
  profiles/pmc_sq_latest.json       {workload: {kernel: {SQ_* counter: average per launch}}} read by bench.py (`roofline_bxb`)
Usage:  python tools/summarize_profile.py condense <dir>        (on the GPU box: raw counter tables -> averages)
        python tools/summarize_profile.py <run> <tag> [workload] (here: gpurun_out/prof_<run> -> profiles/<tag>_*)
HBM bytes = FETCH_FACTOR*FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE counts 128-byte fabric requests at 64 bytes
(MI355X_MICROARCH.md, HBM section).  The factor is calibrated per load shape in profiles/fetch_calib.json (tools/fetch_calib.sh:
1 GiB read once as a stream and as a permutation of 256-byte rows, global_load_dword and dwordx4): 2.000 in all four, so one
factor serves every kernel here, the SpMM's dword gathers included; WRITE_SIZE is uncalibrated."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")


def fetch_factor(shape=None):
    """true bytes per counted FETCH_SIZE byte: profiles/fetch_calib.json (per load shape; the mean when no shape is named --
    the four measured shapes agree to 1e-4), 2.0 when the calibration file is missing"""
    try:
        shapes = json.load(open(os.path.join(DST, "fetch_calib.json")))["shapes"]
        vals = [v["true_over_counter"] for k, v in shapes.items() if shape in (None, k) and "true_over_counter" in v]
        return sum(vals) / len(vals)
    except Exception:
        return 2.0


def short(name):
    m = re.search(r"macr::k_([a-z0-9_]+)", name)
    if not m:
        return None
    k = m.group(1)
    if k == "score_stream":          # k_score_stream<D, KIND, MODE, NC, REPAIR>: MODE 0 is the sampling pass, REPAIR the
        t = re.search(r"k_score_stream<\s*\d+\s*,\s*\d+\s*,\s*(\d+)\s*,\s*\d+\s*,\s*(true|false|0|1)", name)   # repair round
        if t and t.group(1) == "0":
            k = "score_sample"
        if t and t.group(2) in ("true", "1"):
            k += "2"
    if k in ("score_stream_b", "score_stream_c"):      # the bf16 listing pass (k_score_stream_c since round 4); REPAIR instantiations apart
        t = re.search(r"k_score_stream_[bc]<\s*\d+\s*,\s*\d+\s*,\s*(true|false|0|1)", name)
        k = "score_stream2" if (t and t.group(1) in ("true", "1")) else "score_stream_b"
    if k in ("select_b", "score_sample_b"):
        t = re.search(r"k_(?:select_b|score_sample_b)<\s*\d+\s*,\s*\d+\s*,\s*(true|false|0|1)", name)
        if t and t.group(1) in ("true", "1"):
            k = "select2" if k == "select_b" else "score_sample2"
    if k == "prep_tau_seed":
        k = "bf16_prep+tau_seed"
    if k == "spmm_row":              # k_spmm_row<D, SPARSE>: 1 = flagged output rows only, 2 = row-sparse input
        t = re.search(r"k_spmm_row<\s*\d+\s*,\s*(\d+)\s*,\s*(true|false|0|1)", name)
        k = "spmm_csr" + {"1": "_rows", "2": "_sparse"}.get(t.group(1) if t else "0", "")   # (the names the benches use)
        if t and t.group(2) in ("true", "1"):
            k += "+adam"                 # FUSE: the optimizer in the epilogue of the last backward layer
    if k == "adam_dense" and re.search(r"k_adam_dense<\s*(true|1)\s*>", name):
        k = "adam_indexed"           # the pass that sums the staged gradient rows of a large batch itself
    if k == "seg_scan":
        k = "seg_index"
    if k == "select" and re.search(r"k_select<\s*(true|1)\s*>", name):
        k = "select2"
    if k == "tau" and re.search(r"k_tau<\s*\d+\s*,\s*(true|1)\s*>", name):
        k = "tau2"
    if k == "bxb":                   # k_bxb<R, FULL, ADAM>: ADAM=true carries the deferred Adam blocks
        t = re.search(r"k_bxb<\s*\d+\s*,\s*(?:true|false|\d+)\s*,\s*(true|1|2)", name)     # (2: the lazy pass)
        if t:
            k = "bxb+adam"
    return k


def pmc_avgs(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def condense(src):
    """raw per-dispatch counter tables -> <sub>/avgs.json (+ the kernel durations of the SQ pass), raw tables removed"""
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_valu"):
        p = os.path.join(src, sub, "bench_counter_collection.csv")
        if os.path.exists(p):
            json.dump(pmc_avgs(p), open(os.path.join(src, sub, "avgs.json"), "w"))
            os.remove(p)
    trace = os.path.join(src, "pmc_sq", "bench_kernel_trace.csv")
    if os.path.exists(trace):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            k = short(r["Kernel_Name"])
            if k:
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        json.dump({k: sum(v) / len(v) for k, v in dur.items()}, open(os.path.join(src, "pmc_sq", "durations.json"), "w"))
    for sub in ("trace", "pmc_fetch", "pmc_write", "pmc_sq", "pmc_valu"):
        t = os.path.join(src, sub, "bench_kernel_trace.csv")
        if os.path.exists(t):
            os.remove(t)


def main(run, tag, workload=None):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + run)
    os.makedirs(DST, exist_ok=True)
    shutil.copyfile(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(DST, tag + "_kernel_stats.csv"))
    out_txt = os.path.join(src, "stdout_trace.txt")
    if os.path.exists(out_txt) and os.path.getsize(out_txt):
        shutil.copyfile(out_txt, os.path.join(DST, tag + "_under_rocprof.json"))
    pmc = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_valu"):
        p = os.path.join(src, sub, "avgs.json")
        if os.path.exists(p):
            for k, d in json.load(open(p)).items():
                pmc.setdefault(k, {}).update(d)
    traffic = {}
    for k, d in pmc.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (fetch_factor() * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
            traffic[k] = d["hbm_bytes_per_launch"]
    # GRBM_GUI_ACTIVE per nanosecond of kernel time: proportional to the clock the kernel ran at (DVFS).  This
    # rocprofv3 sums the counter over an unknown number of instances, so only RATIOS between kernels are used.
    durp = os.path.join(src, "pmc_sq", "durations.json")
    if os.path.exists(durp):
        dur = json.load(open(durp))
        for k, d in pmc.items():
            if k in dur and "GRBM_GUI_ACTIVE" in d:
                d["avg_ns_under_pmc"] = dur[k]
                d["gui_active_per_ns"] = d["GRBM_GUI_ACTIVE"] / dur[k]
        top = max((d.get("gui_active_per_ns", 0) for d in pmc.values()), default=0)
        for d in pmc.values():
            if top and "gui_active_per_ns" in d:
                d["clock_rel_to_fastest_kernel"] = d["gui_active_per_ns"] / top
    json.dump(pmc, open(os.path.join(DST, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    if workload:
        latest_path = os.path.join(DST, "pmc_latest.json")
        latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
        latest[workload] = traffic
        json.dump(latest, open(latest_path, "w"), indent=1, sort_keys=True)
        # the instruction / cycle counters of the same run, per kernel: bench.py prices the (B,B) launch with them (roofline_bxb)
        sq_path = os.path.join(DST, "pmc_sq_latest.json")
        sq = json.load(open(sq_path)) if os.path.exists(sq_path) else {}
        sq[workload] = {k: {c: v for c, v in d.items() if c.startswith(("SQ_", "avg_ns"))} for k, d in pmc.items()}
        json.dump(sq, open(sq_path, "w"), indent=1, sort_keys=True)
    for k, d in sorted(pmc.items()):
        print(k, {c: round(v, 1) for c, v in d.items()})


if __name__ == "__main__":
    if sys.argv[1] == "condense":
        condense(sys.argv[2])
    else:
        main(*sys.argv[1:])
