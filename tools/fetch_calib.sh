#!/bin/bash
# FETCH_SIZE per load shape (tools/fetch_calib.hip) -> gpurun_out/fetch_calib.json   (copy to profiles/fetch_calib.json)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/tools/fetch_calib $ROOT/tools/fetch_calib.hip || exit 1
O=$ROOT/gpurun_out/fetch_calib; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o c -- $ROOT/tools/fetch_calib > $O/stdout.txt 2> $O/fetch.err
rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/req -o c -- $ROOT/tools/fetch_calib > /dev/null 2> $O/req.err
python3 - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_(stream|gather)<(\d)", r["Kernel_Name"])
        if m:
            agg["%s_dword%s" % (m.group(1), "x4" if m.group(2) == "4" else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"bytes_read_per_launch": 1 << 30, "shapes": {}}
for k, d in sorted(agg.items()):
    e = {c: sum(v) / len(v) for c, v in d.items()}
    if "FETCH_SIZE" in e:
        e["FETCH_SIZE_bytes"] = e["FETCH_SIZE"] * 1024
        e["true_over_counter"] = (1 << 30) / (e["FETCH_SIZE"] * 1024)
    out["shapes"][k] = e
json.dump(out, open("$ROOT/gpurun_out/fetch_calib.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
