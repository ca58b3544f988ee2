#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops from the gfx950 code object inside libmacr_hip.so.

  python tools/isa_mix.py k_bxbILi4ELb1ELb0E [more mangled-name fragments ...]  > profiles/r05_bxb_isa_mix.json

For every kernel whose mangled name contains the fragment: every loop (a backward s_cbranch to an earlier address) with its
instruction classes -- VALU full rate, VALU transcendental (v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos: quarter rate on
CDNA), packed fp32 (v_pk_*: two results per lane), DPP/cross-lane, SALU, VMEM, LDS -- and the SIMD issue cycles one trip of
the loop costs a wave64 (full rate 4, transcendental 16: MI355X_MICROARCH.md, 16 lanes per SIMD and cycle).  bench.py reads the
committed JSON for `roofline_bxb` next to the PMC instruction counts (profiles/pmc_sq_latest.json): the static mix says which
share of the counted VALU instructions is quarter-rate."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def disassemble(lib):
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, "lib.so")
    os.symlink(lib, local)
    subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = ""
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" in f:
            out += subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
    return out


def classify(op):
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "ds_bpermute", "ds_permute", "ds_swizzle")):
        return "cross_lane"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(asm, frag):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if cur and frag in cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur:
            body.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if cur and frag in cur:
        yield cur, body


def loops(body):
    addr_index = {a: k for k, (a, _, _) in enumerate(body)}
    for k, (a, op, args) in enumerate(body):
        if not op.startswith("s_cbranch") and op != "s_branch":
            continue
        m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", args) or re.search(r"<[^>+]*>", args)
        tgt = None
        if m and m.lastindex:
            tgt = body[0][0] + int(m.group(1), 16)
        if tgt is None:                      # encoded simm16: signed dword offset from the next instruction
            m2 = re.match(r"(-?\d+)", args)
            if not m2:
                continue
            off = int(m2.group(1))
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
        if tgt in addr_index and tgt <= a:
            yield addr_index[tgt], k


def main():
    lib = os.path.join(ROOT, "macr_amd", "csrc", "libmacr_hip.so")
    asm = disassemble(lib)
    out = {}
    for frag in sys.argv[1:]:
        for name, body in kernels(asm, frag):
            ls = []
            for lo, hi in sorted(set(loops(body))):
                mix = {}
                dpp = 0
                for a, op, args in body[lo:hi + 1]:
                    c = classify(op)
                    mix[c] = mix.get(c, 0) + 1
                    if "dpp" in args or "row_" in args or "quad_perm" in args:
                        dpp += 1
                valu = mix.get("valu", 0) + mix.get("valu_packed", 0) + mix.get("valu_trans", 0) + mix.get("cross_lane", 0)
                ls.append({"instructions": hi - lo + 1, "mix": mix, "valu_total": valu, "dpp_modified": dpp,
                           "trans_share_of_valu": mix.get("valu_trans", 0) / valu if valu else 0.0,
                           "simd_issue_cycles_per_trip": 4 * (valu - mix.get("valu_trans", 0)) + 16 * mix.get("valu_trans", 0)})
            ls.sort(key=lambda l: -l["mix"].get("valu_trans", 0))
            out[name] = {"instructions": len(body), "loops": ls[:4]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
