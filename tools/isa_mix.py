#!/usr/bin/env python3
"""Instruction mix of the blend kernels' inner loops, read off the SHIPPED code object.

    python tools/isa_mix.py [--lib gaussiansplattingmlx_amd/libgsplat_hip.so] [--out profiles/r05_blend_isa_mix.json] [--dump DIR]

Takes the gfx950 images out of the library's .hip_fatbin section (clang offload bundles, one per translation unit),
disassembles them with llvm-objdump, finds the kernels named below and, in each, the loops (backward branches) of its body.
The INNER loop of a kernel is the innermost loop with the most v_exp_f32 per trip among the loops that hold one (the
per-splat sweep: every splat costs exponentials); its instructions are counted by issue class:

    full     v_mul / v_add / v_sub / v_fma / v_fmac / v_mov / integer add, and, shift ...    (~2.4 - 3.0 cycles per wave64 inst.)
    half     v_min / v_max / v_med3 / v_cmp* / v_cndmask / v_bfi                               (~4.2 - 4.5)
    quarter  v_exp / v_rcp / v_sqrt / v_rsq / v_log                                            (~8.3)
    pk       v_pk_*_f32 (two results per lane)                                                 (~4.8)
    dpp      any VALU instruction with a DPP control (row_shr, quad_perm, row_bcast ...)       (~4.2)
    swap     v_permlane*_swap, v_readlane, v_readfirstlane, v_writelane                        (4.1 - 13.6)
    lds      ds_*                                                                              (LDS pipe)
    vmem     global_* / buffer_* / flat_*                                                      (memory pipe)
    salu     s_* (scalar unit; issues beside the VALU)

The per-class costs are the measured issue intervals of profiles/r01c_microbench_issue_rates.txt (cycles per wave64
instruction per SIMD with 8 waves resident); `mix_cycles_per_valu_inst` = sum(count x cost) / sum(count) over the VALU
classes is what bench.py's `issue_model_frac` multiplies SQ_INSTS_VALU with.
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# cycles per wave64 instruction per SIMD, 8 waves resident (profiles/r01c_microbench_issue_rates.txt)
CLASS_COST = {"full": 2.4, "half": 4.3, "quarter": 8.3, "pk": 4.8, "dpp": 4.2, "swap": 13.6, "readlane": 4.1}
NOMINAL_COST = 2.0      # the guide's nominal issue interval of a wave64 VALU instruction on one SIMD (MI355X_MICROARCH.md)

KERNELS = {
    "blend_bwd_v2_kernel<64,false>": "_ZN2gs19blend_bwd_v2_kernelILi64ELb0EEE",
    "blend_bwd_v2_kernel<64,true>": "_ZN2gs19blend_bwd_v2_kernelILi64ELb1EEE",
    "blend_fwd_v2q_kernel<64,false>": "_ZN2gs20blend_fwd_v2q_kernelILi64ELb0EEE",
    "blend_fwd_v2q_kernel<64,true>": "_ZN2gs20blend_fwd_v2q_kernelILi64ELb1EEE",
    "blend_fwd_v2w_kernel<64,false>": "_ZN2gs20blend_fwd_v2w_kernelILi64ELb0EEE",
    "blend_fwd_v2w_kernel<64,true>": "_ZN2gs20blend_fwd_v2w_kernelILi64ELb1EEE",
}

HALF = re.compile(r"^v_(min|max|med3|cmp|cmpx|cndmask|bfi)")
QUARTER = re.compile(r"^v_(exp|rcp|sqrt|rsq|log|sin|cos)")
SWAP = re.compile(r"^v_(permlane\d*_swap|permlane|writelane)")
READLANE = re.compile(r"^v_(readlane|readfirstlane)")


def classify(mnemonic: str, operands: str) -> str:
    m = mnemonic
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if not m.startswith("v_"):
        return "other"
    if SWAP.match(m):
        return "swap"
    if READLANE.match(m):
        return "readlane"
    if re.search(r"\b(row_shr|row_shl|row_ror|quad_perm|row_bcast|row_mirror|row_half_mirror|wave_shr|wave_shl|row_newbcast|row_share|row_xmask)\b", operands) or m.endswith("_dpp"):
        return "dpp"
    if m.startswith("v_pk_"):
        return "pk"
    if QUARTER.match(m):
        return "quarter"
    if HALF.match(m):
        return "half"
    return "full"


def gfx950_images(lib):
    """The gfx950 ELF images of the library's .hip_fatbin section, in order."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "x.so")], check=True)
        d = open(fat, "rb").read()
    out = []
    pos = 0
    while True:
        pos = d.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", d, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, q)
            triple = d[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(d[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
LABEL = re.compile(r"^([0-9a-f]+) <([^>]+)>:")


def disassemble(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f.name], check=True, capture_output=True, text=True).stdout


def functions(asm):
    """{symbol: [(address, mnemonic, operands)]} (labels inside a function, L<n>, belong to it)."""
    funcs, cur = {}, None
    for ln in asm.splitlines():
        m = LABEL.match(ln)
        if m:
            if not m.group(2).startswith("L") or cur is None:
                cur = funcs.setdefault(m.group(2), [])
            continue
        m = LINE.match(ln)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def loops(insts):
    """Backward branches: (first address, branch address) of every loop, innermost = no other loop strictly inside."""
    addr = {a for a, _, _ in insts}
    found = []
    for i, (a, m, ops) in enumerate(insts):
        if m.startswith("s_cbranch") or m == "s_branch":
            t = re.search(r"(-?\d+)\s*$", ops)
            if not t:
                continue
            # objdump prints the target as a signed word offset from the next instruction, or as an absolute label
            # address in a comment; recompute from the offset
            nxt = insts[i + 1][0] if i + 1 < len(insts) else a + 4
            off = int(t.group(1))
            if off >= 32768:              # (objdump prints the signed 16-bit word offset as unsigned)
                off -= 65536
            target = nxt + 4 * off
            if target <= a and target in addr:
                found.append((target, a))
    return sorted(set(found))


def count(insts, lo, hi):
    c = collections.Counter()
    detail = collections.Counter()
    for a, m, ops in insts:
        if lo <= a <= hi:
            k = classify(m, ops)
            c[k] += 1
            detail[(k, m)] += 1
    return c, detail


def summarise(name, insts):
    ls = loops(insts)
    inner = [(lo, hi) for lo, hi in ls if not any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in ls)]
    cand = []
    for lo, hi in inner:
        c, det = count(insts, lo, hi)
        nexp = sum(v for (k, m), v in det.items() if m.startswith("v_exp"))
        cand.append((nexp, sum(c.values()), lo, hi, c, det))
    cand.sort(key=lambda t: (-t[0], -t[1]))
    if not cand or cand[0][0] == 0:
        return {"kernel": name, "error": "no inner loop with a v_exp_f32 found", "loops": len(ls)}
    nexp, ninst, lo, hi, c, det = cand[0]
    valu = {k: c[k] for k in ("full", "half", "quarter", "pk", "dpp", "swap", "readlane") if c[k]}
    nvalu = sum(valu.values())
    mix = sum(CLASS_COST[k] * v for k, v in valu.items())
    whole, _ = count(insts, insts[0][0], insts[-1][0])
    return {
        "kernel": name,
        "inner_loop": {"first_address": hex(lo), "branch_address": hex(hi), "instructions": ninst,
                       "v_exp_per_trip": nexp,
                       "by_class": dict(sorted(c.items())),
                       "valu_instructions": nvalu,
                       "valu_by_class": valu,
                       "valu_cycles_by_class": {k: round(CLASS_COST[k] * v, 1) for k, v in valu.items()},
                       "mix_cycles_per_trip": round(mix, 1),
                       "mix_cycles_per_valu_inst": round(mix / max(nvalu, 1), 3),
                       "nominal_cycles_per_valu_inst": NOMINAL_COST,
                       "mnemonics": {k: dict(sorted(((m, v) for (kk, m), v in det.items() if kk == k), key=lambda t: -t[1]))
                                     for k in sorted({kk for kk, _ in det})}},
        "whole_kernel_static": {"instructions": sum(whole.values()), "by_class": dict(sorted(whole.items())), "loops": len(ls),
                                "innermost_loops_with_exp": [{"first": hex(l), "branch": hex(h), "v_exp": n, "instructions": t}
                                                             for n, t, l, h, _, _ in cand if n]},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "gaussiansplattingmlx_amd", "libgsplat_hip.so"))
    ap.add_argument("--out", default="")
    ap.add_argument("--dump", default="", help="directory for the disassembly of the images that hold the kernels")
    args = ap.parse_args()
    res = {}
    for n, img in enumerate(gfx950_images(args.lib)):
        asm = None
        for name, sym in KERNELS.items():
            if sym.encode() not in img:
                continue
            if asm is None:
                asm = disassemble(img)
                if args.dump:
                    os.makedirs(args.dump, exist_ok=True)
                    open(os.path.join(args.dump, f"image{n}.s"), "w").write(asm)
                fn = functions(asm)
            for s, insts in fn.items():
                if s.startswith(sym) and insts:
                    res[name] = summarise(name, insts)
    sys.path.insert(0, ROOT)
    import bench
    out = {"note": "static instruction counts of the shipped gfx950 code object (llvm-objdump -d on the .hip_fatbin images of "
                   "libgsplat_hip.so), inner loop = the innermost loop with the most v_exp_f32 per trip; class costs = measured "
                   "issue intervals, cycles per wave64 instruction per SIMD at 8 waves (profiles/r01c_microbench_issue_rates.txt)",
           "csrc_sha": bench.csrc_sha(), "class_cost_cycles": CLASS_COST, "nominal_cost_cycles": NOMINAL_COST, "kernels": res}
    txt = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(txt + "\n")
    for k, v in res.items():
        il = v.get("inner_loop")
        print(k, "->", (il["by_class"], "mix", il["mix_cycles_per_valu_inst"]) if il else v.get("error"))


if __name__ == "__main__":
    main()
