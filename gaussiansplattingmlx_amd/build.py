"""Builds libgsplat_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgsplat_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# per-file extra flags: projection/binning do integer-deciding float math -> no FMA contraction
SOURCES = {
    "api.hip": [],
    "projection.hip": ["-ffp-contract=off"],
    "binning.hip": ["-ffp-contract=off"],
    "blend.hip": [],
    "blend_v2.hip": ["-ffp-contract=off"],
    "ssim.hip": [],
    "optim.hip": [],
    "densify.hip": ["-ffp-contract=off"],
    "ply.hip": [],
    "knn.hip": ["-ffp-contract=off"],
    "dp.hip": [],
}
# -fno-slp-vectorize: on gfx950 v_pk_*_f32 issues at half the rate of the scalar forms, so the SLP vectoriser's packed
# math buys nothing and pays for its operand shuffles in v_mov (measured: blend backward 0.65 -> 0.55 ms without it)
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall", "-fno-slp-vectorize",
          "-Wno-unused-function", "-DGSPLAT_BUILD"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False, variant: str = "", defines=()) -> str:
    """variant / defines: an experiment build (extra -D flags) into libgsplat_hip_<variant>.so with its own object
    directory; load it with GSPLAT_LIB=<path> (_lib.py).  The product build has neither."""
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "gsplat.h"))
    objdir = os.path.join(HERE, "_obj" + ("_" + variant if variant else ""))
    lib = LIB if not variant else os.path.join(HERE, f"libgsplat_hip_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _newer(o, [s] + headers + [os.path.abspath(__file__)]):     # flags live in this file
            jobs.append([HIPCC, *COMMON, *extra, *defines, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib):
        run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs, "-ldl"])
    return lib


if __name__ == "__main__":
    # python -m gaussiansplattingmlx_amd.build [--force] [--variant NAME -DFOO ...]
    argv = sys.argv[1:]
    variant = argv[argv.index("--variant") + 1] if "--variant" in argv else ""
    print(build(force="--force" in argv, verbose=True, variant=variant, defines=[a for a in argv if a.startswith("-D")]))
