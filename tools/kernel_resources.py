#!/usr/bin/env python3
"""Static resources of every kernel in the SHIPPED code object: LDS bytes, VGPRs, AGPRs, SGPRs, scratch, max flat workgroup size.

    python tools/kernel_resources.py [--lib gaussiansplattingmlx_amd/libgsplat_hip.so] [--out profiles/r05_kernel_resources.json]

Read from the AMDGPU metadata note of the gfx950 images in the library's .hip_fatbin section (the images are taken out by
tools/isa_mix.py's bundle reader; llvm-readelf --notes prints the msgpack metadata as YAML).  DESIGN.md section 4's table
quotes these numbers (dynamic LDS, where a kernel asks for it at launch, is not in the note: the table says so).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import isa_mix  # noqa: E402

FIELDS = {".group_segment_fixed_size": "lds_bytes", ".vgpr_count": "vgprs", ".agpr_count": "agprs", ".sgpr_count": "sgprs",
          ".private_segment_fixed_size": "scratch_bytes", ".max_flat_workgroup_size": "max_workgroup", ".vgpr_spill_count": "vgpr_spills"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()


def kernels_of(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        txt = subprocess.run([f"{isa_mix.LLVM}/llvm-readelf", "--notes", f.name], capture_output=True, text=True, check=True).stdout
    res, cur = [], None
    for ln in txt.splitlines():
        m = re.match(r"^\s*-?\s*(\.[a-z_]+):\s*(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == ".agpr_count" and (cur is None or "agprs" in cur):      # first field of a kernel entry (alphabetical order)
            cur = {}
            res.append(cur)
        if cur is None:
            continue
        if k in FIELDS:
            cur[FIELDS[k]] = int(v)
        elif k == ".name":
            cur["symbol"] = v
    return [r for r in res if "symbol" in r]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "gaussiansplattingmlx_amd", "libgsplat_hip.so"))
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rows = []
    for img in isa_mix.gfx950_images(args.lib):
        rows += kernels_of(img)
    for r, d in zip(rows, demangle([r["symbol"] for r in rows])):
        r["kernel"] = re.sub(r"\((?!anonymous namespace\)).*$", "", d).replace("void ", "").replace("(anonymous namespace)::", "")
    rows.sort(key=lambda r: r["kernel"])
    import bench
    out = {"note": "AMDGPU metadata of the gfx950 images in libgsplat_hip.so (llvm-readelf --notes); lds_bytes is the static "
                   "part only", "csrc_sha": bench.csrc_sha(), "kernels": rows}
    if args.out:
        open(args.out, "w").write(json.dumps(out, indent=1) + "\n")
    for r in rows:
        print(f"{r['kernel']:<60} lds {r.get('lds_bytes', 0):>6}  vgpr {r.get('vgprs', 0):>3} agpr {r.get('agprs', 0):>3} sgpr {r.get('sgprs', 0):>3} "
              f"scratch {r.get('scratch_bytes', 0):>4} wg {r.get('max_workgroup', 0):>4}")


if __name__ == "__main__":
    main()
