#!/bin/bash
# VALU counters of the blend kernels on a FIXED scene (tools/bwd_ab.py), per library build: tools/pmc_ab.sh <lib-suffix> ...
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$root/gaussiansplattingmlx_amd/libgsplat_hip${v}.so
  out=$root/gpurun_out/pmc_ab$v; rm -rf $out; mkdir -p $out
  GSPLAT_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/a -o p -- python3 $root/tools/bwd_ab.py c3_300k_800 16 > $out/a.json 2> $out/a.log || { tail -3 $out/a.log; exit 1; }
  python3 - "$out" "$v" <<'PY'
import sys, glob, csv, collections
out, v = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "blend_bwd" in k:
            a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(f"[{v}]", k[:40], {c: round(x[0] / max(x[1], 1) / 1e6, 2) for c, x in d.items()})
PY
done
