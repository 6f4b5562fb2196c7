#!/bin/bash
# Where the waves of the blend kernels wait: SQ wait / active counters per kernel (GPU box, through gpurun).
# usage: tools/pmc_wait.sh <tag> [bench args]
tag=${1:-wait}; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --steps 6 --warmup 2 $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/a -o p -- $B > $out/a.json 2> $out/a.log || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $out/b -o p -- $B > $out/b.json 2> $out/b.log || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/c -o p -- $B > $out/c.json 2> $out/c.log || exit 1
python3 - "$out" <<'PY'
import sys, glob, csv, json, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "blend_" in k or "loss_fused" in k or "proj_" in k or "wide_scatter" in k:
            a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
res = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, d in res.items():
    print(k[:50]); print("   ", d)
PY
