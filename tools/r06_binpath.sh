#!/bin/bash
# round 6, item 7: the binning critical path of c3 (kernel trace + FETCH / WRITE passes)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r06_binpath; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- $B --steps 30 --warmup 5 > $out/trace_bench.json 2> $out/trace.log || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- $B --steps 6 --warmup 2 > $out/pmc_$c.json 2> $out/pmc_$c.log || exit 1
done
cd $root
T=$(find $out/trace -name "*kernel_trace.csv" | head -1); F=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
head -2 $T | cut -c1-600
python3 tools/binning_critical_path.py $T --fetch $F --write $W --out $out/r06_binning_critical_path.json --meta "{\"config\": \"c3_300k_800\", \"mode\": \"train\", \"csrc_sha\": \"$(python3 -c 'import bench; print(bench.csrc_sha())')\"}"
