#!/bin/bash
# idle gaps of the data-parallel step on a 1-rank group: tools/dp_trace.sh <mode ...>   (GPU box, through gpurun)
out=$GRAFT_REPO_ROOT/gpurun_out/dp_trace; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $out/$m -o t -- python3 $GRAFT_REPO_ROOT/tools/dp_host_overhead.py $m > $out/$m.txt 2>&1
  f=$(find $out/$m -name "*kernel_trace.csv" | head -1)
  echo "== $m"; grep "ms/step" $out/$m.txt
  python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $f 3000 | tee $out/gaps_$m.txt | head -14
  rm -rf $out/$m
done
