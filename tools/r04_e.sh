#!/bin/bash
out=gpurun_out/r04e; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in single native_sh; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/$m -o t -- python3 $GRAFT_REPO_ROOT/tools/dp_host_overhead.py $m > $GRAFT_REPO_ROOT/$out/$m.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/$m -name "*kernel_trace.csv" | head -1)
  echo "== $m"; grep "ms/step" $GRAFT_REPO_ROOT/$out/$m.txt
  python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $f 3000 | tee $GRAFT_REPO_ROOT/$out/gaps_$m.txt
  rm -rf $GRAFT_REPO_ROOT/$out/$m
done
