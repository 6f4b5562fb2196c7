#!/bin/bash
out=gpurun_out/r04l; mkdir -p $out
python -m pytest tests -m gpu -x -q -k "cut or bin or two_word or depth" > $out/pytest_cuts.log 2>&1; echo "pytest cuts rc=$?"; tail -3 $out/pytest_cuts.log
GSPLAT_FUSED_CUT_COMPACTION=1 bash tools/kstats_cmd.sh c3cutsf bench.py --steps 60 --warmup 10 --no-cpu-baseline --cut-min-dropped 1000000 > $out/kstats_cuts.txt 2>&1; grep -E "wide_|expand|compact" $out/kstats_cuts.txt
