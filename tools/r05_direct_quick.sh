#!/bin/bash
out=gpurun_out/r05_direct; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_binning_large.py tests/test_gpu_parity.py -m gpu -x -q -k "tile_bin or direct or cut_binning or tile_sorts" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $out/pytest.log
[ $rc -ne 0 ] && exit 1
bash tools/kstats_cmd.sh direct1 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -i "direct\|wide\|expand\|scan_block"
bash tools/kstats_cmd.sh direct_c1 bench.py --config c1_10k_400 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -i "direct\|wide\|expand\|scan_block\|rank_sort"
