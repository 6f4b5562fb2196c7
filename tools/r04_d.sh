#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
python bench.py --config c3_grown_1m --steps 90 --warmup 10 --no-cpu-baseline > $out/bench_grown.json 2> $out/bench_grown.err && echo grown ok
bash tools/kstats_cmd.sh grown bench.py --config c3_grown_1m --steps 40 --warmup 5 --no-cpu-baseline > $out/kstats_grown.txt 2>&1; grep -v "^$" $out/kstats_grown.txt | head -40
