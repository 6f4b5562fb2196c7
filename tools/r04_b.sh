#!/bin/bash
out=gpurun_out/r04b; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -5 $out/pytest.log
python bench.py --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline > $out/bench_grown.json 2> $out/bench_grown.err && echo grown ok
python bench.py --steps 40 --warmup 10 --dp-single --dp-impl native --dp-exchange allreduce --no-cpu-baseline > $out/bench_dp1_native_allreduce.json 2> $out/bench_dp1_native_allreduce.err && echo dp1 native ok
