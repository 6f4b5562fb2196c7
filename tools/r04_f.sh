#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
python -m pytest tests/test_gpu_binning_large.py -x -q > $out/pytest_bin.log 2>&1; echo "pytest bin rc=$?"; tail -4 $out/pytest_bin.log
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_binning_large.py > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
python bench.py --config c3_grown_1m --steps 90 --warmup 10 --no-cpu-baseline > $out/bench_grown.json 2> $out/bench_grown.err && echo grown ok
python bench.py --config c5_garden_2m --steps 120 --warmup 10 --views 8 --no-cpu-baseline > $out/bench_c5.json 2> $out/bench_c5.err && echo c5 ok
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err && echo c3 ok
python bench.py --steps 40 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline > $out/bench_dp1_native.json 2> $out/bench_dp1_native.err && echo dp1 native ok
