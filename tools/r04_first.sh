#!/bin/bash
# round 4, first GPU call: the whole GPU suite, the bench at 100 / 8 views, the grown scene, the exchange rehearsals
out=gpurun_out/r04a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -5 $out/pytest.log
python bench.py --steps 100 --warmup 10 > $out/bench_c3_v100.json 2> $out/bench_c3_v100.err && echo c3v100 ok
python bench.py --steps 100 --warmup 10 --views 8 --no-cpu-baseline > $out/bench_c3_v8.json 2> $out/bench_c3_v8.err && echo c3v8 ok
python bench.py --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline > $out/bench_grown.json 2> $out/bench_grown.err && echo grown ok
python bench.py --steps 40 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline > $out/bench_dp1_torch.json 2> $out/bench_dp1_torch.err && echo dp1 torch ok
python bench.py --steps 40 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline > $out/bench_dp1_native.json 2> $out/bench_dp1_native.err && echo dp1 native ok
GSPLAT_BENCH_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_dp2_gloo.json 2> $out/bench_dp2_gloo.err && echo dp2 gloo ok
timeout -k 10 120 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_dp2_refused.out 2> $out/bench_dp2_refused.err; echo "refused rc=$?" >> $out/bench_dp2_refused.err
GSPLAT_BENCH_DEVICE=0 timeout -k 10 180 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_dp2_nccl_onecard.out 2> $out/bench_dp2_nccl_onecard.err; echo "nccl one card rc=$?" >> $out/bench_dp2_nccl_onecard.err
tail -3 $out/*.err | tail -60
