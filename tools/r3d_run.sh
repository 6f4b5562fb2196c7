mkdir -p gpurun_out/r3d
L=$GRAFT_REPO_ROOT/gaussiansplattingmlx_amd
GSPLAT_LIB=$L/libgsplat_hip_mfma.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_render or bench_workload or config2 or randomized or adversarial or depth_gradient or sh_compressed" > gpurun_out/r3d/mfma_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3d/mfma_tests.log
for i in 1 2; do
python tools/bwd_ab.py c3_300k_800 16 12 > gpurun_out/r3d/ab_default_$i.json 2>/dev/null
GSPLAT_LIB=$L/libgsplat_hip_mfma.so python tools/bwd_ab.py c3_300k_800 16 12 > gpurun_out/r3d/ab_mfma_$i.json 2>/dev/null
done
tail -3 gpurun_out/r3d/mfma_tests.log; cat gpurun_out/r3d/ab_*.json
