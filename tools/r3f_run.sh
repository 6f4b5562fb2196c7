mkdir -p gpurun_out/r3f; rm -f gpurun_out/r3f/*
L=$GRAFT_REPO_ROOT/gaussiansplattingmlx_amd
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loss or fused_render or train_steps or config1 or depth_cuts" > gpurun_out/r3f/tests_default.log 2>&1; echo "rc=$?" >> gpurun_out/r3f/tests_default.log
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3f/b_default_$i.json 2>/dev/null
for v in l32x32x256; do
GSPLAT_LIB=$L/libgsplat_hip_$v.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3f/b_${v}_$i.json 2>/dev/null
done; done
tail -n 2 gpurun_out/r3f/tests_default.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f/b_*.json')):
    try:
        j=json.load(open(f)); print(f, j['value'], j['stages']['loss']['ms'], j['loss'][:2])
    except Exception as e: print(f, 'ERR', e)
PY
