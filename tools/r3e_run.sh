mkdir -p gpurun_out/r3e
L=$GRAFT_REPO_ROOT/gaussiansplattingmlx_amd
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3e/default_$i.json 2>/dev/null
GSPLAT_LIB=$L/libgsplat_hip_nt.so python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3e/nt_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/*.json')):
    j=json.load(open(f)); print(f, j['value'], {k:v['ms'] for k,v in j['stages'].items()})
PY
