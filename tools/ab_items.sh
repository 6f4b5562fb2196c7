for v in "" _own1024 _own0; do
  export GSPLAT_LIB=$GRAFT_REPO_ROOT/gaussiansplattingmlx_amd/libgsplat_hip$v.so
  echo "lib $v"
  python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stages']['loss'], d['stages']['blend_bwd'])"
  python tools/stages_at_n.py 1500 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['N'], d['ms_per_step'], d['stages']['loss'])"
done
