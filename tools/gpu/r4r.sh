cd $GRAFT_REPO_ROOT; export OUT=r4r; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
export S2VT_PERSIST_X3_BWD=1
run_step c2 300 python bench.py --headline-only --steps 20
python - <<'PY'
import json
for n in ('c2',):
    try:
        p=json.loads(open('gpurun_out/r4r/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
