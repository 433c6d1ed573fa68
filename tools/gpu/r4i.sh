cd $GRAFT_REPO_ROOT; export OUT=r4i; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step base 300 python bench.py --headline-only --steps 20
export S2VT_PERSIST_F32_BWD=1
run_step bwd32 300 python bench.py --headline-only --steps 20
export S2VT_PIPE_BLOCK=27
run_step bwd27 300 python bench.py --headline-only --steps 20
export S2VT_PERSIST_F32_FWD=1
run_step both27 300 python bench.py --headline-only --steps 20
export S2VT_PERSIST_F32_BWD=0
run_step fwd27 300 python bench.py --headline-only --steps 20
python - <<'PY'
import json
for n in ('base','bwd32','bwd27','both27','fwd27'):
    try:
        p=json.loads(open('gpurun_out/r4i/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
