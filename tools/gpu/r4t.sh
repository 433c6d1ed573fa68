cd $GRAFT_REPO_ROOT; export OUT=r4t; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step b128_step 300 python bench.py --headline-only --steps 20 --batch 128
run_step b256_step 300 python bench.py --headline-only --steps 10 --batch 256
export S2VT_PERSIST_X3_BWD=1
run_step b128_x3 300 python bench.py --headline-only --steps 20 --batch 128
run_step b256_x3 300 python bench.py --headline-only --steps 10 --batch 256
python - <<'PY'
import json
for n in ('b128_step','b128_x3','b256_step','b256_x3'):
    try:
        p=json.loads(open('gpurun_out/r4t/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
