cd $GRAFT_REPO_ROOT; export OUT=r5g; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step gpu_tests 1100 python -m pytest tests -m gpu -x -q
run_step b256 300 python bench.py --headline-only --steps 10 --batch 256
run_step b128 300 python bench.py --headline-only --steps 20 --batch 128
python - <<'PY'
import json
for n in ('b128','b256'):
    try:
        p=json.loads(open('gpurun_out/r5g/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
