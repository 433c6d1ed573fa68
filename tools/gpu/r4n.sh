cd $GRAFT_REPO_ROOT; export OUT=r4n; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step gpu_tests 1100 python -m pytest tests -m gpu -x -q
run_step c2 300 python bench.py --headline-only --steps 20
run_step c2_128 300 python bench.py --headline-only --steps 20 --batch 128
python - <<'PY'
import json
for n in ('c2','c2_128'):
    try:
        p=json.loads(open('gpurun_out/r4n/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
        print({k:v for k,v in p['roofline_lstm_step'].items() if k not in ('note',)})
    except Exception as e: print(n,'ERR',e)
PY
