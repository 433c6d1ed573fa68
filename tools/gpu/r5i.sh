cd $GRAFT_REPO_ROOT; export OUT=r5i; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step base 300 python bench.py --headline-only --steps 20
export S2VT_PERSIST_X3_BWD=2
run_step parity 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c2 or c4 or mid64"
run_step lanes 300 python bench.py --headline-only --steps 20
run_step lanes128 300 python bench.py --headline-only --steps 20 --batch 128
python - <<'PY'
import json
for n in ('base','lanes','lanes128'):
    try:
        p=json.loads(open('gpurun_out/r5i/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
