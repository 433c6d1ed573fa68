cd $GRAFT_REPO_ROOT; export OUT=r5f; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step b128_16 300 python bench.py --headline-only --steps 20 --batch 128
run_step b64_16 300 python bench.py --headline-only --steps 20
export S2VT_BWD_ROWS=32
run_step tests 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "bwd or bptt or c2 or c4 or seq"
run_step b128_32 300 python bench.py --headline-only --steps 20 --batch 128
run_step b64_32 300 python bench.py --headline-only --steps 20
python - <<'PY'
import json
for n in ('b128_16','b128_32','b64_16','b64_32'):
    try:
        p=json.loads(open('gpurun_out/r5f/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
