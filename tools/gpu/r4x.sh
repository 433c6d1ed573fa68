cd $GRAFT_REPO_ROOT; export OUT=r4x; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step tests 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "split or plane or gemm or fused or c2 or c3 or mid64 or dropout"
run_step c2 300 python bench.py --headline-only --steps 20
run_step c3 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
python - <<'PY'
import json
for n in ('c2','c3'):
    try:
        p=json.loads(open('gpurun_out/r4x/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
