cd $GRAFT_REPO_ROOT; export OUT=r5q; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step tests 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "persist or bf16 or c3 or mid64 or rccl"
run_step c3 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
python - <<'PY'
import json
for n in ('c3',):
    try:
        p=json.loads(open('gpurun_out/r5q/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
        print(p['roofline_lstm_step']['frac'], p['roofline_lstm_step_bwd']['frac'])
    except Exception as e: print(n,'ERR',e)
PY
