cd $GRAFT_REPO_ROOT; export OUT=r5p; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step tests 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "gemm or c3 or bf16 or mid64"
run_step shapes 300 python tools/bench_gemm_shapes.py 256 1
run_step c3 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
export S2VT_B1_MIXED=0
run_step c3_nomix 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
cat gpurun_out/$OUT/shapes.log
python - <<'PY'
import json
for n in ('c3','c3_nomix'):
    try:
        p=json.loads(open('gpurun_out/r5p/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
