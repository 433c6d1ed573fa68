cd $GRAFT_REPO_ROOT; export OUT=r3e; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_bptt32 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "persistent_bf16"
export S2VT_BPTT_UNITS=16
run_step pytest_bptt16 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "persistent_bf16_bptt"
run_step c3_u16 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
unset S2VT_BPTT_UNITS
run_step c3_u32 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
run_step pytest_c3 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3 or bf16 or mid64"
tail -3 gpurun_out/$OUT/pytest_bptt32.log; tail -3 gpurun_out/$OUT/pytest_bptt16.log; tail -3 gpurun_out/$OUT/pytest_c3.log
python - <<'PY'
import json
for f in ('c3_u16','c3_u32'):
    try:
        p=json.loads(open('gpurun_out/r3e/'+f+'.log').read().strip().splitlines()[-1])
        print(f, p['ms_per_step'], p['kernel_ms_per_step'], p['roofline_lstm_step_bwd']['frac'], p['roofline_lstm_step_bwd']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
