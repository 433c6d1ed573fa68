cd $GRAFT_REPO_ROOT; export OUT=r4d; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_bptt 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "persistent_bf16 or lds_holding"
grep -q "passed" gpurun_out/$OUT/pytest_bptt.log && ! grep -q "failed" gpurun_out/$OUT/pytest_bptt.log || { tail -30 gpurun_out/$OUT/pytest_bptt.log; exit 1; }
export S2VT_BPTT_UNITS=16
run_step pytest_bptt16 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "persistent_bf16_bptt"
unset S2VT_BPTT_UNITS
run_step c3 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
run_step stamps 300 python tools/bench_bptt_stamps.py
run_step pytest_c3 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3 or bf16 or mid64 or rccl"
tail -2 gpurun_out/$OUT/pytest_bptt.log; tail -2 gpurun_out/$OUT/pytest_bptt16.log; tail -2 gpurun_out/$OUT/pytest_c3.log; grep -A1 "workgroup 17" gpurun_out/$OUT/stamps.log
python - <<'PY'
import json
p=json.loads(open('gpurun_out/r4d/c3.log').read().strip().splitlines()[-1])
print(p['value'], p['ms_per_step'], p['kernel_ms_per_step'], p['roofline_lstm_step_bwd']['frac'], p['roofline_lstm_step_bwd']['avg_launch_us'])
PY
