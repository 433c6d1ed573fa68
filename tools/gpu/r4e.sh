cd $GRAFT_REPO_ROOT; export OUT=r4e; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_persist 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "persistent_bf16 or lds_holding"
grep -q "passed" gpurun_out/$OUT/pytest_persist.log && ! grep -q "failed" gpurun_out/$OUT/pytest_persist.log || { tail -30 gpurun_out/$OUT/pytest_persist.log; exit 1; }
run_step c3_xcd 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
export S2VT_PERSIST_XCD=0
run_step c3_plain 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
unset S2VT_PERSIST_XCD
run_step pytest_c3 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3 or bf16 or mid64 or rccl"
tail -2 gpurun_out/$OUT/pytest_persist.log; tail -2 gpurun_out/$OUT/pytest_c3.log
python - <<'PY'
import json
for f in ('c3_xcd','c3_plain'):
    p=json.loads(open('gpurun_out/r4e/'+f+'.log').read().strip().splitlines()[-1])
    print(f, p['value'], p['ms_per_step'], p['kernel_ms_per_step'], p['roofline_lstm_step']['frac'], p['roofline_lstm_step_bwd']['frac'])
PY
