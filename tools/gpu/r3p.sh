cd $GRAFT_REPO_ROOT; export OUT=r3q; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_fused 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_criterion"
run_step pytest_all 1000 python -m pytest tests -m gpu -x -q
run_step c2 300 python bench.py --headline-only --steps 20
run_step c3 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
tail -5 gpurun_out/$OUT/pytest_fused.log; tail -3 gpurun_out/$OUT/pytest_all.log
python - <<'PY'
import json
for f in ('c2','c3'):
    p=json.loads(open('gpurun_out/r3q/'+f+'.log').read().strip().splitlines()[-1])
    print(f, p['value'], p['ms_per_step'], p['kernel_ms_per_step'])
PY
