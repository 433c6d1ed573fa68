cd $GRAFT_REPO_ROOT; export OUT=r5o; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step tests 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_train_eval_parity.py -m gpu -x -q -k "beam or c5 or train_harness"
run_step bench 900 python bench.py --steps 10 --no-config3
python - <<'PY'
import json
p=json.loads(open('gpurun_out/r5o/bench.log').read().strip().splitlines()[-1])
print('decode', p['decode']['value'], 'beam', p['beam'])
PY
