cd $GRAFT_REPO_ROOT; export OUT=r5n; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step tests 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_train_eval_parity.py -m gpu -x -q -k "greedy or decode or c2_full or c5 or c1 or split_precision_recurrence or split_precision_two or train_harness"
run_step bench 900 python bench.py --steps 10
python - <<'PY'
import json
p=json.loads(open('gpurun_out/r5n/bench.log').read().strip().splitlines()[-1])
print(p['value'], p['ms_per_step'])
print('decode', {k:v for k,v in p['decode'].items() if not isinstance(v,dict)})
print('dp_b128', p['dp_b128']['ms_per_step'])
PY
