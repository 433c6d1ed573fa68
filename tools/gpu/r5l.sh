cd $GRAFT_REPO_ROOT; export OUT=r5l; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step kern 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "split_precision_recurrence or split_precision_two or c4 or c2_full or mid64long"
export STAMPS=0
run_step t 300 python tools/bench_x3_persist_stamps.py
unset STAMPS
run_step b128 300 python bench.py --headline-only --steps 20 --batch 128
run_step b64 300 python bench.py --headline-only --steps 20
run_step b256 300 python bench.py --headline-only --steps 10 --batch 256
cat gpurun_out/$OUT/t.log
python - <<'PY'
import json
for n in ('b64','b128','b256'):
    try:
        p=json.loads(open('gpurun_out/r5l/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
