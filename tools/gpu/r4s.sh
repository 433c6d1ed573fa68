cd $GRAFT_REPO_ROOT; export OUT=r4s; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step kern 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "split_precision_bptt"
run_step stamps 600 python tools/bench_x3_bptt_stamps.py
export S2VT_PERSIST_X3_BWD=1
run_step c2 300 python bench.py --headline-only --steps 20
cat gpurun_out/$OUT/stamps.log
python - <<'PY'
import json
for n in ('c2',):
    try:
        p=json.loads(open('gpurun_out/r4s/%s.log'%n).read().strip().splitlines()[-1])
        print(n, p['value'], p['ms_per_step'], p['final_loss'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
