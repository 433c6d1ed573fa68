cd $GRAFT_REPO_ROOT; export OUT=r3v; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_graph 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "graph_replay"
tail -12 gpurun_out/$OUT/pytest_graph.log
run_step c2_eager 300 python bench.py --headline-only --steps 20 --graphs 0
run_step c2_graph 300 python bench.py --headline-only --steps 20 --graphs 1
run_step c3_graph 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20 --graphs 1
python - <<'PY'
import json
for f in ('c2_eager','c2_graph','c3_graph'):
    try:
        p=json.loads(open('gpurun_out/r3v/'+f+'.log').read().strip().splitlines()[-1])
        print(f, p['value'], p['ms_per_step'], p['host_enqueue_ms_per_step'], p['host_enqueue_ms_by_phase'], p['graph_mode'])
    except Exception as e: print(f,'ERR',e)
PY
