cd $GRAFT_REPO_ROOT; export OUT=r3o; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_all 1000 python -m pytest tests -m gpu -x -q
run_step bench 600 python bench.py
tail -3 gpurun_out/$OUT/pytest_all.log
python - <<'PY'
import json
p=json.loads(open('gpurun_out/r3o/bench.log').read().strip().splitlines()[-1])
print(p['value'],p['ms_per_step'],p['host_enqueue_ms_per_step'])
print('decode',p['decode']['value'],p['decode']['ms_per_call'],p['decode']['roofline_logits_argmax']['avg_launch_us'],p['decode']['roofline_logits_argmax']['frac'])
print('beam',p['beam']['value']); c=p['config3']; print('c3',c['value'],c['ms_per_step'],c['roofline_gemm']['frac'],c['roofline_lstm_step']['frac'],c['roofline_lstm_step_bwd']['frac'])
print(p['dp_b128']['ms_per_step'], p['roofline']['kernel'], p['roofline']['frac'], p['roofline_min'])
PY
