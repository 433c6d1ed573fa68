cd $GRAFT_REPO_ROOT; export OUT=r3y; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_cache 600 python -m pytest tests/test_gpu_parity.py tests/test_train_eval_parity.py -m gpu -x -q -k "decode_cache or golden or c5 or eval"
tail -3 gpurun_out/$OUT/pytest_cache.log
run_step decode 300 python tools/bench_decode.py
head -1 gpurun_out/$OUT/decode.log
run_step pytest_all 1000 python -m pytest tests -m gpu -x -q
tail -3 gpurun_out/$OUT/pytest_all.log
run_step bench 600 python bench.py
python - <<'PY'
import json
p=json.loads(open('gpurun_out/r3y/bench.log').read().strip().splitlines()[-1])
print(p['value'],p['ms_per_step']); dd=p['decode']; print({k:dd[k] for k in ('value','ms_per_call','cold_ms_per_call','cold_captions_per_s')}, dd['roofline_logits_argmax']['avg_launch_us'])
c=p['config3']; print('c3',c['value'],c['ms_per_step']); print(p['dp_b128']['ms_per_step'], p['beam']['value'])
PY
