cd $GRAFT_REPO_ROOT; export OUT=r4u; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
for blk in 32 16 20 24 40; do
  export S2VT_PIPE_BLOCK=$blk
  run_step c2_b$blk 300 python bench.py --headline-only --steps 20
done
python - <<'PY'
import json
for blk in (32,16,20,24,40):
    try:
        p=json.loads(open('gpurun_out/r4u/c2_b%d.log'%blk).read().strip().splitlines()[-1])
        print(blk, p['value'], p['ms_per_step'], p['kernel_ms_per_step'], p.get('kernel_busy_ms_per_step'))
    except Exception as e: print(blk,'ERR',e)
PY
