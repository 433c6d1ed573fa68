cd $GRAFT_REPO_ROOT; export OUT=r4g; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
for blk in 32 27 20 16 40; do
  export S2VT_PIPE_BLOCK=$blk
  run_step c3_b$blk 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
  run_step c2_b$blk 300 python bench.py --headline-only --steps 20
done
python - <<'PY'
import json
for blk in (32,27,20,16,40):
    for c in ('c3','c2'):
        try:
            p=json.loads(open('gpurun_out/r4g/%s_b%d.log'%(c,blk)).read().strip().splitlines()[-1])
            print(c, blk, p['value'], p['ms_per_step'], p['kernel_ms_per_step'])
        except Exception as e: print(c,blk,'ERR',e)
PY
