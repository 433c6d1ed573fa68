cd $GRAFT_REPO_ROOT; export OUT=r3k; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
for dbg in 0 1 2 3; do
  export S2VT_AX_DBG=$dbg
  run_step stamps$dbg 200 python tools/bench_argmax_x3_stamps.py
  cat gpurun_out/$OUT/stamps$dbg.log
done
