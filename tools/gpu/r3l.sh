cd $GRAFT_REPO_ROOT; export OUT=r3m; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_argmax 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "argmax"
grep -q "passed" gpurun_out/$OUT/pytest_argmax.log && ! grep -q "failed" gpurun_out/$OUT/pytest_argmax.log || { tail -30 gpurun_out/$OUT/pytest_argmax.log; exit 1; }
run_step pytest_greedy 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or c5"
for dbg in 0 1 2; do
  export S2VT_AX_DBG=$dbg
  run_step stamps$dbg 200 python tools/bench_argmax_x3_stamps.py
  cat gpurun_out/$OUT/stamps$dbg.log
done
unset S2VT_AX_DBG
run_step decode_new 300 python tools/bench_decode.py
tail -2 gpurun_out/$OUT/pytest_argmax.log; tail -2 gpurun_out/$OUT/pytest_greedy.log; head -2 gpurun_out/$OUT/decode_new.log
