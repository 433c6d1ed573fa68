cd $GRAFT_REPO_ROOT; export OUT=r3x; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_greedy 900 python -m pytest tests/test_gpu_parity.py tests/test_train_eval_parity.py tests/test_gpu_kernels.py -m gpu -x -q -k "golden or c5 or eval or oracle or step or argmax"
tail -3 gpurun_out/$OUT/pytest_greedy.log
run_step decode_tab 300 python tools/bench_decode.py
export S2VT_DECODE_TABLE=0
run_step decode_notab 300 python tools/bench_decode.py
unset S2VT_DECODE_TABLE
run_step decode64 300 python tools/bench_decode.py 64
head -1 gpurun_out/$OUT/decode_tab.log; head -1 gpurun_out/$OUT/decode_notab.log; head -1 gpurun_out/$OUT/decode64.log
