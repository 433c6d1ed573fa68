cd $GRAFT_REPO_ROOT; export OUT=r3f; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_argmax 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "argmax"
run_step pytest_greedy 900 python -m pytest tests/test_gpu_parity.py tests/test_train_eval_parity.py -m gpu -x -q -k "golden or oracle or c5 or greedy or eval"
run_step decode_new 300 python tools/bench_decode.py
export S2VT_ARGMAX_F32=1
run_step decode_old 300 python tools/bench_decode.py
unset S2VT_ARGMAX_F32
tail -4 gpurun_out/$OUT/pytest_argmax.log; tail -4 gpurun_out/$OUT/pytest_greedy.log; tail -8 gpurun_out/$OUT/decode_new.log; tail -8 gpurun_out/$OUT/decode_old.log
