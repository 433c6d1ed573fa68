cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; export OUT=r3i; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_argmax 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "argmax"
run_step pytest_greedy 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or c5"
run_step decode_new 300 python tools/bench_decode.py
run_step prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT/d -o t -- python3 tools/prof_path.py c5 0 --decode
f=$(find gpurun_out/$OUT/d -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-150
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/$OUT/pytest_argmax.log; tail -2 gpurun_out/$OUT/pytest_greedy.log; head -2 gpurun_out/$OUT/decode_new.log
