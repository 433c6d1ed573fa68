cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; export OUT=r3c; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_c3long 600 python -m pytest tests -m gpu -x -q -k "c3long" -s
export S2VT_PIPE_BLOCK=0
run_step trace_c2iso 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$OUT/c2iso -o t -- python3 tools/prof_path.py c2 3
unset S2VT_PIPE_BLOCK
export S2VT_GEMM_MODE=1
run_step trace_c3 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$OUT/c3 -o t -- python3 tools/prof_path.py c3 3
unset S2VT_GEMM_MODE
run_step trace_c2 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$OUT/c2 -o t -- python3 tools/prof_path.py c2 3
for t in c2iso c3 c2; do f=$(find gpurun_out/$OUT/$t -name "*kernel_trace.csv" | head -1); python3 tools/trace_iter.py $f dump > gpurun_out/$OUT/$t.iter.txt 2>&1; done
find gpurun_out/$OUT -name "*kernel_trace.csv" -size +20M -delete
tail -5 gpurun_out/$OUT/pytest_c3long.log; head -30 gpurun_out/$OUT/c2iso.iter.txt
