cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export OUT=r4z && mkdir -p gpurun_out/$OUT && . tools/gpu/run_steps.sh
run_step trace 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$OUT -o c2 -- python3 tools/prof_path.py c2 3
python3 tools/trace_iter.py gpurun_out/$OUT/c2_kernel_trace.csv dump > gpurun_out/$OUT/iter.txt 2>&1
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete
grep -c . gpurun_out/$OUT/iter.txt
