cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; export OUT=r3g; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step prof_decode 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT/dec -o t -- python3 tools/prof_path.py c5 0 --decode
f=$(find gpurun_out/$OUT/dec -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
find gpurun_out/$OUT -name "*kernel_trace.csv" -size +20M -delete
