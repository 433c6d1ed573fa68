cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; export OUT=r3j; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
for dbg in 0 1 2 3 4 7; do
  export S2VT_AX_DBG=$dbg
  run_step prof_dbg$dbg 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT/d$dbg -o t -- python3 tools/prof_path.py c5 0 --decode
  f=$(find gpurun_out/$OUT/d$dbg -name "*kernel_stats.csv" | head -1); echo "dbg=$dbg"; grep "argmax_x3" $f | cut -c1-150
done
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete
