cd $GRAFT_REPO_ROOT; export OUT=r5d; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step att64 300 python tools/bench_att.py 64
cat gpurun_out/$OUT/att64.log
