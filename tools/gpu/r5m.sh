cd $GRAFT_REPO_ROOT; export OUT=r5m; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step soak 1100 python tools/soak.py 3000
grep "^configs" gpurun_out/$OUT/soak.log
