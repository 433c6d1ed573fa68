cd $GRAFT_REPO_ROOT; export OUT=r4v; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step bench 900 python bench.py
run_step soak 900 python tools/soak.py 400
tail -5 gpurun_out/$OUT/soak.log
