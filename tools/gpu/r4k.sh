cd $GRAFT_REPO_ROOT; export OUT=r4k; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step stamps 600 python tools/bench_x3_persist_stamps.py
cat gpurun_out/$OUT/stamps.log
