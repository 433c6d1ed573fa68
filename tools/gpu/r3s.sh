cd $GRAFT_REPO_ROOT; export OUT=r3s; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step stamps 300 python tools/bench_bptt_stamps.py
cat gpurun_out/$OUT/stamps.log; tail -5 gpurun_out/$OUT/stamps.err
