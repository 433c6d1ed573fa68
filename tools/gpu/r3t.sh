cd $GRAFT_REPO_ROOT; export OUT=r3t; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step stamps_sc1 300 python tools/bench_bptt_stamps.py
export PLAIN_LOADS=1
run_step stamps_plain 300 python tools/bench_bptt_stamps.py
cat gpurun_out/$OUT/stamps_sc1.log gpurun_out/$OUT/stamps_plain.log
