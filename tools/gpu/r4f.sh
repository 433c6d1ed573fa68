cd $GRAFT_REPO_ROOT; export OUT=r4f; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step stamps_sc1 300 python tools/bench_bptt_stamps.py
export PLAIN_STORES=1
run_step stamps_plainst 300 python tools/bench_bptt_stamps.py
export PLAIN_LOADS=1
run_step stamps_plainboth 300 python tools/bench_bptt_stamps.py
for f in stamps_sc1 stamps_plainst stamps_plainboth; do echo "== $f"; grep -v "^build" gpurun_out/$OUT/$f.log | head -4; done
