cd $GRAFT_REPO_ROOT; export OUT=r3b; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_new 900 python -m pytest tests -m gpu -x -q -k "c5_full_batch or beam_step_at_config5 or lds_holding or single_rank_rccl" -s
run_step pytest_all 900 python -m pytest tests -m gpu -x -q
run_step bench 500 python bench.py
tail -15 gpurun_out/$OUT/pytest_new.log; tail -3 gpurun_out/$OUT/pytest_all.log
