cd $GRAFT_REPO_ROOT; export OUT=r3z; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_idx 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index_error"
tail -2 gpurun_out/$OUT/pytest_idx.log
run_step soak 900 python tools/soak.py 400
grep "steps, step time" gpurun_out/$OUT/soak.log
