cd $GRAFT_REPO_ROOT; export OUT=r5k; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step gpu_tests 1100 python -m pytest tests -m gpu -x -q
