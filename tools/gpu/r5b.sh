cd $GRAFT_REPO_ROOT; export OUT=r5b; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step att 600 python -m pytest tests/test_gpu_att_baseline.py -m gpu -x -q
