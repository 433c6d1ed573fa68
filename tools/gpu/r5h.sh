cd $GRAFT_REPO_ROOT; export OUT=r5h; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step t 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "opt_in_persistent_bptt or c2long"
