cd $GRAFT_REPO_ROOT; export OUT=r4p; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step kern 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "split_precision_bptt"
