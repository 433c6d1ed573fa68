cd $GRAFT_REPO_ROOT; export OUT=r4h; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step parity 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -x -q -k "c3 or bf16 or mid64 or rccl or persist or graph or fused"
run_step c3 300 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
run_step c2 300 python bench.py --headline-only --steps 20
