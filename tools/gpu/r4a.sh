cd $GRAFT_REPO_ROOT; export OUT=r4b; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest_gemm 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "split_precision or gemm"
tail -3 gpurun_out/$OUT/pytest_gemm.log
grep -q "passed" gpurun_out/$OUT/pytest_gemm.log && ! grep -q "failed" gpurun_out/$OUT/pytest_gemm.log || { grep -n "^E " gpurun_out/$OUT/pytest_gemm.log | head; exit 1; }
run_step shapes_wide 300 python tools/bench_gemm_shapes.py 64 3
export S2VT_X3_WIDE=0
run_step shapes_old 300 python tools/bench_gemm_shapes.py 64 3
unset S2VT_X3_WIDE
paste -d'|' <(cut -c1-75 gpurun_out/$OUT/shapes_wide.log) <(cut -c52-75 gpurun_out/$OUT/shapes_old.log)
