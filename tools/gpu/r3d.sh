cd $GRAFT_REPO_ROOT; export OUT=r3d; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step blaslt 300 python tools/bench_blaslt_shapes.py 256
run_step b1 300 python tools/bench_gemm_shapes.py 256 1
cat gpurun_out/$OUT/blaslt.log
