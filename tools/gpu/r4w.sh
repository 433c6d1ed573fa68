cd $GRAFT_REPO_ROOT; export OUT=r4w; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step shapes64 300 python tools/bench_gemm_shapes.py 64 3
cat gpurun_out/$OUT/shapes64.log
