cd $GRAFT_REPO_ROOT; export OUT=r3a; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step pytest 600 python -m pytest tests -m gpu -x -q
run_step host_c2 120 python tools/host_cost.py
run_step host_c3 120 python tools/host_cost.py --batch 256 --gemm-mode 1
run_step c3_par 200 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
export S2VT_SERIAL_TAIL=1
run_step c3_ser 200 python bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
run_step c2_ser 200 python bench.py --headline-only --steps 20
unset S2VT_SERIAL_TAIL
run_step bench 400 python bench.py
tail -3 gpurun_out/$OUT/pytest.log; cat gpurun_out/$OUT/host_c2.log gpurun_out/$OUT/host_c3.log
