cd $GRAFT_REPO_ROOT; export OUT=r5j; mkdir -p gpurun_out/$OUT; . tools/gpu/run_steps.sh
run_step smoke 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')"
run_step bench_default 900 python bench.py
tail -2 gpurun_out/$OUT/smoke.log
