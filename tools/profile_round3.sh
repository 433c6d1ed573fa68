# Round-3 profiles (run on the GPU box through gpurun; S2VT_COMMIT = the commit the tree was built from): rocprofv3 kernel
# stats of the bench command for config 2 (default) and config 3 (--batch 256 --gemm-mode 1), a greedy decode, then PMC traffic
# passes (separate FETCH_SIZE / WRITE_SIZE / L2 passes, no other trace domain beside them).  Summaries are copied to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export OUT=prof_r3 && mkdir -p gpurun_out/$OUT && . tools/gpu/run_steps.sh
run_step c2_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o c2 -- python3 bench.py --headline-only
cp gpurun_out/$OUT/c2_stats.log gpurun_out/$OUT/c2_bench_line_under_rocprof.json
run_step c3_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o c3 -- python3 bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
cp gpurun_out/$OUT/c3_stats.log gpurun_out/$OUT/c3_bench_line_under_rocprof.json
run_step dec_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o dec -- python3 tools/prof_path.py c5 0 --decode
export S2VT_GEMM_MODE=1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; tag=$1; shift
  mkdir -p gpurun_out/$OUT/pmc_c3
  run_step pmc_c3_$tag 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$OUT/pmc_c3 -o $tag -- python3 tools/prof_path.py c3 2
done
python3 tools/pmc_traffic.py gpurun_out/$OUT/pmc_c3 "c3 (S2VT_GEMM_MODE=1)" 2 > gpurun_out/$OUT/traffic_c3.json
unset S2VT_GEMM_MODE
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; tag=$1; shift
  mkdir -p gpurun_out/$OUT/pmc_c2
  run_step pmc_c2_$tag 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$OUT/pmc_c2 -o $tag -- python3 tools/prof_path.py c2 2 --decode
done
python3 tools/pmc_traffic.py gpurun_out/$OUT/pmc_c2 "c2 (+ one greedy decode at B=64)" 2 > gpurun_out/$OUT/traffic_c2.json
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete; find gpurun_out/$OUT -name "*counter_collection.csv" -delete
ls gpurun_out/$OUT | head -40
