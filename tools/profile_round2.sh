# Round-2 profiles (run on the GPU box through gpurun): rocprofv3 kernel stats of the bench command for config 2 (default)
# and config 3 (--batch 256 --gemm-mode 1), then PMC traffic passes (separate FETCH_SIZE / WRITE_SIZE / L2 passes, no other
# trace domain beside them).  Summaries are copied to profiles/ by hand afterwards.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_r2 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2 -o c2 -- python3 bench.py --headline-only > gpurun_out/prof_r2/c2_bench_line_under_rocprof.json 2> gpurun_out/prof_r2/c2.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2 -o c3 -- python3 bench.py --batch 256 --gemm-mode 1 --headline-only --steps 10 > gpurun_out/prof_r2/c3_bench_line_under_rocprof.json 2> gpurun_out/prof_r2/c3.err &&
export S2VT_GEMM_MODE=1 &&
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; tag=$1; shift
  mkdir -p gpurun_out/prof_r2/pmc_c3 &&
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof_r2/pmc_c3 -o $tag -- python3 tools/prof_path.py c3 2 > gpurun_out/prof_r2/pmc_c3/$tag.log 2>&1 || exit 1
done
python3 tools/pmc_traffic.py gpurun_out/prof_r2/pmc_c3 "c3 (S2VT_GEMM_MODE=1)" 2 > gpurun_out/prof_r2/traffic_c3.json
unset S2VT_GEMM_MODE
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; tag=$1; shift
  mkdir -p gpurun_out/prof_r2/pmc_c2 &&
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof_r2/pmc_c2 -o $tag -- python3 tools/prof_path.py c2 2 > gpurun_out/prof_r2/pmc_c2/$tag.log 2>&1 || exit 1
done
python3 tools/pmc_traffic.py gpurun_out/prof_r2/pmc_c2 "c2" 2 > gpurun_out/prof_r2/traffic_c2.json
ls gpurun_out/prof_r2 | head -30
