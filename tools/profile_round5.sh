# Round-5 profiles (run on the GPU box through gpurun; S2VT_COMMIT = the commit the tree was built from).
#   usage (gpurun command): S2VT_COMMIT=<sha> bash tools/profile_round5.sh [stats] [pmc] [traffic]
# stats:   rocprofv3 --kernel-trace --stats of the bench command for config 2 / config 3 and of one greedy decode
# pmc:     SQ / GRBM counter passes (matrix-pipe busy, wait / issue-stall shares, LDS conflicts, instruction mix) over
#          tools/prof_path.py c2, c3, c5 --decode (B = 128) and tools/prof_beam_device.py: counters only, program directly after --
# traffic: FETCH_SIZE / WRITE_SIZE / L2 passes (tools/pmc_traffic.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export OUT=${OUT:-prof_r5} && mkdir -p gpurun_out/$OUT && . tools/gpu/run_steps.sh
WHAT="${*:-stats pmc traffic}"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
case " $WHAT " in *" stats "*)
  run_step c2_stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o c2 -- python3 bench.py --headline-only --steps 25
  cp gpurun_out/$OUT/c2_stats.log gpurun_out/$OUT/c2_bench_line_under_rocprof.json
  run_step c3_stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o c3 -- python3 bench.py --batch 256 --gemm-mode 1 --headline-only --steps 25
  cp gpurun_out/$OUT/c3_stats.log gpurun_out/$OUT/c3_bench_line_under_rocprof.json
  run_step dec_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o dec -- python3 tools/prof_path.py c5 0 --decode
;; esac
case " $WHAT " in *" pmc "*)
  for cfg in c2 c3 dec beam; do
    mkdir -p gpurun_out/$OUT/sq_$cfg
    case $cfg in
      c2) unset S2VT_GEMM_MODE; P="python3 tools/prof_path.py c2 2";;
      c3) export S2VT_GEMM_MODE=1; P="python3 tools/prof_path.py c3 2";;
      dec) unset S2VT_GEMM_MODE; P="python3 tools/prof_path.py c5 0 --decode";;
      beam) unset S2VT_GEMM_MODE; P="python3 tools/prof_beam_device.py";;
    esac
    run_step sq1_$cfg 300 rocprofv3 --pmc $SQ1 --output-format csv -d gpurun_out/$OUT/sq_$cfg -o p1 -- $P
    run_step sq2_$cfg 300 rocprofv3 --pmc $SQ2 --output-format csv -d gpurun_out/$OUT/sq_$cfg -o p2 -- $P
    python3 tools/pmc_busy.py gpurun_out/$OUT/sq_$cfg "$P (S2VT_GEMM_MODE=${S2VT_GEMM_MODE:-default})" > gpurun_out/$OUT/pmc_$cfg.json 2> gpurun_out/$OUT/pmc_$cfg.txt
  done
  unset S2VT_GEMM_MODE
;; esac
case " $WHAT " in *" traffic "*)
  # whole optimisation steps (tools/prof_path.py: train_step with FlatAdam), three passes each; the decode's kernels in a run of their own
  for cfg in c3 c2 dec; do
    case $cfg in
      c3) export S2VT_GEMM_MODE=1; P="python3 tools/prof_path.py c3 3"; N=3; W="c3 (S2VT_GEMM_MODE=1), whole train steps";;
      c2) unset S2VT_GEMM_MODE; P="python3 tools/prof_path.py c2 3"; N=3; W="c2, whole train steps";;
      dec) unset S2VT_GEMM_MODE; P="python3 tools/prof_path.py c5 0 --decode"; N=1; W="c5: one greedy decode at B=128 (cold)";;
    esac
    mkdir -p gpurun_out/$OUT/pmc_$cfg
    for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
      set -- $pass; tag=$1; shift
      run_step pmc_${cfg}_$tag 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$OUT/pmc_$cfg -o $tag -- $P
    done
    python3 tools/pmc_traffic.py gpurun_out/$OUT/pmc_$cfg "$W" $N > gpurun_out/$OUT/traffic_$cfg.json
  done
  unset S2VT_GEMM_MODE
;; esac
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete; find gpurun_out/$OUT -name "*counter_collection.csv" -delete
ls gpurun_out/$OUT | head -60
