# PMC passes over config-3 train steps (bf16 mode, persistent recurrence): matrix-pipe busy share, issue / memory stalls and LDS
# conflicts of gemm_b1_kernel and of the persistent recurrence kernels.  Separate passes, counters only.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_c3b && export S2VT_GEMM_MODE=1
P="python3 tools/prof_path.py c3 2"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_c3b -o p1 -- $P > gpurun_out/pmc_c3b/p1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_c3b -o p2 -- $P > gpurun_out/pmc_c3b/p2.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/pmc_c3b/p*_counter_collection.csv" > gpurun_out/pmc_c3b/summary.txt 2>&1
grep -A16 "gemm_b1\|lstm_seq_fwd_bf16_persist\|lstm_seq_bwd_bf16_persist" gpurun_out/pmc_c3b/summary.txt | head -70
