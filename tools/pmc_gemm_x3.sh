cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc3
P="python3 tools/prof_gemm_x3.py"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc3 -o p1 -- $P > gpurun_out/pmc3/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d gpurun_out/pmc3 -o p2 -- $P > gpurun_out/pmc3/p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o p3 -- $P > gpurun_out/pmc3/p3.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/pmc3/p*_counter_collection.csv" | grep -A30 "gemm_x3"
