# PMC passes over greedy decode calls at B=128 (separate passes, no trace domain beside the counters): where do the waves of
# logits_argmax_kernel spend their cycles?  Summarised by tools/pmc_summary.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_dec
P="python3 tools/prof_decode.py 128"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_dec -o p1 -- $P > gpurun_out/pmc_dec/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_dec -o p2 -- $P > gpurun_out/pmc_dec/p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_dec -o p3 -- $P > gpurun_out/pmc_dec/p3.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_dec -o p4 -- $P > gpurun_out/pmc_dec/p4.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --output-format csv -d gpurun_out/pmc_dec -o p5 -- $P > gpurun_out/pmc_dec/p5.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/pmc_dec/*counter_collection.csv" > gpurun_out/pmc_dec/summary.txt 2>&1
grep -A24 "logits_argmax" gpurun_out/pmc_dec/summary.txt | head -40
