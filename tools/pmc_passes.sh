cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc
P="python3 tools/prof_path.py c2 1"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc -o p1 -- $P > gpurun_out/pmc/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc -o p2 -- $P > gpurun_out/pmc/p2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc -o p3 -- $P > gpurun_out/pmc/p3.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc -o p4 -- $P > gpurun_out/pmc/p4.log 2>&1
ls gpurun_out/pmc; tail -3 gpurun_out/pmc/p4.log
