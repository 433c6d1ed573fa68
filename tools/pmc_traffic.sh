# HBM traffic per launch from PMC counters (MI355X_MICROARCH.md §HBM): separate passes for FETCH_SIZE and WRITE_SIZE.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc2
P="python3 tools/prof_path.py c2 2"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc2 -o fetch -- $P > gpurun_out/pmc2/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc2 -o write -- $P > gpurun_out/pmc2/write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc2 -o l2 -- $P > gpurun_out/pmc2/l2.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc2 > gpurun_out/pmc2/traffic.json; cat gpurun_out/pmc2/traffic.json
