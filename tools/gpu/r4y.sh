cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export OUT=r4y && mkdir -p gpurun_out/$OUT && . tools/gpu/run_steps.sh
run_step c2_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o c2 -- python3 bench.py --headline-only
run_step c3_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -o c3 -- python3 bench.py --batch 256 --gemm-mode 1 --headline-only --steps 20
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv
for n in ('c2','c3'):
    rows=list(csv.DictReader(open('gpurun_out/r4y/%s_kernel_stats.csv'%n)))
    for r in rows[:12]:
        if 'split' in r['Name'] or 'reduce' in r['Name']:
            print(n, "%-70s calls %6s tot %8.3f ms avg %8.2f us"%(r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
