# One config-2 optimisation step as a kernel timeline (gpurun command: bash tools/trace_step.sh): rocprofv3 --kernel-trace of
# tools/prof_path.py c2 8, then (a) tools/trace_timeline.py over the second half of the run, (b) every kernel of the LAST step in start
# order with the idle gaps between them -> gpurun_out/r5q/{timeline,gaps}.txt (profiles/round5_c2_step_timeline.txt is gaps.txt).
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5q
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5q -o tr -- python3 tools/prof_path.py c2 8 > gpurun_out/r5q/run.log 2>&1
f=$(find gpurun_out/r5q -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 0.5 0.98 > gpurun_out/r5q/timeline.txt
cat gpurun_out/r5q/timeline.txt
python3 - "$f" <<'PY' > gpurun_out/r5q/gaps.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# last step: from the last adam kernel but one to the last adam kernel
ad = [i for i, e in enumerate(ev) if "adam_flat" in e[2]]
lo, hi = ev[ad[-2]][1], ev[ad[-1]][1]
step = [e for e in ev if e[0] >= lo and e[1] <= hi]
print("last step: %.3f ms, %d kernels" % ((hi - lo) / 1e6, len(step)))
last = lo
cur_end = lo
gaps = []
for s, e, n in step:
    if s > cur_end:
        gaps.append((s - cur_end, n.split("(")[0][-50:], prevn))
    if e > cur_end:
        cur_end = e
        prevn = n.split("(")[0][-50:]
print("idle total %.3f ms in %d gaps" % (sum(g[0] for g in gaps) / 1e6, len(gaps)))
for g in sorted(gaps, reverse=True)[:25]:
    print("  %7.1f us idle before %-52s after %s" % (g[0] / 1e3, g[1], g[2]))
t0 = lo
for s, e, n in step:
    print("%9.1f +%7.1f  %s" % ((s - lo) / 1e3, (e - s) / 1e3, n.split("(")[0].replace("void ", "").replace("s2vt::", "")[:60]))
PY
head -30 gpurun_out/r5q/gaps.txt
find gpurun_out/r5q -name "*kernel_trace.csv" -delete
