"""Last repetition of a repeated workload in a rocprofv3 --kernel-trace CSV: the window from the last launch of kernel
`marker` (substring) that starts a repetition to the end.  Prints the kernel families' time and the sequence.
usage: trace_window.py trace.csv marker [v]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
marks = [i for i, e in enumerate(ev) if sys.argv[2] in e[2]]
# repetitions start at the first marker after a long gap of no markers: take the last quarter of launches
n_rep = 4
per = len(ev) // n_rep
ev = ev[-per:]
lo, hi = ev[0][0], max(e[1] for e in ev)
tot = {}
for s, e, n, q in ev:
    k = n.split("(")[0].replace("void ", "").replace("s2vt::", "")[:40]
    tot.setdefault(k, [0, 0]); tot[k][0] += e - s; tot[k][1] += 1
print("window %.3f ms, %d kernels" % ((hi - lo) / 1e6, len(ev)))
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-42s %7.3f ms %5d launches avg %7.2f us" % (k, t / 1e6, c, t / c / 1e3))
busy, last = 0, lo
pts = sorted([(s, 1) for s, e, n, q in ev] + [(e, -1) for s, e, n, q in ev])
depth = 0
for t, dd in pts:
    if depth > 0: busy += t - last
    depth += dd; last = t
print("  union busy %.3f ms, idle %.3f ms" % (busy / 1e6, (hi - lo - busy) / 1e6))
if len(sys.argv) > 3:
    prev = lo
    for s, e, n, q in ev[:120]:
        print("%9.1f +%6.1f gap %5.1f q%s %s" % ((s - lo) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, q, n.split("(")[0][-40:]))
        prev = max(prev, e)
