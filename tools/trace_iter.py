"""One training iteration out of a rocprofv3 --kernel-trace CSV (tools/prof_path.py run): the window between the last two
ce_row launches = one full forward+backward period.  Prints which kernel families own the time line."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))


def fam(n):
    return ("gemm" if ("gemm_f32" in n or "gemm_x3" in n or "gemm_bf16" in n or "gemm_b1" in n) else "split" if "split_" in n else
            "splitk" if "splitk" in n else "step_fwd" if ("lstm_step_fwd" in n or "lstm_seq_fwd" in n) else "step_bwd" if ("lstm_step_bwd" in n or "lstm_seq_bwd" in n) else
            "argmax" if "logits_argmax" in n else "ce" if "ce_" in n else "adam" if "multi_tensor" in n else "other")


def grid(r):
    try:
        return "%sx%s/%s" % (int(r.get("Grid_Size_X", 0)) // max(int(r.get("Workgroup_Size_X", 1)), 1),
                             int(r.get("Grid_Size_Y", 0)) // max(int(r.get("Workgroup_Size_Y", 1)), 1), r.get("Workgroup_Size_X", "?"))
    except Exception:
        return "?"


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"]), r["Kernel_Name"] + " [" + grid(r) + "]", r.get("Queue_Id", "?")) for r in rows)
ce = [e[0] for e in ev if "ce_row" in e[3]]
lo, hi = ce[-2], ce[-1]
ev = [e for e in ev if e[0] >= lo and e[0] < hi]
pts = []
for s, e, f, _, _ in ev:
    pts.append((s, 1, f)); pts.append((min(e, hi), -1, f))
pts.sort()
active, last, famtime, depth_hist = {}, lo, {}, {}
for t, d, f in pts:
    dt = t - last
    if dt > 0:
        key = "+".join(sorted(k for k, v in active.items() if v > 0)) or "idle"
        famtime[key] = famtime.get(key, 0) + dt
    active[f] = active.get(f, 0) + d
    last = t
wall = hi - lo
print("iteration period %.3f ms, %d kernels" % (wall / 1e6, len(ev)))
for k, v in sorted(famtime.items(), key=lambda kv: -kv[1])[:16]:
    print("  %-32s %7.3f ms (%4.1f%%)" % (k, v / 1e6, 100.0 * v / wall))
tot = {}
for s, e, f, n, q in ev:
    tot.setdefault(f, [0, 0]); tot[f][0] += e - s; tot[f][1] += 1
print("kernel time by family (sum of durations):")
for f, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("  %-10s %7.3f ms  %5d launches  avg %7.2f us" % (f, t / 1e6, c, t / c / 1e3))
if len(sys.argv) > 2:       # dump the sequence of long kernels / gaps
    prev_end = lo
    for s, e, f, n, q in ev:
        if f in ("gemm", "split", "ce", "other", "splitk") or s - prev_end > 20000:
            print("%9.1f us +%7.1f us gap %6.1f  q%s %s" % ((s - lo) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, (n[:48] + " " + n[n.rindex("["):]) if "[" in n else n[:60]))
        prev_end = max(prev_end, e)
