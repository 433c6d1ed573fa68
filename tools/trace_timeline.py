"""Timeline statistics from a rocprofv3 --kernel-trace CSV: per kernel family busy time, union busy time, idle
time and how much of the wall time has 1 / 2+ kernels in flight.  usage: trace_timeline.py trace.csv [t_from_frac t_to_frac]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    fam = ("gemm" if ("gemm_f32" in n or "gemm_x3" in n or "gemm_b1" in n) else "splitk" if "splitk" in n else "split" if ("split_" in n or "split3" in n) else
           "rec_fwd" if ("lstm_step_fwd" in n or "lstm_seq_fwd" in n) else "rec_bwd" if ("lstm_step_bwd" in n or "lstm_seq_bwd" in n) else
           "argmax" if "logits_argmax" in n else "ce" if ("ce_" in n or "mask_criterion" in n) else
           "adam" if ("multi_tensor" in n or "adam" in n) else "emb" if "emb_" in n else "other")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
hi = t0 + (t1 - t0) * float(sys.argv[3]) if len(sys.argv) > 3 else t1
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
pts = []
for s, e, f in ev:
    pts.append((s, 1, f)); pts.append((e, -1, f))
pts.sort()
depth, last, hist, famtime = 0, pts[0][0], {}, {}
active = {}
for t, d, f in pts:
    dt = t - last
    if dt > 0:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + dt
        key = "+".join(sorted(k for k, v in active.items() if v > 0)) or "idle"
        famtime[key] = famtime.get(key, 0) + dt
    depth += d
    active[f] = active.get(f, 0) + d
    last = t
wall = pts[-1][0] - pts[0][0]
print("window %.3f ms, kernels %d" % (wall / 1e6, len(ev)))
for k in sorted(hist):
    print("  %d%s kernels in flight: %.3f ms (%.1f%%)" % (k, "+" if k == 3 else "", hist[k] / 1e6, 100.0 * hist[k] / wall))
for k, v in sorted(famtime.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-40s %.3f ms (%.1f%%)" % (k, v / 1e6, 100.0 * v / wall))
