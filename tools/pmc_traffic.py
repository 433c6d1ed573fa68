"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel family.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request of wide streaming reads ->
doubled; WRITE_SIZE is exact for 16-B/lane streaming stores.  Units of both counters: KB."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
total = collections.defaultdict(float)          # every kernel of the passes, whatever its family: HBM-side bytes of a whole pass
nrows = collections.defaultdict(int)
for tag in ("fetch", "write", "l2"):
    for f in glob.glob(os.path.join(d, tag + "*_counter_collection.csv")):
        rows = list(csv.DictReader(open(f)))
        # the passes start at the library's first dispatch: what comes before (moving the model, the optimizer's flat buffers) is set-up
        first = min((int(r["Dispatch_Id"]) for r in rows if "s2vt::" in r["Kernel_Name"]), default=0)
        for r in rows:
            n = r["Kernel_Name"]
            if int(r["Dispatch_Id"]) >= first:
                total[r["Counter_Name"]] += float(r["Counter_Value"])
                nrows[r["Counter_Name"]] += 1
            fam = None
            for key in ("gemm_x3_kernel", "gemm_f32_kernel", "gemm_b1_kernel", "lstm_step_fwd_kernel", "lstm_step_bwd_kernel",
                        "lstm_step_fwd_bf16_kernel", "lstm_step_bwd_bf16_kernel", "lstm_seq_fwd_bf16_persist_kernel",
                        "lstm_seq_bwd_bf16_persist_kernel", "lstm_seq_fwd_f32_persist_kernel", "lstm_seq_bwd_f32_persist_kernel",
                        "lstm_seq_fwd_x3_persist_kernel", "lstm_seq_bwd_x3_persist_kernel", "split3_rows_kernel",
                        "split_dual_kernel", "logits_argmax_x3_kernel", "logits_argmax_kernel", "ce_row_kernel", "ce_bwd_kernel"):
                if key in n and fam is None:
                    fam = key
            if fam:
                acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for fam, c in acc.items():
    f = c.get("FETCH_SIZE", [])
    w = c.get("WRITE_SIZE", [])
    e = {"launches": max(len(f), len(w))}
    if f:
        e["fetch_bytes_per_launch_raw"] = sum(f) / len(f) * 1024
        e["fetch_bytes_per_launch_corrected"] = 2 * sum(f) / len(f) * 1024
    if w:
        e["write_bytes_per_launch"] = sum(w) / len(w) * 1024
    if f and w:
        e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch_corrected"] + e["write_bytes_per_launch"]
    h, m = c.get("TCC_HIT_sum", []), c.get("TCC_MISS_sum", [])
    if h and m:
        e["l2_hit_rate"] = sum(h) / (sum(h) + sum(m))
    if "persist" in fam and "hbm_bytes_per_launch" in e:
        # a persistent launch covers a block of timesteps of one or two layers: 2 x 159 layer timesteps per pass in all
        iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
        e["layer_timesteps_per_pass"] = 318
        e["hbm_bytes_per_layer_timestep"] = e["hbm_bytes_per_launch"] * e["launches"] / (318.0 * iters)
    out[fam] = e
what = sys.argv[2] if len(sys.argv) > 2 else "c2"
commit = os.environ.get("S2VT_COMMIT", "unrecorded")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import source_digest  # noqa: E402
iters_ = int(sys.argv[3]) if len(sys.argv) > 3 else 2
step_total = None
if total.get("FETCH_SIZE") and total.get("WRITE_SIZE"):
    # every dispatch of the profiled program (library kernels, torch's, runtime fills and copies), FETCH_SIZE doubled throughout
    # (the guide's correction for wide streaming reads: an upper bound where a kernel reads narrow), per pass of the driver
    step_total = {"passes": iters_, "dispatches_per_pass": nrows["FETCH_SIZE"] / iters_,
                  "fetch_bytes_corrected_per_pass": 2 * total["FETCH_SIZE"] * 1024 / iters_,
                  "write_bytes_per_pass": total["WRITE_SIZE"] * 1024 / iters_,
                  "hbm_bytes_per_pass": (2 * total["FETCH_SIZE"] + total["WRITE_SIZE"]) * 1024 / iters_}
print(json.dumps({"commit": commit, "source_digests": source_digest.digests(), "all_kernels": step_total, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (separate passes), tools/prof_path.py %s "
                            "(train forward+backward passes), FETCH_SIZE doubled per MI355X_MICROARCH.md" % what,
                  "kernels": out}, indent=1))
