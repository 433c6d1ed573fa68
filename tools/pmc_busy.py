"""Reduce rocprofv3 --pmc SQ passes (tools/profile_round4.sh) to per-kernel pipe figures, one JSON + one text table.

  python3 tools/pmc_busy.py DIR "what ran" > profiles/round4_pmc_<cfg>.json      (text table on stderr)

Per kernel (template instantiations kept apart), means over its dispatches:
  mfma_busy        = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): share of all SIMD-cycles of the dispatch in
                     which the matrix pipe was executing (MI355X_MICROARCH.md: the counter counts cycles, 32 per 32x32x16 bf16 MFMA;
                     GRBM_GUI_ACTIVE is summed over the 8 XCDs)
  mfma_busy_sq     = the same against SQ_BUSY_CYCLES x 4 SIMDs ... (SQ_BUSY_CYCLES is per SE-summed; kept for cross-checking only)
  wait_share       = SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked on s_waitcnt / barriers
  issue_stall      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   waves stalled at issue (MFMA dependency, pipe busy)
  active_share     = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  mfma_insts, lds_insts, valu_insts, mfma_mops_bf16 / f32 per dispatch
"""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
what = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "s2vt::" not in n:
            continue
        n = re.sub(r"\(.*$", "", n.replace("void ", "").replace("s2vt::", "")).strip()
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
KEEP = ("gemm", "lstm", "logits_argmax", "split", "ce_", "top20")
for n, c in sorted(acc.items()):
    if not any(k in n for k in KEEP):
        continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    e = {"dispatches": max(len(v) for v in c.values())}
    g = m.get("GRBM_GUI_ACTIVE")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        e["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8.0 * 1024.0), 4)
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for key, ctr in (("wait_share", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("active_share", "SQ_ACTIVE_INST_ANY"),
                         ("lds_issue_stall", "SQ_WAIT_INST_LDS")):
            if ctr in m:
                e[key] = round(m[ctr] / wc, 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    for key, ctr in (("mfma_insts", "SQ_INSTS_MFMA"), ("lds_insts", "SQ_INSTS_LDS"), ("valu_insts", "SQ_INSTS_VALU"),
                     ("mfma_mops_bf16", "SQ_INSTS_VALU_MFMA_MOPS_BF16"), ("mfma_mops_f32", "SQ_INSTS_VALU_MFMA_MOPS_F32"),
                     ("gui_active_cycles_8xcd", "GRBM_GUI_ACTIVE"), ("mfma_busy_cycles", "SQ_VALU_MFMA_BUSY_CYCLES"),
                     ("sq_busy_cycles", "SQ_BUSY_CYCLES")):
        if ctr in m:
            e[key] = round(m[ctr], 1)
    out[n] = e
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import source_digest  # noqa: E402
print(json.dumps({"commit": os.environ.get("S2VT_COMMIT", "unrecorded"), "source_digests": source_digest.digests(),
                  "source": "rocprofv3 --pmc (SQ / GRBM passes, counters only, program directly after --): %s" % what,
                  "kernels": out}, indent=1))
for n, e in out.items():
    print("%-58s n=%4d  mfma_busy %6s  wait %6s  issue_stall %6s  active %6s  lds_conflict %6s" % (
        n[:58], e["dispatches"], e.get("mfma_busy", "-"), e.get("wait_share", "-"), e.get("issue_stall", "-"),
        e.get("active_share", "-"), e.get("lds_conflict", "-")), file=sys.stderr)
