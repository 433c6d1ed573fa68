"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter over dispatches."""
import collections
import csv
import glob
import sys

out = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void s2vt::", "")[:44]
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(out.items()):
    if not any(k in name for k in ("gemm", "lstm", "logits", "ce_", "colsum", "splitk")):
        continue
    n = max(len(v) for v in cs.values())
    print("%-46s n=%d" % (name, n))
    for c, v in sorted(cs.items()):
        print("    %-28s mean=%14.1f  min=%14.1f max=%14.1f" % (c, sum(v) / len(v), min(v), max(v)))
