"""Run bench.py against library variants built with extra -D defines (each variant replaces the in-tree library
for its run, the product build is restored afterwards).  usage: VARIANTS="name:DEF1,DEF2;..." python tools/bench_variants.py [bench args]"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from s2vt_video_caption_amd import build  # noqa: E402

variants = [("base", [])]
for spec in os.environ.get("VARIANTS", "").split(";"):
    if spec:
        n, d = spec.split(":")
        variants.append((n, d.split(",")))
os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
keep = build.LIB + ".keep"
shutil.copy(build.LIB, keep)
try:
    for name, defs in variants:
        path = build.build(defines=defs, out_path=os.path.join(ROOT, "gpurun_out", "variants", "libv_%s.so" % name))
        shutil.copy(path, build.LIB)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:],
                           capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-14s %9.0f frames/s %7.3f ms/step  live %s | isolated %s | decode %s" % (
                name, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["kernel_ms_per_step_isolated"],
                d["decode"]["value"]), flush=True)
        except Exception as e:
            print(name, "FAILED", e, r.stdout[-300:], r.stderr[-600:], flush=True)
finally:
    shutil.copy(keep, build.LIB)
    os.remove(keep)
