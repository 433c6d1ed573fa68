"""Where do the 1500 workgroups of one logits_argmax_kernel launch (B=128, H=1000, V=12000) spend the launch?  Builds the
EXPERIMENT library (in-kernel 100-MHz stamps, csrc/experiment.h) and prints, from the stamps of EVERY workgroup: when the
workgroups start (dispatch rounds), how long one lives, and the phases of wave 0 (contraction, wait for the slowest wave,
8-way reduction, argmax).  Stamped builds run slower than the product: read the shares."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import s2vt_video_caption_amd  # noqa
from s2vt_video_caption_amd import build, capi

HERE = os.path.dirname(os.path.abspath(build.__file__))
xlib = os.path.join(HERE, "libs2vt_hip_stamps.so")
build.build(defines=("S2VT_EXPERIMENT_STAMPS",), out_path=xlib)
capi.LIB_PATH = xlib
lib = capi.load()
lib.s2vt_experiment_set_stamps.restype = ctypes.c_int
lib.s2vt_experiment_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
from s2vt_video_caption_amd import ops

DEV = "cuda:0"
B, H, V = 128, 1000, 12000
g = torch.Generator().manual_seed(1)
h = torch.randn(B, H, generator=g).to(DEV)
w = (torch.randn(V, H, generator=g) * 0.03).to(DEV)
b = torch.zeros(V, device=DEV)
for _ in range(3):
    ops.decode_step_argmax(h, w, b)
stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=DEV)
lib.s2vt_experiment_set_stamps(ctypes.c_void_p(stamps.data_ptr()), 0)
ops.decode_step_argmax(h, w, b)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(4096, 16)
rec = s[s[:, 0] > 0][:, :5].astype(np.float64) * 0.01          # us
t0 = rec[:, 0].min()
start, end = rec[:, 0] - t0, rec[:, 4] - t0
life = end - start
print("workgroups with stamps: %d; launch span (first start -> last end): %.1f us" % (len(rec), end.max()))
print("start times: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us; lifetime: min %.1f  p50 %.1f  p90 %.1f  max %.1f us" %
      (np.percentile(start, 10), np.percentile(start, 50), np.percentile(start, 90), start.max(), life.min(), np.percentile(life, 50),
       np.percentile(life, 90), life.max()))
edges = np.arange(0, end.max() + 5, 5.0)
alive = [(int(((start <= t) & (end > t)).sum())) for t in edges]
print("workgroups alive every 5 us: " + " ".join(str(a) for a in alive))
d = np.diff(rec, axis=1)
print("wave 0 phases (us, mean / p90): contraction %.2f / %.2f  wait for slowest wave %.2f / %.2f  partial tiles + barrier %.2f / %.2f  "
      "reduce + argmax %.2f / %.2f" % tuple(x for i in range(4) for x in (d[:, i].mean(), np.percentile(d[:, i], 90))))
for lo, hi in ((0, 5), (5, 15), (15, 30), (30, 1e9)):
    sel = (start >= lo) & (start < hi)
    if sel.any():
        print("  started in [%g, %g) us: %d workgroups, mean lifetime %.1f us, contraction %.1f us" % (lo, hi, sel.sum(), life[sel].mean(), d[sel, 0].mean()))
