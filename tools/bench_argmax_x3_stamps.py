"""In-kernel stamps of one logits_argmax_x3_kernel launch (B=128, H=1000, V=12000): EXPERIMENT build (csrc/experiment.h).
Per workgroup: start, first stage landed, stage 0 done, stage 15 done, main loop done, tail done, epilogue done (100-MHz
wall clock) and the core clock the workgroup ran at (s_memtime / s_memrealtime).  S2VT_AX_DBG: 1 no DMA requests, 2 no
atomics (safe here: nothing consumes the result)."""
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
for _ in range(5):
    ops.decode_step_argmax(h, w, b, planes=True)
stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=DEV)
lib.s2vt_experiment_set_stamps(ctypes.c_void_p(stamps.data_ptr()), 0)
ops.decode_step_argmax(h, w, b, planes=True)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(4096, 16)
s = s[s[:, 0] > 0]
rec = s[:, :7].astype(np.float64) * 0.01          # us
t0 = rec[:, 0].min()
names = ["start", "requests issued", "stage 0 done", "stage 15 done", "main loop done", "tail done", "epilogue done"]
print("workgroups: %d   (S2VT_AX_DBG=%s)" % (len(rec), os.environ.get("S2VT_AX_DBG", "0")))
for i, n in enumerate(names):
    v = rec[:, i] - t0
    print("  %-18s p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (n, np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
d = np.diff(rec, axis=1)
print("  phases (p50 us): " + "  ".join("%s %.2f" % (names[i + 1], np.percentile(d[:, i], 50)) for i in range(6)))
cyc = (s[:, 9] - s[:, 8]).astype(np.float64)
wall = (s[:, 5] - s[:, 0]).astype(np.float64) * 10e-9
print("  core clock while in the kernel: p50 %.2f GHz (cycles start->tail %.0f)" % (np.percentile(cyc / wall, 50) / 1e9, np.percentile(cyc, 50)))
per_stage = (rec[:, 3] - rec[:, 2]) / 15.0
print("  per k32 stage (stages 1..15): p50 %.3f us = %.0f cycles" % (np.percentile(per_stage, 50), np.percentile(per_stage * (cyc / wall) * 1e-6, 50)))
