"""Where does one workgroup of gemm_b1_kernel spend a k64 stage?  Experiment library (in-kernel 100-MHz stamps of wave 0 of one
workgroup) on the logits shape of config 3 (20224 x 12000 x 1024: 16 stages per tile)."""
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
lib.s2vt_experiment_set_b1_stamps.restype = ctypes.c_int
lib.s2vt_experiment_set_b1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
from s2vt_video_caption_amd import ops

DEV = "cuda:0"
M, N, K = 20224, 12000, 1000
a = torch.randn(M, K, device=DEV)
b = torch.randn(N, K, device=DEV) * 0.05
pa, pb = ops.split_planes(a, 1), ops.split_planes(b, 1)
c = torch.empty(M, N, device=DEV)
for _ in range(3):
    ops.gemm_planes(pa, pb, M, N, nplanes=1, out=c)
NAMES = ["wait own requests (vmcnt 0)", "barrier", "first fragments", "product 0 + 2 requests", "product 1 + 2 requests",
         "product 2 + 2 requests", "product 3 + 2 requests"]
for blk in (0, 77, 1500):
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=DEV)
    lib.s2vt_experiment_set_b1_stamps(ctypes.c_void_p(stamps.data_ptr()), blk)
    ops.gemm_planes(pa, pb, M, N, nplanes=1, out=c)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(4096, 16)[:16, :8].astype(np.float64) * 0.01
    if s[:, 0].min() == 0:
        print("block %d: no stamps" % blk)
        continue
    d = np.diff(s, axis=1)
    per = np.diff(s[:, 0])
    print("block %d: stage period %.2f us (min %.2f max %.2f); phases (mean over stages 2..15):" % (blk, per[1:].mean(), per[1:].min(), per[1:].max()))
    print("   " + "  ".join("%s %.2f" % (n, v) for n, v in zip(NAMES, d[2:].mean(axis=0))))
