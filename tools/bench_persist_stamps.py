"""Where a sub-step of the persistent recurrence kernel spends its time: builds an EXPERIMENT library with in-kernel
100-MHz wall-clock stamps (csrc/experiment.h) next to the product library and prints per-phase averages for one
workgroup.  Stamped builds run slower than the product: read the shares, not the length."""
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
T, H = 48, 1000
NAMES = ["poll", "dma issue", "first chunk landed", "k loop", "partials", "cell math", "store issue", "drain", "signal"]
for B in (256, 64):
    g = torch.Generator().manual_seed(1)
    gx = torch.randn(T * B, 4 * H, generator=g).to(DEV)
    bias = (torch.randn(4 * H, generator=g) * 0.3).to(DEV)
    w = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
    for blockid in (0, 17, 100):
        stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=DEV)
        lib.s2vt_experiment_set_stamps(ctypes.c_void_p(stamps.data_ptr()), blockid)
        ops.lstm_seq_fwd_bf16(gx, 24, bias, w, T, B, H, persistent=True, block=0)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(4096, 16)
        ns = 2 if B % 128 == 0 else 1
        rec = s[ns * 4:ns * T]                       # skip the first steps
        if rec[:, 0].min() == 0:
            print("B=%d block %d: no stamps (block not in grid)" % (B, blockid))
            continue
        d = np.diff(rec[:, :10].astype(np.float64), axis=1) * 0.01        # us
        tot = (rec[1:, 0] - rec[:-1, 0]).astype(np.float64) * 0.01
        print("B=%d workgroup %d: sub-step period %.2f us (min %.2f max %.2f)" % (B, blockid, tot.mean(), tot.min(), tot.max()))
        print("   " + "  ".join("%s %.2f" % (n, v) for n, v in zip(NAMES, d.mean(axis=0))))


# ---- fp32 forward (config 2 arithmetic)
NAMES32 = ["poll", "ring prologue issue", "first chunk (wait + 16 MFMA)", "7 more chunks", "partials", "cell math", "store issue", "drain", "signal"]
for B, pair in ((64, False), (64, True)):
    g = torch.Generator().manual_seed(1)
    gx = torch.randn(24 * B, 4 * H, generator=g).to(DEV)
    bias = (torch.randn(4 * H, generator=g) * 0.3).to(DEV)
    w = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=DEV)
    lib.s2vt_experiment_set_stamps(ctypes.c_void_p(stamps.data_ptr()), 17)
    ops.lstm_seq_fwd_persist(T, B, gx, 24, bias, w, block=0, second=(gx, bias, w) if pair else None)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(4096, 16)
    rec = s[4:T]
    d = np.diff(rec[:, :10].astype(np.float64), axis=1) * 0.01
    tot = (rec[1:, 0] - rec[:-1, 0]).astype(np.float64) * 0.01
    print("fp32 B=%d %s workgroup 17: sub-step period %.2f us (min %.2f max %.2f)" % (B, "two layers" if pair else "one layer", tot.mean(), tot.min(), tot.max()))
    print("   " + "  ".join("%s %.2f" % (n, v) for n, v in zip(NAMES32, d.mean(axis=0))))
