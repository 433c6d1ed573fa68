"""Where a sub-step of the split-precision persistent BPTT (lstm_persist_x3.hip, reduce-scatter over the gate columns) spends its
time: HIP-event time of the product kernel, then (STAMPS=1) an EXPERIMENT build with in-kernel 100-MHz stamps."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import s2vt_video_caption_amd  # noqa
from s2vt_video_caption_amd import build, capi

STAMPS = os.environ.get("STAMPS", "1") == "1"
if STAMPS:
    HERE = os.path.dirname(os.path.abspath(build.__file__))
    xlib = os.path.join(HERE, "libs2vt_hip_stamps.so")
    build.build(defines=("S2VT_EXPERIMENT_STAMPS",), out_path=xlib)
    capi.LIB_PATH = xlib
lib = capi.load()
if STAMPS:
    lib.s2vt_experiment_set_stamps.restype = ctypes.c_int
    lib.s2vt_experiment_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
from s2vt_video_caption_amd import ops

DEV = "cuda:0"
T, H = 48, 1000
NAMES = ["poll+barrier", "gather issue", "gather landed + barrier", "reduce 63 partials", "cell math + tile", "barrier + 192 MFMA + stores", "drain", "barrier + signal"]
for B, pair in ((64, False), (64, True), (128, True)):
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
    gates = torch.sigmoid(torch.randn(T * B, 4 * H, generator=g)).to(DEV)
    c_all = (torch.randn(T * B, H, generator=g) * 0.7).to(DEV)
    dh = (torch.randn(T * B, H, generator=g) * 0.1).to(DEV)
    second = (w, dh, c_all, gates) if pair else None
    if STAMPS:
        lib.s2vt_experiment_set_stamps(None, -1)
    for _ in range(2):
        ops.lstm_seq_bwd_persist(T, B, w, dh, 0, c_all, gates, block=0, second=second)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    ops.lstm_seq_bwd_persist(T, B, w, dh, 0, c_all, gates, block=0, second=second)
    e1.record()
    torch.cuda.synchronize()
    print("B=%d %s: %.1f us per timestep (whole call incl. W^T transpose / split and allocations, T=%d)" % (B, "two layers" if pair else "one layer", e0.elapsed_time(e1) * 1e3 / T, T))
    if not STAMPS:
        continue
    for blockid in (0, 17, 62, 100):
        stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=DEV)
        lib.s2vt_experiment_set_stamps(ctypes.c_void_p(stamps.data_ptr()), blockid)
        ops.lstm_seq_bwd_persist(T, B, w, dh, 0, c_all, gates, block=0, second=second)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(4096, 16)
        ns = 2 if (B == 128 and pair) else 1
        rec = s[ns * 4:ns * (T - 1)]
        if rec[:, 0].min() == 0:
            print("   workgroup %d: no stamps" % blockid)
            continue
        d = np.diff(rec[:, :9].astype(np.float64), axis=1) * 0.01
        tot = (rec[1:, 0] - rec[:-1, 0]).astype(np.float64) * 0.01
        print("   workgroup %d: sub-step period %.2f us (min %.2f max %.2f)" % (blockid, tot.mean(), tot.min(), tot.max()))
        print("      " + "  ".join("%s %.2f" % (n, v) for n, v in zip(NAMES, d.mean(axis=0))))
