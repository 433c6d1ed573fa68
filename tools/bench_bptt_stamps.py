"""Where a sub-step of the persistent bf16 BPTT kernel spends its time (two layers per launch, B = 256, H = 1000: the config-3
launch): EXPERIMENT build with in-kernel 100-MHz stamps (csrc/experiment.h), one workgroup's view."""
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
DEFS = (("S2VT_EXPERIMENT_STAMPS",) + (("S2VT_EXPERIMENT_PLAIN_LOADS",) if os.environ.get("PLAIN_LOADS") == "1" else ()) +
        (("S2VT_EXPERIMENT_PLAIN_STORES",) if os.environ.get("PLAIN_STORES") == "1" else ()))
print("build defines:", DEFS)
build.build(defines=DEFS, out_path=xlib)          # (an out_path build always recompiles)
capi.LIB_PATH = xlib
lib = capi.load()
lib.s2vt_experiment_set_stamps.restype = ctypes.c_int
lib.s2vt_experiment_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
from s2vt_video_caption_amd import ops

DEV = "cuda:0"
T, B, H = 32, 256, 1000
NAMES = ["poll+barrier", "operand loads + first requests", "k loop", "partials + barrier", "cell math", "barrier", "store issue",
         "drain (wave 0)", "signal"]
g = torch.Generator().manual_seed(1)
w0 = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
w1 = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
c0 = torch.randn(T * B, H, generator=g).to(DEV)
c1 = torch.randn(T * B, H, generator=g).to(DEV)
g0 = torch.rand(T * B, 4 * H, generator=g).to(DEV)
g1 = torch.rand(T * B, 4 * H, generator=g).to(DEV)
dh0 = (torch.randn(T * B, H, generator=g) * 0.01).to(DEV)
dh1 = (torch.randn(T * B, H, generator=g) * 0.01).to(DEV)
for _ in range(2):
    ops.lstm_seq_bwd_bf16_pair(w0, w1, dh0, dh1, 0, c0, c1, g0, g1, T, B, H, block=0)
ref = ops.lstm_seq_bwd_bf16(w0, dh0, 0, c0, g0, T, B, H, persistent=False)
got = ops.lstm_seq_bwd_bf16_pair(w0, w1, dh0, dh1, 0, c0, c1, g0, g1, T, B, H, block=0)[0]
print("max |dG(persistent pair) - dG(launch per timestep)| = %.3e (scale %.3e)" % ((got - ref).abs().max().item(), ref.abs().max().item()))
for blockid in (0, 17, 100, 200):
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=DEV)
    lib.s2vt_experiment_set_stamps(ctypes.c_void_p(stamps.data_ptr()), blockid)
    ops.lstm_seq_bwd_bf16_pair(w0, w1, dh0, dh1, 0, c0, c1, g0, g1, T, B, H, block=0)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(4096, 16)
    ns = 2
    rec = s[ns * 4:ns * T]
    if rec[:, 0].min() == 0:
        print("workgroup %d: no stamps" % blockid)
        continue
    d = np.diff(rec[:, :10].astype(np.float64), axis=1) * 0.01
    tot = (rec[1:, 0] - rec[:-1, 0]).astype(np.float64) * 0.01
    print("workgroup %d: sub-step period %.2f us (min %.2f max %.2f)" % (blockid, tot.mean(), tot.min(), tot.max()))
    print("   " + "  ".join("%s %.2f" % (n, v) for n, v in zip(NAMES, d.mean(axis=0))))
