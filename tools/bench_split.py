"""Bandwidth of the plane-split kernel (split_dual_kernel<3,false>, row planes only) on the operand shapes of a config-2 step:
bytes = fp32 read + three bf16 planes written.  usage: python tools/bench_split.py   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from s2vt_video_caption_amd import capi
from s2vt_video_caption_amd.functional import _ptr, _stream
lib = capi.load()
dev = torch.device("cuda:0")
SHAPES = [("feats [B L, F]", 5120, 4096), ("dG [T B, 4H]", 10176, 4000), ("dG block [27 B, 4H]", 1728, 4000), ("logits [R, V]", 5056, 12000),
          ("h block [27 B, H]", 1728, 1000), ("x1 [L B, H]", 5120, 1000), ("W_o [V, H]", 12000, 1000), ("W_f [H, F]", 1000, 4096)]
for name, rows, cols in SHAPES:
    x = torch.randn(rows, cols, device=dev)
    kpad = (cols + 63) // 64 * 64
    out = torch.zeros((rows + 63) // 64 * 64, 3 * kpad, dtype=torch.int16, device=dev)
    best = 1e9
    for it in range(8):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        capi.check(lib.s2vt_split_planes(3, 0, _ptr(x), x.stride(0), rows, cols, _ptr(out), 3 * kpad, kpad, rows, _stream(dev)), "split")
        e1.record()
        torch.cuda.synchronize()
        if it:
            best = min(best, e0.elapsed_time(e1))
    nbytes = rows * cols * 4 + rows * kpad * 6
    print("%-22s %6d x %5d  %7.1f us  %6.2f TB/s  (%.0f MB)" % (name, rows, cols, best * 1e3, nbytes / best / 1e9, nbytes / 1e6), flush=True)
