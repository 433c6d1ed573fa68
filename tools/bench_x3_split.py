"""Time the split-precision GEMM on every shape of the C2 train step for forced split-K factors (S2VT_X3_NSPLIT is read
once per process, so each factor runs in a child process).  usage: python tools/bench_x3_split.py   (GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("x1", 5120, 1000, 4096), ("gx1", 5120, 4000, 1000), ("gx2blk", 2048, 4000, 1000), ("logits", 5056, 12000, 1000),
          ("dh2dec", 5056, 1000, 12000), ("dW_o", 12000, 1000, 5056), ("dh1blk", 2048, 1000, 4000), ("dW_hh", 4000, 1000, 10112),
          ("dW_e", 4000, 1000, 5056), ("de", 5056, 1000, 4000), ("dW_f", 1000, 4096, 5120)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    from s2vt_video_caption_amd import capi, ops
    capi.load()
    dev = "cuda:0"
    ws = torch.empty(256 << 20, device=dev)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
        pa, pb = ops.split_planes(a), ops.split_planes(b)
        c = torch.empty(M, N, device=dev)
        best = 1e9
        for it in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_planes(pa, pb, M, N, out=c, splitk_ws=ws); e1.record()
            torch.cuda.synchronize()
            if it:
                best = min(best, e0.elapsed_time(e1))
        print("%s %.1f" % (name, best * 1e3), flush=True)
    sys.exit(0)
res = {}
for n in (0, 1, 2, 3, 4, 6, 8):
    env = dict(os.environ)
    env["S2VT_X3_NSPLIT"] = str(n) if n else "0"
    if n == 1:
        env["S2VT_X3_NSPLIT"] = "1"
    out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=300).stdout
    for line in out.splitlines():
        k, v = line.split()
        res.setdefault(k, {})[n] = float(v)
print("%-8s %s" % ("shape", " ".join("%8s" % ("auto" if n == 0 else "n=%d" % n) for n in (0, 1, 2, 3, 4, 6, 8))))
for name, M, N, K in SHAPES:
    print("%-8s %s   (M=%d N=%d K=%d)" % (name, " ".join("%8.1f" % res.get(name, {}).get(n, float("nan")) for n in (0, 1, 2, 3, 4, 6, 8)), M, N, K))
