"""Does a low-footprint timestep-like kernel (tools/micro/coresident.hip: 4 waves, 58 VGPRs, 12.7 KB LDS) make progress
INSIDE the split-precision GEMM, which keeps the shipped timestep kernels (73 KB LDS) off its CUs?  Times: the chain of
light kernels alone, the GEMM alone, both on two streams; and the shipped timestep kernel beside the GEMM for comparison.
usage: python tools/bench_coresident.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import capi, ops  # noqa: E402

capi.load()
so = os.path.join(ROOT, "gpurun_out", "libcoresident.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
                       os.path.join(ROOT, "tools", "micro", "coresident.hip"), "-o", so])
light = ctypes.CDLL(so)
vp = ctypes.c_void_p
dev = "cuda:0"
B, H = 64, 1000
h = torch.randn(B, H, device=dev)
w = torch.randn(4 * H, H, device=dev) * 0.03
out = torch.empty(B, 4 * H, device=dev)
M, N, K = 5056, 12000, 1000
pa, pb = ops.split_planes(torch.randn(M, K, device=dev)), ops.split_planes(torch.randn(N, K, device=dev))
c = torch.empty(M, N, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
T = 159
gx = torch.randn(T * B, 4 * H, device=dev)
bias = torch.zeros(4 * H, device=dev)
NG, NL = 12, 480


def chain_light():
    for _ in range(NL):
        light.light_step_launch(vp(sa.cuda_stream), vp(h.data_ptr()), vp(w.data_ptr()), vp(out.data_ptr()), B, H, H)


def chain_shipped():
    with torch.cuda.stream(sa):
        for _ in range(NL // T):
            ops.lstm_seq_fwd(T, B, gx, T, bias, w, want_stash=True)


def gemms():
    with torch.cuda.stream(sb):
        for _ in range(NG):
            ops.gemm_planes(pa, pb, M, N, out=c)


def timed(fa, fb):
    torch.cuda.synchronize()
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    if fa:
        ea[0].record(sa)
    if fb:
        eb[0].record(sb)
    # interleave the enqueue so that both streams have work from the start
    if fb:
        fb()
    if fa:
        fa()
    if fa:
        ea[1].record(sa)
    if fb:
        eb[1].record(sb)
    torch.cuda.synchronize()
    return (ea[0].elapsed_time(ea[1]) if fa else 0.0), (eb[0].elapsed_time(eb[1]) if fb else 0.0)


for name, chain, n in (("light kernel", chain_light, NL), ("shipped lstm_step_fwd_kernel", chain_shipped, (NL // T) * T)):
    chain(); gemms()
    a0, _ = timed(chain, None)
    _, g0 = timed(None, gemms)
    a1, g1 = timed(chain, gemms)
    print("%s: chain alone %.2f us/launch; GEMM alone %.1f us/call; together: chain %.2f us/launch (%.2f ms), GEMM %.1f us/call (%.2f ms); "
          "sum of alone %.2f ms, together max %.2f ms" % (name, a0 * 1e3 / n, g0 * 1e3 / NG, a1 * 1e3 / n, a1, g1 * 1e3 / NG, g1,
                                                          a0 + g0, max(a1, g1)), flush=True)
