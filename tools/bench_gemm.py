"""Micro-benchmark of gemm_f32_kernel on the GEMM shapes of the C2 train step (B=64): TFLOP/s per shape, with
and without split-K, for library variants built with -D defines.   usage: python tools/bench_gemm.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import build  # noqa: E402

VARIANTS = {"base": []}
for spec in os.environ.get("GEMM_VARIANTS", "").split(";"):
    if spec:
        name, defs = spec.split(":")
        VARIANTS[name] = defs.split(",")
# (name, a_kmajor, b_kmajor, M, N, K)
SHAPES = [("x1 NT", 1, 1, 5120, 1000, 4096), ("gx1 NT", 1, 1, 5120, 4000, 1000), ("gx2 NT", 1, 1, 10176, 4000, 1000),
          ("logits NT", 1, 1, 5056, 12000, 1000), ("dh2dec NN", 1, 0, 5056, 1000, 12000),
          ("dh1 NN", 1, 0, 10176, 1000, 4000), ("dx1 NN", 1, 0, 5120, 1000, 4000),
          ("dW_o TN", 0, 0, 12000, 1000, 5056), ("dW_hh TN", 0, 0, 4000, 1000, 10112), ("dW_ih1 TN", 0, 0, 4000, 1000, 5120),
          ("dW_f TN", 0, 0, 1000, 4096, 5120)]
dev = "cuda:0"
vp = ctypes.c_void_p
os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
ws = torch.empty(64 << 20, device=dev)
for vname, defs in VARIANTS.items():
    path = build.build(defines=defs, out_path=os.path.join(ROOT, "gpurun_out", "variants", "libg_%s.so" % vname))
    lib = ctypes.CDLL(path)
    tot = {0: 0.0, 1: 0.0}
    for (name, ak, bk, M, N, K) in SHAPES:
        a = torch.randn((M, K) if ak else (K, M), device=dev)
        b = torch.randn((N, K) if bk else (K, N), device=dev)
        c = torch.empty(M, N, device=dev)
        st = vp(torch.cuda.current_stream().cuda_stream)
        res = []
        for use_ws in (0, 1):
            def run():
                rc = lib.s2vt_gemm_f32_splitk(ak, bk, M, N, K, vp(a.data_ptr()), ctypes.c_int64(a.stride(0)),
                                              vp(b.data_ptr()), ctypes.c_int64(b.stride(0)), vp(c.data_ptr()),
                                              ctypes.c_int64(N), vp(0), 0, vp(ws.data_ptr() if use_ws else 0),
                                              ctypes.c_size_t(ws.numel() if use_ws else 0), st)
                assert rc == 0
            run()
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            res.append(best)
            tot[use_ws] += best
        fl = 2.0 * M * N * K
        print("%-6s %-10s M=%5d N=%5d K=%5d  direct %7.1f us %6.1f TF | split-K ws %7.1f us %6.1f TF" %
              (vname, name, M, N, K, res[0] * 1e3, fl / res[0] / 1e9, res[1] * 1e3, fl / res[1] / 1e9), flush=True)
    print("%-6s total direct %.2f ms, with split-K %.2f ms" % (vname, tot[0], tot[1]), flush=True)
