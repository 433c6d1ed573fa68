"""Accuracy and speed of the split-precision (3 bf16 planes, 6 products) GEMM vs the fp32-MFMA GEMM.
usage: python tools/bench_gemm_bf16.py   (GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import build, capi  # noqa: E402

lib = capi.load()
if os.environ.get("X3_DEFS"):       # experiment: a variant library built with extra -D defines
    os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
    lib = ctypes.CDLL(build.build(defines=os.environ["X3_DEFS"].split(","),
                                  out_path=os.path.join(ROOT, "gpurun_out", "variants", "libx3_%s.so" % os.environ["X3_DEFS"].replace("=", "").replace(",", "_"))))
dev = "cuda:0"
vp = ctypes.c_void_p
i64 = ctypes.c_int64
SHAPES = [("logits", 5056, 12000, 1000), ("gx2", 10176, 4000, 1000), ("x1", 5120, 1000, 4096),
          ("dW_hh", 4000, 1000, 10112), ("dh2dec", 5056, 1000, 12000), ("small", 200, 300, 104)]
ws = torch.empty(64 << 20, device=dev)
st = vp(torch.cuda.current_stream().cuda_stream)


def pad64(x):
    return (x + 63) // 64 * 64


def split(x, nplanes):
    rows, cols = x.shape
    kpad = pad64(cols)
    ldo = nplanes * kpad
    out = torch.empty((rows + 63) // 64 * 64, ldo, dtype=torch.int16, device=dev)   # blocked layout: whole 64-row blocks
    capi.check(lib.s2vt_split_planes(nplanes, 0, vp(x.data_ptr()), i64(x.stride(0)), rows, cols, vp(out.data_ptr()),
                                     i64(ldo), kpad, rows, st), "split")
    return out, ldo, kpad


def timeit(fn, n=5):
    fn()
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for name, M, N, K in SHAPES:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev) * 0.05
    c32 = torch.empty(M, N, device=dev)
    c3 = torch.empty(M, N, device=dev)
    c1 = torch.empty(M, N, device=dev)
    for npl, c in ((3, c3), (1, c1)):
        pa, lda, Kp = split(a, npl)
        pb, ldb, _ = split(b, npl)

        def run(npl=npl, pa=pa, pb=pb, lda=lda, ldb=ldb, Kp=Kp, c=c):
            capi.check(lib.s2vt_gemm_bf16_nt(npl, M, N, Kp, vp(pa.data_ptr()), i64(lda), vp(pb.data_ptr()),
                                             i64(ldb), vp(c.data_ptr()), i64(N), vp(0), 0, vp(ws.data_ptr()),
                                             ctypes.c_size_t(ws.numel()), st), "gemm_bf16")
        t = timeit(run)
        tsplit = timeit(lambda: (split(a, npl), split(b, npl)), 3)
        print("%-7s planes=%d  M=%5d N=%5d K=%5d  %8.1f us  %6.1f TF-equivalent   (split of both operands %.1f us)" %
              (name, npl, M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e9, tsplit * 1e3), flush=True)

    def run32():
        capi.check(lib.s2vt_gemm_f32_splitk(1, 1, M, N, K, vp(a.data_ptr()), i64(K), vp(b.data_ptr()), i64(K),
                                            vp(c32.data_ptr()), i64(N), vp(0), 0, vp(ws.data_ptr()),
                                            ctypes.c_size_t(ws.numel()), st), "gemm_f32")
    t32 = timeit(run32)
    sel = torch.cat([torch.arange(0, min(M, 200)), torch.arange(max(M - 200, 0), M)]).unique().to(dev)   # head and tail tiles
    ref = (a[sel].double() @ b.double().t())
    scale = ref.abs().max().item()
    e32 = (c32[sel].double() - ref).abs().max().item() / scale
    e3 = (c3[sel].double() - ref).abs().max().item() / scale
    e1 = (c1[sel].double() - ref).abs().max().item() / scale
    print("%-7s fp32-MFMA %8.1f us %6.1f TF | max err / max|C|: fp32 %.2e  bf16x3 %.2e  bf16 %.2e" %
          (name, t32 * 1e3, 2.0 * M * N * K / t32 / 1e9, e32, e3, e1), flush=True)
