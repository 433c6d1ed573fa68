"""Can a long batched GEMM (stream B) overlap a chain of LSTM timestep kernels (stream A)?
Measures: steps alone, GEMM alone, both concurrently, for library variants.  (GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import build  # noqa: E402

VARIANTS = {"base": []}
B, H, T = 64, 1000, 159
dev = "cuda:0"
vp, i64 = ctypes.c_void_p, ctypes.c_int64
os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
for name, defs in VARIANTS.items():
    lib = ctypes.CDLL(build.build(defines=defs, out_path=os.path.join(ROOT, "gpurun_out", "variants", "libo_%s.so" % name)))
    torch.manual_seed(0)
    k = 1.0 / H ** 0.5
    w = ((torch.rand(4 * H, H) * 2 - 1) * k).to(dev)
    bias = torch.zeros(4 * H, device=dev)
    gx = torch.randn(T * B, 4 * H, device=dev)
    h = torch.empty(T * B, H, device=dev); c = torch.empty(T * B, H, device=dev)
    M, N, K = 10176, 4000, 3072
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    pa = torch.empty((M + 63) // 64 * 64, 3 * K, dtype=torch.int16, device=dev); pb = torch.empty((N + 63) // 64 * 64, 3 * K, dtype=torch.int16, device=dev)
    cc = torch.empty(M, N, device=dev)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    st0 = vp(torch.cuda.current_stream().cuda_stream)
    lib.s2vt_split_planes(3, 0, vp(a.data_ptr()), i64(K), M, K, vp(pa.data_ptr()), i64(3 * K), K, M, st0)
    lib.s2vt_split_planes(3, 0, vp(b.data_ptr()), i64(K), N, K, vp(pb.data_ptr()), i64(3 * K), K, N, st0)
    torch.cuda.synchronize()

    def steps(stream):
        assert lib.s2vt_lstm_seq_fwd(T, B, H, vp(gx.data_ptr()), T, vp(bias.data_ptr()), vp(w.data_ptr()), vp(h.data_ptr()),
                                     vp(c.data_ptr()), vp(0), vp(stream.cuda_stream)) == 0

    def gemm(stream):
        assert lib.s2vt_gemm_bf16_nt(3, M, N, K, vp(pa.data_ptr()), i64(3 * K), vp(pb.data_ptr()), i64(3 * K), vp(cc.data_ptr()),
                                     i64(N), vp(0), 0, vp(0), ctypes.c_size_t(0), vp(stream.cuda_stream)) == 0

    def timeit(fn):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            e0.record(cur); s0.wait_event(e0); s1.wait_event(e0)
            fn()
            d0, d1 = torch.cuda.Event(), torch.cuda.Event()
            d0.record(s0); d1.record(s1); cur.wait_event(d0); cur.wait_event(d1)
            e1.record(cur)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    ts = timeit(lambda: steps(s1))
    tg = timeit(lambda: gemm(s0))
    tb = timeit(lambda: (gemm(s0), steps(s1)))
    tb2 = timeit(lambda: (steps(s1), gemm(s0)))
    print("%-12s steps alone %.3f ms | gemm alone %.3f ms | concurrent %.3f / %.3f ms (sum %.3f)" %
          (name, ts, tg, tb, tb2, ts + tg), flush=True)
