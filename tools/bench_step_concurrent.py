"""Does running two independent LSTM sequences on two HIP streams overlap?  (feasibility check for the
layer-pipelined driver).  usage: python tools/bench_step_concurrent.py [variant defines...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import build  # noqa: E402

B, H, T, NGX = 64, 1000, 159, 80
dev = "cuda:0"
vp = ctypes.c_void_p
os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
for name, defs in (("w4_pf2", ["S2VT_NWAVE=4", "S2VT_PF=2"]), ("w4_pf1", ["S2VT_NWAVE=4", "S2VT_PF=1"]),
                   ("w8_pf2", ["S2VT_NWAVE=8", "S2VT_PF=2"])):
    path = build.build(defines=defs, out_path=os.path.join(ROOT, "gpurun_out", "variants", "lib_%s.so" % name))
    ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(path)
    k = 1.0 / H ** 0.5
    bufs = []
    for i in range(2):
        torch.manual_seed(i)
        bufs.append(dict(w=((torch.rand(4 * H, H) * 2 - 1) * k).to(dev), bias=torch.zeros(4 * H, device=dev),
                         gx=torch.randn(T * B, 4 * H).to(dev), h=torch.empty(T * B, H, device=dev),
                         c=torch.empty(T * B, H, device=dev)))
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(buf, stream):
        rc = lib.s2vt_lstm_seq_fwd(T, B, H, vp(buf["gx"].data_ptr()), NGX, vp(buf["bias"].data_ptr()),
                                   vp(buf["w"].data_ptr()), vp(buf["h"].data_ptr()), vp(buf["c"].data_ptr()),
                                   vp(buf["gx"].data_ptr()), vp(stream.cuda_stream))
        assert rc == 0

    def timeit(fn):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            s0.wait_event(e0); s1.wait_event(e0)
            fn()
            d0, d1 = torch.cuda.Event(), torch.cuda.Event()
            d0.record(s0); d1.record(s1)
            torch.cuda.current_stream().wait_event(d0); torch.cuda.current_stream().wait_event(d1)
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    one = timeit(lambda: run(bufs[0], s0))
    seq = timeit(lambda: (run(bufs[0], s0), run(bufs[1], s0)))
    par = timeit(lambda: (run(bufs[0], s0), run(bufs[1], s1)))
    print("%-8s one seq %.3f ms | two back-to-back %.3f ms | two on two streams %.3f ms (%.2fx of one)" %
          (name, one, seq, par, par / one), flush=True)
