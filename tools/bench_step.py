"""Micro-benchmark of the fused LSTM timestep kernels: builds library variants (-D defines) and times
s2vt_lstm_seq_fwd / s2vt_lstm_seq_bwd (159 steps, zero-padded tail) with torch events.
usage: python tools/bench_step.py [B] [H]        (run on the GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import build  # noqa: E402

VARIANTS = {
    "f4b8_kc64": [],
    "f8b8_kc32": ["S2VT_NWAVE_FWD=8", "S2VT_KC=32"],
    "f4b8_kc32": ["S2VT_KC=32"],
    "f8b8_kc32_pf1": ["S2VT_NWAVE_FWD=8", "S2VT_KC=32", "S2VT_PF=1"],
    "f8b8_kc32_pf3": ["S2VT_NWAVE_FWD=8", "S2VT_KC=32", "S2VT_PF=3"],
}
if os.environ.get("VARIANTS"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["VARIANTS"].split(",")}

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
T, NGX = 159, 80
dev = "cuda:0"
torch.manual_seed(0)
k = 1.0 / H ** 0.5
w_hh = ((torch.rand(4 * H, H) * 2 - 1) * k).to(dev)
bias = ((torch.rand(4 * H) * 2 - 1) * k).to(dev)
gx0 = torch.randn(NGX * B, 4 * H).to(dev)
dh = (torch.randn(T * B, H) * 0.01).to(dev)
vp = ctypes.c_void_p


def ptr(t):
    return vp(t.data_ptr())


os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
for name, defs in VARIANTS.items():
    path = build.build(defines=defs, out_path=os.path.join(ROOT, "gpurun_out", "variants", "lib_%s.so" % name))
    rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(path)
    h_all = torch.empty(T * B, H, device=dev)
    c_all = torch.empty(T * B, H, device=dev)
    stash = torch.empty(T * B, 4 * H, device=dev)
    wt = torch.empty(H, 4 * H, device=dev)
    dc = torch.empty(B, H, device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)

    def fwd():
        stash[:NGX * B].copy_(gx0)
        rc = lib.s2vt_lstm_seq_fwd(T, B, H, ptr(stash), NGX, ptr(bias), ptr(w_hh), ptr(h_all), ptr(c_all), ptr(stash), st)
        assert rc == 0

    def bwd():
        rc = lib.s2vt_lstm_seq_bwd(T, B, H, ptr(w_hh), ptr(dh), 0, ptr(c_all), ptr(stash), ptr(wt), ptr(dc), st)
        assert rc == 0

    res = {}
    for label, fn in (("fwd", fwd), ("bwd", bwd)):
        ts = []
        for it in range(4):
            if label == "bwd":
                fwd()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if label == "fwd":
                stash[:NGX * B].copy_(gx0)
                torch.cuda.synchronize()
                e0.record()
                lib.s2vt_lstm_seq_fwd(T, B, H, ptr(stash), NGX, ptr(bias), ptr(w_hh), ptr(h_all), ptr(c_all), ptr(stash), st)
            else:
                e0.record()
                bwd()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / T)
        res[label] = min(ts[1:])
    print("%-16s B=%d H=%d  fwd %.2f us/step   bwd %.2f us/step  (loop-bracketed, incl. launch gaps)" %
          (name, B, H, res["fwd"], res["bwd"]), flush=True)
