"""Where does a forward timestep launch spend its time?  Builds the library with -DS2VT_STAMPS (one wave per launch
records 100-MHz wall-clock stamps), runs one 159-step layer and prints the average phase boundaries.
usage: python tools/bench_step_stamps.py [B] [extra -D defines ...]     (GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from s2vt_video_caption_amd import build  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
defs = ["S2VT_STAMPS"] + sys.argv[2:]
H, T, NGX = 1000, 159, 80
dev = "cuda:0"
os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
path = build.build(defines=defs, out_path=os.path.join(ROOT, "gpurun_out", "variants", "lib_stamps_%d.so" % len(defs)))
ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
lib = ctypes.CDLL(path)
torch.manual_seed(0)
k = 1.0 / H ** 0.5
w_hh = ((torch.rand(4 * H, H) * 2 - 1) * k).to(dev)
bias = ((torch.rand(4 * H) * 2 - 1) * k).to(dev)
stash = torch.randn(T * B, 4 * H).to(dev)
h_all = torch.empty(T * B, H, device=dev)
c_all = torch.empty(T * B, H, device=dev)
vp = ctypes.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)
out = np.zeros((8192, 8), dtype=np.uint64)
for it in range(3):
    lib.s2vt_debug_stamps(vp(0), 0, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.s2vt_lstm_seq_fwd(T, B, H, vp(stash.data_ptr()), NGX, vp(bias.data_ptr()), vp(w_hh.data_ptr()),
                                 vp(h_all.data_ptr()), vp(c_all.data_ptr()), vp(stash.data_ptr()), st) == 0
    e1.record()
    torch.cuda.synchronize()
    n = lib.s2vt_debug_stamps(out.ctypes.data_as(vp), 8192, 0)
s = out[:n].astype(np.int64)
s = s[np.argsort(s[:, 0])]
print("B=%d: %d stamped launches, %.2f us per launch by events" % (B, n, e0.elapsed_time(e1) * 1e3 / T))
print("  (stamping atomic returned at %.2f us after entry: that much of the next line is the instrumentation)" % ((s[1:-1, 7] - s[1:-1, 0]) * 0.01).mean())
names = ["entry", "loads issued", "first chunk staged", "K loop done", "all waves done (barrier)", "partials reduced (barrier)",
         "outputs stored+drained"]
rel = (s[1:-1, :7] - s[1:-1, :1]) * 0.01     # us since entry (skip first launch: no recurrent term)
for i, nm in enumerate(names):
    print("  %-30s %6.2f us (median %6.2f)" % (nm, rel[:, i].mean(), np.median(rel[:, i])))
gap = (s[2:, 0] - s[1:-1, 6]) * 0.01
print("  exit -> next launch's entry    %6.2f us (median %6.2f)" % (gap.mean(), np.median(gap)))
print("  entry -> next entry            %6.2f us" % ((s[2:, 0] - s[1:-1, 0]) * 0.01).mean())
