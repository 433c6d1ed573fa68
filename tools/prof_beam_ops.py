"""Micro-timing of the per-depth device ops of the batched beam search at C5 sizes (R = B*beam rows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from s2vt_video_caption_amd import ops, capi
capi.load()
dev = "cuda:0"
R, H, E, V = 640, 1000, 1000, 12000
torch.manual_seed(0)
wh = torch.randn(R, H, device=dev); w_o = torch.randn(V, H, device=dev) * 0.03; b_o = torch.randn(V, device=dev)
w_e = torch.randn(4 * H, E, device=dev) * 0.03; emb = torch.randn(V, E, device=dev); bsum = torch.randn(4 * H, device=dev)
w_hh = torch.randn(4 * H, H, device=dev) * 0.03
h = torch.randn(R, H, device=dev); c = torch.randn(R, H, device=dev)
idx_np = np.random.randint(0, R, size=R).astype(np.int64)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


logits = ops.gemm(wh, w_o, bias=b_o)
logp = torch.log_softmax(logits, dim=1)
top = logp.topk(20, dim=1).indices
print("as_tensor H2D (640 int64) x3 : %.3f ms" % t(lambda: [torch.as_tensor(idx_np, device=dev) for _ in range(3)]))
idx = torch.as_tensor(idx_np, device=dev)
print("3 gathers [640,1000]         : %.3f ms" % t(lambda: (h[idx], c[idx], emb[idx])))
print("gemm 640x4000x1000 (fp32 MFMA): %.3f ms" % t(lambda: ops.gemm(wh, w_e, bias=bsum)))
gx = ops.gemm(wh, w_e, bias=bsum)
print("lstm_step_fwd R=640          : %.3f ms" % t(lambda: ops.lstm_step_fwd(gx, None, w_hh, h, c)))
print("logits gemm 640x12000x1000   : %.3f ms" % t(lambda: ops.gemm(wh, w_o, bias=b_o)))
print("log_softmax [640,12000]      : %.3f ms" % t(lambda: torch.log_softmax(logits, dim=1)))
print("topk(20)                     : %.3f ms" % t(lambda: logp.topk(20, dim=1)))
print("sort idx + gather + cat      : %.3f ms" % t(lambda: torch.cat([logp.gather(1, top.sort(dim=1).values), top.to(torch.float32)], dim=1)))
both = torch.cat([logp.gather(1, top), top.to(torch.float32)], dim=1)
print(".cpu() of [640,40]           : %.3f ms" % t(lambda: both.cpu()))
