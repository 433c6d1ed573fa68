"""Per-shape time of the batched GEMMs of one train step (every call the schedule makes, by role), for the bf16 (planes=1,
gemm_b1) or split-precision (planes=3, gemm_x3) kernel.  usage: python tools/bench_gemm_shapes.py [B] [planes]   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import capi, ops  # noqa: E402

capi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L, F, H, V, BLK = 80, 4096, 1000, 12000, 32
R = B * (L - 1)
T = 2 * L - 1
SHAPES = [("x1 = feats Wf^T", B * L, H, F), ("gx1 = x1 Wih1^T", B * L, 4 * H, H), ("gx2 block = h1 Wv^T", BLK * B, 4 * H, H),
          ("gxe = emb We^T", R, 4 * H, H), ("logits = h2 Wo^T", R, V, H), ("dh2 = dlogits Wo", R, H, V),
          ("dWo = dlogits^T h2", V, H, R), ("dh1 block = dG2 Wv", BLK * B, H, 4 * H), ("dWhh = dG^T h", 4 * H, H, T * B),
          ("dWe = dG2^T emb", 4 * H, H, R), ("demb = dG2 We", R, H, 4 * H), ("dx1 = dG1 Wih1", B * L, H, 4 * H),
          ("dWih1 = dG1^T x1", 4 * H, H, B * L), ("dWf = dx1^T feats", H, F, B * L)]
dev = "cuda:0"
ws = torch.empty(256 << 20, device=dev)
tot = 0.0
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    pa, pb = ops.split_planes(a, NP), ops.split_planes(b, NP)
    del a, b
    c = torch.empty(M, N, device=dev)
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm_planes(pa, pb, M, N, nplanes=NP, out=c, splitk_ws=ws); e1.record()
        torch.cuda.synchronize()
        if it:
            best = min(best, e0.elapsed_time(e1))
    tot += best
    print("%-22s M=%6d N=%6d K=%6d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9), flush=True)
    del pa, pb, c
print("sum of one call each: %.3f ms" % tot)
