"""Weight-gradient GEMM shapes of one train step in both operand forms: NT (k-major operands = the transposed copies round 3 made)
against TT (both operands read transposed from their row images).  usage: python tools/bench_gemm_tt.py [B] [planes]   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from s2vt_video_caption_amd import capi, ops  # noqa: E402

capi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L, F, H, V = 80, 4096, 1000, 12000
R, T = B * (L - 1), 2 * L - 1
SHAPES = [("dWo = dlogits^T h2", V, H, R), ("dWhh = dG^T h", 4 * H, H, (T - 1) * B), ("dWih2 = dG2^T h1", 4 * H, H, T * B),
          ("dWe = dG2^T emb", 4 * H, H, R), ("dWih1 = dG1^T x1", 4 * H, H, B * L), ("dWf = dx1^T feats", H, F, B * L)]
dev = "cuda:0"
ws = torch.empty(256 << 20, device=dev)
for name, M, N, K in SHAPES:
    xa = torch.randn(K, M, device=dev)
    xb = torch.randn(K, N, device=dev)
    pa_t, pb_t = ops.split_planes(xa, NP, transpose=True), ops.split_planes(xb, NP, transpose=True)      # [M][K], [N][K]
    pa_r, pb_r = ops.split_planes(xa, NP), ops.split_planes(xb, NP)                                      # row images
    del xa, xb
    c = torch.empty(M, N, device=dev)
    res = []
    for form in ("nt", "tt"):
        best = 1e9
        for it in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if form == "nt":
                ops.gemm_planes(pa_t, pb_t, M, N, nplanes=NP, out=c, splitk_ws=ws)
            else:
                ops.gemm_planes_tt(pa_r, pb_r, M, N, K, out=c, splitk_ws=ws, nplanes=NP)
            e1.record()
            torch.cuda.synchronize()
            if it:
                best = min(best, e0.elapsed_time(e1))
        res.append(best)
    print("%-22s M=%6d N=%6d K=%6d  NT %8.1f us  TT %8.1f us  (%.2fx)" % (name, M, N, K, res[0] * 1e3, res[1] * 1e3, res[1] / res[0]), flush=True)
    del pa_t, pb_t, pa_r, pb_r, c
