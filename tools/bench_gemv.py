"""The launch-per-timestep forward step at B <= 8: gate-GEMV kernel (csrc/lstm_gemv.hip, option gemv = 1) against the 16-row
fp32-MFMA tile kernel (option gemv = 0) - per-launch time of a chain of dependent steps (the decode's regime: step t + 1 reads
step t's h) with and without the embedded-word segment, and a whole greedy decode at B = 1 / 8 / 10 in gemm mode 0 (every step
one launch) and in the default plane mode (batch padded to 64: persistent encode phase, plane-path argmax).
usage: python tools/bench_gemv.py        (GPU box; profiles/round5_gemv_small_batch.txt)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import S2VTModel  # noqa: E402
from s2vt_video_caption_amd import capi, ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
lib = capi.load()
H = E = 1000
V = 12000
g = torch.Generator().manual_seed(0)
w_hh = ((torch.rand(4 * H, H, generator=g) * 2 - 1) * H ** -0.5).to(dev)
w_ih = ((torch.rand(4 * H, E + H, generator=g) * 2 - 1) * H ** -0.5).to(dev)
emb = torch.randn(V, E, generator=g).to(dev)


def chain(B, token, n=200):
    gx = torch.randn(B, 4 * H, generator=g).to(dev)
    h = torch.zeros(B, H, device=dev)
    c = torch.zeros(B, H, device=dev)
    tok = torch.randint(0, V, (B,), generator=g, dtype=torch.int32).to(dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if token:
                h, c = ops.lstm_step_fwd_token(gx, w_hh, h, c, emb, w_ih, tok=tok)
            else:
                h, c = ops.lstm_step_fwd(gx, None, w_hh, h, c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    return dt * 1e6


print("per-launch time of a dependent chain of timesteps, H = E = %d (us)" % H)
print("%4s %12s %12s %12s %12s" % ("B", "gemv", "tile", "gemv+emb", "tile+emb"))
for B in (1, 2, 4, 8):
    row = []
    for token in (False, True):
        for on in (1, 0):
            lib.s2vt_set_option(b"gemv", 2 if on else 0)
            row.append(chain(B, token))
    print("%4d %12.2f %12.2f %12.2f %12.2f" % (B, row[0], row[1], row[2], row[3]))
lib.s2vt_set_option(b"gemv", 1)

L, F = 80, 4096
sd = synth.make_state_dict(V, F, H, E, seed=0)
m = S2VTModel.S2VT(V, F, L, dim_hid=H, dim_embed=E)
m.load_state_dict(sd)
m.to(dev).eval()
print("\ngreedy decode, ms per call (79 decode steps + 159 / 80 encode steps)")
print("%4s %22s %22s %22s" % ("B", "mode 0, gemv", "mode 0, tile kernel", "plane mode (padded to 64)"))
for B in (1, 2, 3, 4, 6, 8, 10, 16, 24, 32, 48, 64):
    feats = synth.make_batch(B, L, F, V, seed=3)[0].to(dev)
    res = []
    for mode, on in ((0, 1), (0, 0), (3, 1)):
        lib.s2vt_set_gemm_mode(mode)
        lib.s2vt_set_option(b"gemv", 2 if on else 0)
        with torch.no_grad():
            for _ in range(2):
                ids = m(feats, mode="test")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ids = m(feats, mode="test")
            torch.cuda.synchronize()
        res.append(((time.perf_counter() - t0) / 5 * 1e3, ids.cpu()))
    same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], res[2][1])       # (random-init weights: near-ties may differ)
    print("%4d %22.3f %22.3f %22.3f   ids identical on all three: %s" % (B, res[0][0], res[1][0], res[2][0], same))
lib.s2vt_set_gemm_mode(3)
lib.s2vt_set_option(b"gemv", 1)
capi.check_async_error()
