"""Batches that are not multiples of 64: the train step and the greedy decode on the plane path with the batch padded inside the
library's workspace (gemm mode 3, option pad_min_batch = 1: every ragged batch pads) against the launch-per-timestep fp32-MFMA
path at the batch as it is (gemm mode 0) - where is the crossover?  The library pads above the size this table shows
(profiles/round5_ragged_batches.txt).   usage: python tools/bench_ragged.py [H ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import S2VTModel  # noqa: E402
import utils  # noqa: E402
from s2vt_video_caption_amd import capi, dp, synth  # noqa: E402

dev = torch.device("cuda", 0)
lib = capi.load()
L, F, V = 80, 4096, 12000


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    capi.check_async_error()
    return (time.perf_counter() - t0) / n * 1e3


for H in [int(x) for x in sys.argv[1:]] or [512, 1000]:
    sd = synth.make_state_dict(V, F, H, H, seed=1)
    m = S2VTModel.S2VT(V, F, L, dim_hid=H, dim_embed=H)
    m.load_state_dict(sd)
    m.to(dev)
    crit = utils.MaskCriterion()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
    print("H = E = %d, L = %d, F = %d, V = %d: ms per train step | ms per greedy decode" % (H, L, F, V))
    print("%5s %16s %16s | %16s %16s" % ("B", "padded, plane", "as is, fp32-MFMA", "padded, plane", "as is, fp32-MFMA"))
    for B in (4, 8, 16, 24, 32, 48, 64, 100, 128):
        batch = tuple(t.to(dev) for t in synth.make_batch(B, L, F, V, seed=B))
        row = []
        for mode in (3, 0):
            lib.s2vt_set_gemm_mode(mode)
            lib.s2vt_set_option(b"pad_min_batch", 1)
            m.train()
            row.append(timed(lambda: dp.train_step(m, crit, opt, batch[0], batch[1], batch[2], None), 10))
        for mode in (3, 0):
            lib.s2vt_set_gemm_mode(mode)
            m.eval()

            def dec():
                with torch.no_grad():
                    m(batch[0], mode="test")
            row.append(timed(dec, 5))
        print("%5d %16.3f %16.3f | %16.3f %16.3f" % (B, row[0], row[1], row[2], row[3]))
    lib.s2vt_set_gemm_mode(3)
    del m, opt
