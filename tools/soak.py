"""Soak run: N optimisation steps back to back on one synthetic batch per configuration; reports step-time spread (HIP events
per step), the loss at intervals, and checks the asynchronous error ring (IndexError / hand-off timeouts of the persistent
kernels) at the end.  usage: python tools/soak.py [steps]   (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import S2VTModel  # noqa: E402
import utils  # noqa: E402
from s2vt_video_caption_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lib = capi.load()
dev = "cuda:0"
for name, B, mode in (("configs[1] B=64 fp32-equivalent", 64, 3), ("configs[2] B=256 bf16, persistent recurrence", 256, 1),
                      ("configs[3] shard B=128 fp32-equivalent", 128, 3)):
    lib.s2vt_set_gemm_mode(mode)
    d = synth.CONFIGS["c2"]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=3)
    feats, caps, mask = (t.to(dev) for t in synth.make_batch(B, d["L"], d["F"], d["V"], seed=4))
    m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"]).to(dev)
    m.load_state_dict(sd)
    m.train()
    crit = utils.MaskCriterion()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    losses = []
    ev[0].record()
    for s in range(N):
        opt.zero_grad()
        loss = crit(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
        loss.backward()
        opt.step()
        ev[s + 1].record()
        if s % (N // 8) == 0 or s == N - 1:
            losses.append((s, float(loss.detach())))
            print("  %s step %d loss %.4f" % (name, s, losses[-1][1]), flush=True)
    torch.cuda.synchronize()
    capi.check_async_error()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10, N))
    assert all(l == l and abs(l) < 1e4 for _, l in losses)
    assert losses[-1][1] < losses[0][1]
    print("%s: %d steps, step time ms min %.3f median %.3f p99 %.3f max %.3f; loss %.4f -> %.4f; no asynchronous error" %
          (name, N, ts[0], ts[len(ts) // 2], ts[int(len(ts) * 0.99)], ts[-1], losses[0][1], losses[-1][1]), flush=True)
    del m, opt, feats, caps, mask
    torch.cuda.empty_cache()
