"""Soak run: N optimisation steps back to back on one synthetic batch per configuration; reports step-time spread (HIP events
per step), the loss at intervals, and checks the asynchronous error ring (IndexError / hand-off timeouts of the persistent
kernels) at the end.  usage: python tools/soak.py [steps] [gc=on|off|freeze]   (GPU box)
gc=off / gc=freeze: the interpreter's cyclic collector disabled / the start-up objects frozen out of it (round 4's soak showed one
95-117 ms step in 3000: the five slowest steps are printed with their index and the HOST time of their enqueue, which tells a
host stall (collector, allocator) from a device one)."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import S2VTModel  # noqa: E402
import utils  # noqa: E402
from s2vt_video_caption_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
GC = ([a[3:] for a in sys.argv[2:] if a.startswith("gc=")] or ["on"])[0]
from s2vt_video_caption_amd.optim import FlatAdam  # noqa: E402
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
    opt = FlatAdam(m, lr=1e-4)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    losses = []
    host = [0.0] * N
    gcs = []
    gc.callbacks.append(lambda phase, info: gcs.append((cur[0], info.get("generation"))) if phase == "start" else None)
    cur = [0]
    if GC == "off":
        gc.disable()
    elif GC == "freeze":
        gc.collect()
        gc.freeze()
    ev[0].record()
    for s in range(N):
        cur[0] = s
        h0 = time.perf_counter()
        opt.zero_grad()
        loss = crit(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
        loss.backward()
        opt.step()
        ev[s + 1].record()
        host[s] = (time.perf_counter() - h0) * 1e3
        if s % (N // 8) == 0 or s == N - 1:
            losses.append((s, float(loss.detach())))
            print("  %s step %d loss %.4f" % (name, s, losses[-1][1]), flush=True)
    torch.cuda.synchronize()
    capi.check_async_error()
    gc.callbacks.pop()
    gc.enable()
    gc.unfreeze()
    per = [(ev[i].elapsed_time(ev[i + 1]), i) for i in range(10, N)]
    ts = sorted(t for t, _ in per)
    gen2 = sorted({i for i, g in gcs if g == 2})
    print("  gc=%s: %d collections (generation 2 at steps %s); five slowest steps: %s" %
          (GC, len(gcs), gen2[:12], ", ".join("step %d: %.2f ms on the device, host enqueue %.2f ms%s" %
                                              (i, t, host[i], " (generation-2 collection)" if i in gen2 else "")
                                              for t, i in sorted(per, reverse=True)[:5])), flush=True)
    assert all(l == l and abs(l) < 1e4 for _, l in losses)
    assert losses[-1][1] < losses[0][1]
    print("%s: %d steps, step time ms min %.3f median %.3f p99 %.3f max %.3f; loss %.4f -> %.4f; no asynchronous error" %
          (name, N, ts[0], ts[len(ts) // 2], ts[int(len(ts) * 0.99)], ts[-1], losses[0][1], losses[-1][1]), flush=True)
    del m, opt, feats, caps, mask
    torch.cuda.empty_cache()
