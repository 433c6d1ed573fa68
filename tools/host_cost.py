"""Host-side enqueue cost of one train step, by phase (no GPU work is waited for inside a phase; a synchronise between
phases empties the queues so that back-pressure never shows up as host time).

  python tools/host_cost.py [--batch 64] [--gemm-mode 3] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--gemm-mode", type=int, default=None)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import S2VTModel
    import utils
    from s2vt_video_caption_amd import capi, synth
    lib = capi.load()
    if args.gemm_mode is not None:
        lib.s2vt_set_gemm_mode(args.gemm_mode)
    dev = torch.device("cuda", 0)
    L, F, H, E, V = 80, 4096, 1000, 1000, 12000
    B = args.batch
    model = S2VTModel.S2VT(V, F, L, dim_hid=H, dim_embed=E)
    model.load_state_dict(synth.make_state_dict(V, F, H, E, seed=0))
    model.to(dev)
    crit = utils.MaskCriterion()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    feats, caps, mask = (t.to(dev) for t in synth.make_batch(B, L, F, V, seed=1234))
    phases = ["zero_grad", "forward", "criterion", "backward", "adam"]
    acc = {k: 0.0 for k in phases}
    wall = 0.0

    def step(timed):
        nonlocal wall
        def ph(name, fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            if timed:
                acc[name] += time.perf_counter() - t0
            return r
        ph("zero_grad", lambda: opt.zero_grad())
        probs = ph("forward", lambda: model(feats, targets=caps[:, :-1], mode="train"))
        loss = ph("criterion", lambda: crit(probs, caps, mask))
        ph("backward", lambda: loss.backward())
        ph("adam", lambda: opt.step())

    for _ in range(3):
        step(False)
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    # whole step, no synchronisation inside: host time to enqueue vs wall time to execute
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.zero_grad()
        loss = crit(model(feats, targets=caps[:, :-1], mode="train"), caps, mask)
        loss.backward()
        opt.step()
    host = (time.perf_counter() - t0) / args.steps
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps
    print("B=%d gemm_mode=%d: host enqueue per phase (queues empty at phase start), ms:" % (B, lib.s2vt_set_gemm_mode(-1)))
    for k in phases:
        print("  %-10s %7.3f" % (k, acc[k] / args.steps * 1e3))
    print("  sum        %7.3f" % (sum(acc.values()) / args.steps * 1e3))
    print("free-running: host %.3f ms/step, wall %.3f ms/step" % (host * 1e3, wall * 1e3))


if __name__ == "__main__":
    main()
