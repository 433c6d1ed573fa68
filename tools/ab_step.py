"""A/B timing of train-step variants in ONE process on ONE device (interleaved rounds; boxes of the pool differ by ~5 %,
so two bench.py calls on two boxes cannot rank two builds).

usage: python tools/ab_step.py [--batch 64] [--gemm-mode 3] [--rounds 5] [--steps 20] variant [variant ...]
  variant = comma-separated settings applied before its rounds:
     name=value     library option (s2vt_set_option), e.g. persist_x3_bwd=1
     crit=old|new   MaskCriterion as the round-4 composition of torch ops / the fused two-launch form
     opt=torch|hip  optimizer: torch.optim.Adam(fused=True) / the library's flat fused Adam (if built)
  e.g.  python tools/ab_step.py crit=old crit=new  "crit=new,persist_x3_bwd=1"
Prints per variant the median / min ms per step over the rounds."""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import S2VTModel  # noqa: E402
import utils  # noqa: E402
from s2vt_video_caption_amd import capi, dp, synth  # noqa: E402
from s2vt_video_caption_amd import functional as F_  # noqa: E402


class OldCriterion(torch.nn.Module):
    """round 4's MaskCriterion: the mean CE from the HIP kernels, the three reference lines around it as torch ops"""

    def forward(self, logits, target, mask):
        mean_ce = F_.mean_cross_entropy(logits, target)
        weights = mask[:, 1:].reshape(-1)
        return (mean_ce * weights).sum() / weights.sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--gemm-mode", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--hidden", type=int, default=1000)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = capi.load()
    lib.s2vt_set_gemm_mode(a.gemm_mode)
    L, F, H, E, V = 80, 4096, a.hidden, a.hidden, 12000
    B = a.batch
    sd = synth.make_state_dict(V, F, H, E, seed=0)
    model = S2VTModel.S2VT(V, F, L, dim_hid=H, dim_embed=E)
    model.load_state_dict(sd)
    model.to(dev)
    feats, caps, mask = (t.to(dev) for t in synth.make_batch(B, L, F, V, seed=1234))
    defaults = {}

    def apply(variant):
        st = {"crit": "new", "opt": "torch"}
        for name, prev in defaults.items():          # back to the library defaults first
            lib.s2vt_set_option(name.encode(), prev)
        for kv in variant.split(","):
            if not kv:
                continue
            k, v = kv.split("=")
            if k in st:
                st[k] = v
            else:
                prev = lib.s2vt_set_option(k.encode(), int(v))
                if prev == -(2 ** 31):
                    raise SystemExit("unknown library option %r" % k)
                defaults.setdefault(k, prev)
        crit = OldCriterion() if st["crit"] == "old" else utils.MaskCriterion()
        if st["opt"] == "hip":
            from s2vt_video_caption_amd import optim
            opt = optim.FlatAdam(model, lr=1e-4)
        else:
            F_.set_grad_sink(model, None)
            for p in model.parameters():
                p.grad = None
            opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
        return crit, opt

    results = {v: [] for v in a.variants}
    for rnd in range(a.rounds + 1):                   # round 0 warms every variant up (lazy initialisations, allocator)
        for v in a.variants:
            crit, opt = apply(v)
            for _ in range(3):
                dp.train_step(model, crit, opt, feats, caps, mask, None)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                dp.train_step(model, crit, opt, feats, caps, mask, None)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            capi.check_async_error()
            if rnd:
                results[v].append(ms)
    for v in a.variants:
        r = results[v]
        print("%-40s median %.3f  min %.3f  max %.3f ms/step  (%d rounds x %d steps)" % (v, statistics.median(r), min(r), max(r), len(r), a.steps))


if __name__ == "__main__":
    main()
