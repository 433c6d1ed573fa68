"""Time the bf16 layer forward: one launch per timestep vs the persistent kernel (tools, GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import s2vt_video_caption_amd  # noqa
from s2vt_video_caption_amd import build, capi, ops

build.build()
capi.load()
DEV = "cuda:0"
T, H = 159, 1000
for B in (256, 128, 64):
    g = torch.Generator().manual_seed(1)
    gx = torch.randn(T * B, 4 * H, generator=g).to(DEV)
    bias = (torch.randn(4 * H, generator=g) * 0.3).to(DEV)
    w = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
    for persistent, block in ((False, 0), (True, 0), (True, 32)):
        lib = capi.load()
        for rep in range(3):
            lib.s2vt_prof_reset()
            lib.s2vt_prof_enable(1)
            ops.lstm_seq_fwd_bf16(gx, 80, bias, w, T, B, H, persistent=persistent, block=block)
            torch.cuda.synchronize()
            lib.s2vt_prof_enable(0)
            ms, n = capi.prof_read(1)
            dt = ms * 1e-3
        # bytes of the SURVEY 8(d) accounting for a vid-shaped layer (I = H, bf16 operands, train)
        s = 2
        by = s * 4 * H * 2 * H + 4 * 8 * H + s * B * 2 * H + 4 * B * H + s * B * H + 4 * B * H + s * B * 4 * H
        if persistent and block == 32:      # both layers in one launch: time per PAIR of layer steps
            for rep in range(3):
                lib.s2vt_prof_reset()
                lib.s2vt_prof_enable(1)
                ops.lstm_seq_fwd_bf16_pair(gx, gx, 80, bias, bias, w, w, T, B, H, block=block)
                torch.cuda.synchronize()
                lib.s2vt_prof_enable(0)
                ms2, n2 = capi.prof_read(1)
            print("B=%d two layers in one launch, block=%d: %.3f ms, %.2f us per step PAIR, %.2f TB/s (8d bytes of a vid+word pair)" %
                  (B, block, ms2, ms2 / T * 1e3, (by + s * 4 * H * H + s * B * H) / (ms2 * 1e-3 / T) / 1e12), flush=True)
        print("B=%d persistent=%s block=%d: %.3f ms per layer pass (HIP events), %.2f us/step, %.2f TB/s (8d bytes)" %
              (B, persistent, block, dt * 1e3, dt / T * 1e6, by / (dt / T) / 1e12), flush=True)

# ---- BPTT: one launch per timestep vs persistent (single layer, and two layers in one launch)
print("---- BPTT", flush=True)
for B in (256, 64):
    g = torch.Generator().manual_seed(2)
    w = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
    gates = torch.sigmoid(torch.randn(T * B, 4 * H, generator=g)).to(DEV)
    c_all = (torch.randn(T * B, H, generator=g) * 0.7).to(DEV)
    dh = (torch.randn(T * B, H, generator=g) * 0.1).to(DEV)
    lib = capi.load()
    for name, fn in (("per-step launches", lambda: ops.lstm_seq_bwd_bf16(w, dh, 0, c_all, gates, T, B, H, persistent=False)),
                     ("persistent block=32", lambda: ops.lstm_seq_bwd_bf16(w, dh, 0, c_all, gates, T, B, H, persistent=True, block=32)),
                     ("persistent two layers one launch block=32",
                      lambda: ops.lstm_seq_bwd_bf16_pair(w, w, dh, dh, 0, c_all, c_all, gates, gates, T, B, H, block=32))):
        for rep in range(3):
            lib.s2vt_prof_reset()
            lib.s2vt_prof_enable(1)
            fn()
            torch.cuda.synchronize()
            lib.s2vt_prof_enable(0)
            ms, n = capi.prof_read(2)
        print("B=%d BPTT %s: %.3f ms, %.2f us per layer step (%d layer steps)" % (B, name, ms, ms * 1e3 / max(n, 1), n), flush=True)

# ---- fp32 (config 2 arithmetic): launch per timestep vs persistent
print("---- fp32", flush=True)
for B in (64, 128):
    g = torch.Generator().manual_seed(3)
    gx = torch.randn(80 * B, 4 * H, generator=g).to(DEV)
    bias = (torch.randn(4 * H, generator=g) * 0.3).to(DEV)
    w = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(DEV)
    gates = torch.sigmoid(torch.randn(T * B, 4 * H, generator=g)).to(DEV)
    c_all = (torch.randn(T * B, H, generator=g) * 0.7).to(DEV)
    dh = (torch.randn(T * B, H, generator=g) * 0.1).to(DEV)
    lib = capi.load()
    runs = (("fwd per-step launches", 1, lambda: ops.lstm_seq_fwd(T, B, gx, 80, bias, w, want_stash=True)),
            ("fwd persistent block=32", 1, lambda: ops.lstm_seq_fwd_persist(T, B, gx, 80, bias, w, block=32)),
            ("fwd persistent two layers one launch", 1, lambda: ops.lstm_seq_fwd_persist(T, B, gx, 80, bias, w, block=32, second=(gx, bias, w))),
            ("bwd per-step launches", 2, lambda: ops.lstm_seq_bwd(T, B, w, dh, 0, c_all, gates.clone())),
            ("bwd persistent block=32", 2, lambda: ops.lstm_seq_bwd_persist(T, B, w, dh, 0, c_all, gates, block=32)),
            ("bwd persistent two layers one launch", 2, lambda: ops.lstm_seq_bwd_persist(T, B, w, dh, 0, c_all, gates, block=32,
                                                                                       second=(w, dh, c_all, gates))))
    for name, kind, fn in runs:
        for rep in range(3):
            lib.s2vt_prof_reset()
            lib.s2vt_prof_enable(1)
            fn()
            torch.cuda.synchronize()
            lib.s2vt_prof_enable(0)
            ms, n = capi.prof_read(kind)
        print("B=%d fp32 %s: %.3f ms, %.2f us per layer step (%d layer steps)" % (B, name, ms, ms * 1e3 / max(n, 1), n), flush=True)
