"""Train step and greedy decode of the drop-in attention_baseline.Att_Baseline at the reference's default
widths on synthetic data: the per-op C-ABI composition (split-precision plane GEMMs, launch-per-timestep recurrences).
usage: python tools/bench_att.py [B]   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import attention_baseline  # noqa: E402
import utils  # noqa: E402
from s2vt_video_caption_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L, F, H, E, V = 80, 4096, 500, 500, 12000
dev = "cuda:0"
torch.manual_seed(0)
m = attention_baseline.Att_Baseline(V, F, L, dim_hid=H, dim_embed=E).to(dev)
feats, caps, mask = (t.to(dev) for t in synth.make_batch(B, L, F, V, seed=5))
crit = utils.MaskCriterion()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)


def step():
    opt.zero_grad()
    loss = crit(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("Att_Baseline train step B=%d L=%d F=%d H=E=%d V=%d: %.2f ms = %.0f frames/s (loss %.4f)" % (B, L, F, H, V, dt * 1e3, B * L / dt, float(loss)))
m.eval()
with torch.no_grad():
    m(feats, mode="test")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ids = m(feats, mode="test")
    torch.cuda.synchronize()
print("Att_Baseline greedy decode B=%d: %.2f ms per call = %.0f captions/s" % (B, (time.perf_counter() - t0) / 5 * 1e3, B * 5 / (time.perf_counter() - t0)))
