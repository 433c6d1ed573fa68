"""Where does the data-parallel wrapper spend time on ONE rank (RCCL 1-rank group)?  (GPU box, under torchrun)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import S2VTModel
import utils
from s2vt_video_caption_amd import dp, synth

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29512")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
L, F, H, E, V, B = 80, 4096, 1000, 1000, 12000, 64
m = S2VTModel.S2VT(V, F, L, dim_hid=H, dim_embed=E)
m.load_state_dict(synth.make_state_dict(V, F, H, E, seed=0))
m.to(dev)
crit = utils.MaskCriterion()
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
feats, caps, mask = [t.to(dev) for t in synth.make_batch(B, L, F, V, seed=1)]


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def step_plain():
    opt.zero_grad()
    loss = crit(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
    loss.backward()
    opt.step()


print("plain step BEFORE init_process_group  %.3f ms" % timed(step_plain))
dist.init_process_group(backend="nccl", device_id=dev)
print("plain step after init_process_group   %.3f ms" % timed(step_plain))
dist.all_reduce(torch.zeros(4, device=dev))
torch.cuda.synchronize()
print("plain step after first RCCL collective %.3f ms" % timed(step_plain))
red = dp.FlatGradAllReducer(m.parameters())


def step_flat(do_ar, do_opt=True):
    red.zero_grad()
    loss = crit(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
    loss.backward()
    if do_ar:
        red.all_reduce()
    if do_opt:
        opt.step()


print("flat grads, no all-reduce             %.3f ms" % timed(lambda: step_flat(False)))
print("flat grads + all-reduce (1 rank)      %.3f ms" % timed(lambda: step_flat(True)))
print("all_reduce(192 MB flat) alone         %.3f ms" % timed(lambda: red.all_reduce()))
print("one dist.all_reduce of the flat buf   %.3f ms" % timed(lambda: dist.all_reduce(red.flat)))
print("opt.step alone (flat-view grads)      %.3f ms" % timed(lambda: opt.step()))
dist.destroy_process_group()
