"""Wall time of the training forward alone (s2vt_train_forward through the model), and of forward + backward, at one batch size.
usage: python tools/bench_fwd_only.py [B] [gemm_mode]   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import S2VTModel  # noqa: E402
import utils  # noqa: E402
from s2vt_video_caption_amd import capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = capi.load()
lib.s2vt_set_gemm_mode(mode)          # returns the previous mode
d = synth.CONFIGS["c2"]
dev = "cuda:0"
sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=1)
feats, caps, mask = (t.to(dev) for t in synth.make_batch(B, d["L"], d["F"], d["V"], seed=2))
m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"]).to(dev)
m.load_state_dict(sd)
crit = utils.MaskCriterion()
m.train()


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwd():
    return m(feats, targets=caps[:, :-1], mode="train")


def fwd_bwd():
    for p in m.parameters():
        p.grad = None
    crit(fwd(), caps, mask).backward()


print("B=%d gemm mode %d: forward %.3f ms, forward+loss+backward %.3f ms" % (B, mode, timed(fwd), timed(fwd_bwd)), flush=True)
