"""Profiling driver: greedy decode calls at C2/C5 dims (run under rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import S2VTModel
from s2vt_video_caption_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = synth.CONFIGS["c2"]
m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
m.load_state_dict(synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=0))
m.to("cuda:0").eval()
feats = synth.make_batch(B, d["L"], d["F"], d["V"], seed=5)[0].cuda()
with torch.no_grad():
    for _ in range(4):
        m(feats, mode="test")
torch.cuda.synchronize()
