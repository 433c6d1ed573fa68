"""Profiling driver: BASELINE configs[4] beam search (B=128, beam 5, depth 30) on the default path (device queues + plane-path
depth step), one greedy decode first (fills the weight-image cache), then three searches.
usage: rocprofv3 --kernel-trace --stats -- python3 tools/prof_beam_device.py   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import S2VTModel
from s2vt_video_caption_amd import synth, beam
d = synth.CONFIGS["c5"]
m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
m.load_state_dict(synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=0))
m.to("cuda:0").eval()
feats = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=5)[0].cuda()
with torch.no_grad():
    m(feats, mode="test")
    m(feats, mode="beam_search", beam_width=5, max_beam_depth=30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m(feats, mode="beam_search", beam_width=5, max_beam_depth=30)
    torch.cuda.synchronize()
print("beam: %.2f ms per call (%s)" % ((time.perf_counter() - t0) / 3 * 1e3, beam.LAST_PATH))
