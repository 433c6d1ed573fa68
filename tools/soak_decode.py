"""Soak of the inference paths: N greedy decodes and N beam searches (B=128, beam 5, depth 30) back to back on fixed inputs; every call
must reproduce the first call's ids bit for bit (the fused decode schedule and the device beam queues have no atomics whose order
could matter, so any difference is a race), no asynchronous error at the end.  usage: python tools/soak_decode.py [N]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import S2VTModel
from s2vt_video_caption_amd import synth, capi, beam
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = synth.CONFIGS["c5"]
m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
m.load_state_dict(synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=0))
m.to("cuda:0").eval()
feats = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=5)[0].cuda()
with torch.no_grad():
    ref_ids = m(feats, mode="test").clone()
    ref_beam = [[int(t.item()) for t in s] for s in m(feats, mode="beam_search", beam_width=5, max_beam_depth=30)]
    t0 = time.perf_counter(); bad = 0
    for i in range(N):
        ids = m(feats, mode="test")
        if not torch.equal(ids, ref_ids): bad += 1
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / N
    t0 = time.perf_counter(); badb = 0
    for i in range(N):
        out = m(feats, mode="beam_search", beam_width=5, max_beam_depth=30)
        if [[int(t.item()) for t in s] for s in out] != ref_beam: badb += 1
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / N
capi.check_async_error()
print("greedy: %d calls, %d differ from the first, %.2f ms per call (ids compared on the host each call)" % (N, bad, tg * 1e3))
print("beam (%s): %d calls, %d differ from the first, %.2f ms per call (captions converted on the host each call)" % (beam.LAST_PATH, N, badb, tb * 1e3))
assert bad == 0 and badb == 0
print("no asynchronous error")
