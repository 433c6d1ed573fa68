"""Where does a batched beam search call spend its time: host queue bookkeeping (pop / push / finish) vs the rest
(device work + transfers).  usage: python tools/prof_beam.py [B] [beam]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import S2VTModel
from s2vt_video_caption_amd import synth, beam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = synth.CONFIGS["c5"]
m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
m.load_state_dict(synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=0))
m.to("cuda:0").eval()
feats = synth.make_batch(B, d["L"], d["F"], d["V"], seed=5)[0].cuda()
acc = {"pop": 0.0, "push": 0.0, "finish": 0.0}
Q = beam.BeamQueues
for name in acc:
    orig = getattr(Q, name)
    def wrap(self, *a, _o=orig, _n=name):
        t = time.perf_counter(); r = _o(self, *a); acc[_n] += time.perf_counter() - t; return r
    setattr(Q, name, wrap)
with torch.no_grad():
    m(feats, mode="beam_search", beam_width=bw, max_beam_depth=30)
    torch.cuda.synchronize()
    for k in acc: acc[k] = 0.0
    t0 = time.perf_counter()
    m(feats, mode="beam_search", beam_width=bw, max_beam_depth=30)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
print("B=%d beam=%d: total %.1f ms; host queues: pop %.1f, push %.1f, finish %.1f ms; rest (device + transfers + glue) %.1f ms" %
      (B, bw, tot * 1e3, acc["pop"] * 1e3, acc["push"] * 1e3, acc["finish"] * 1e3, (tot - sum(acc.values())) * 1e3))
if os.environ.get("CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable()
        m(feats, mode="beam_search", beam_width=bw, max_beam_depth=30)
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
# the same call with the queues on the host (round 3's path) beside the device queues (default since round 4)
for dq in (True, False, True, False):
    beam.DEVICE_QUEUES = dq
    with torch.no_grad():
        m(feats, mode="beam_search", beam_width=bw, max_beam_depth=30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m(feats, mode="beam_search", beam_width=bw, max_beam_depth=30)
        torch.cuda.synchronize()
    print("device queues %s: %.2f ms per call" % (dq, (time.perf_counter() - t0) / 3 * 1e3))
