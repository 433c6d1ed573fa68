# Per-call timing of mode="beam_search" before and after training steps in the same process (the first call after training
# pays a generation-2 pass of the cyclic collector: profiles/round4_beam_first_call_after_training.txt).
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, S2VTModel, utils
from s2vt_video_caption_amd import synth, dp, capi, beam
lib = capi.load()
L, F, H, E, V = 80, 4096, 1000, 1000, 12000
dev = "cuda:0"
sd = synth.make_state_dict(V, F, H, E, seed=0)
model = S2VTModel.S2VT(V, F, L, dim_hid=H, dim_embed=E); model.load_state_dict(sd); model.to(dev)
crit = utils.MaskCriterion()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
feats, caps, mask = (t.to(dev) for t in synth.make_batch(64, L, F, V, seed=1234))
dfe = synth.make_batch(128, L, F, V, seed=99)[0].to(dev)
PROFILE = True
def beam_time(tag):
    model.load_state_dict(sd); model.eval()
    with torch.no_grad():
        model(dfe, mode="test"); model(dfe, mode="beam_search", beam_width=5, max_beam_depth=30); torch.cuda.synchronize()
        ts = []
        seg0 = torch.cuda.memory_stats()["segment.all.allocated"]
        import cProfile, pstats, gc
        for i in range(5):
            pr = cProfile.Profile() if (i == 0 and PROFILE) else None
            g0 = [s_["collections"] for s_ in gc.get_stats()]
            t0 = time.perf_counter()
            if pr: pr.enable()
            model(dfe, mode="beam_search", beam_width=5, max_beam_depth=30)
            torch.cuda.synchronize()
            if pr: pr.disable()
            ts.append((time.perf_counter() - t0) * 1e3)
            g1 = [s_["collections"] for s_ in gc.get_stats()]
            if ts[-1] > 40:
                print("slow call", i, "gc collections", [b - a for a, b in zip(g0, g1)], flush=True)
                if pr: pstats.Stats(pr).sort_stats("tottime").print_stats(14)
        seg1 = torch.cuda.memory_stats()["segment.all.allocated"]
    print(tag, "beam calls ms", ["%.1f" % t for t in ts], "new segments", seg1 - seg0, flush=True)
    model.train()
beam_time("fresh")
for n in (5, 5):
    for _ in range(n): dp.train_step(model, crit, opt, feats, caps, mask, None)
    torch.cuda.synchronize()
    beam_time("after %d more train steps" % n)
f2, c2, m2 = (t.to(dev) for t in synth.make_batch(128, L, F, V, seed=4321))
for _ in range(20): dp.train_step(model, crit, opt, f2, c2, m2, None)
torch.cuda.synchronize()
beam_time("after B=128 steps")
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
