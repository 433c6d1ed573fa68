"""Profiling driver: a few train forward+backward passes (and one greedy decode) at a BASELINE config, meant
to run under `rocprofv3 --kernel-trace --stats` or `rocprofv3 --pmc ...` (program after `--`, no wrappers)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import S2VTModel  # noqa: E402
import utils  # noqa: E402
from s2vt_video_caption_amd import synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
decode = "--decode" in sys.argv
d = synth.CONFIGS[cfg]
sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=0)
m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
m.load_state_dict(sd)
m.to("cuda:0")
feats, caps, mask = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=1234)
feats, caps, mask = feats.cuda(), caps.cuda(), mask.cuda()
crit = utils.MaskCriterion()
from s2vt_video_caption_amd import dp, optim  # noqa: E402
opt = optim.FlatAdam(m, lr=1e-4)          # (allocations and fills of the flat buffers happen here, before the passes)
loss = None
for _ in range(iters):                    # whole optimisation steps, as bench.py times them
    loss = dp.train_step(m, crit, opt, feats, caps, mask, None)
if decode:
    with torch.no_grad():
        m.eval()(feats, mode="test")
torch.cuda.synchronize()
print("loss", float(loss) if iters else None)
