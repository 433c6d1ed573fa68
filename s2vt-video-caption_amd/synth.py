"""Synthetic weights and inputs (SURVEY.md §8(c)/(d) recipe).

Weights come from an explicit per-key recipe that does not depend on module
construction order, so the reference model (loaded through ``load_state_dict``),
the CPU oracle and the HIP path all see bit-identical parameters.  Inputs mirror
the shapes the reference's loader yields (`dataloader.py:28-50`): fp32 features
``[B, L, F]``, int64 captions ``[B, L]`` starting with ``<sos>``=3, a random
number of word ids, ``<eos>``=4, then ``<pad>``=0, and a 0/1 float mask.
"""
import math

import torch

# state_dict layout of the reference model (S2VTModel.py:19-28), see SURVEY.md §5.
def param_shapes(V, F, H, E):
    return {
        "vid_rnn.weight_ih_l0": (4 * H, H),
        "vid_rnn.weight_hh_l0": (4 * H, H),
        "vid_rnn.bias_ih_l0": (4 * H,),
        "vid_rnn.bias_hh_l0": (4 * H,),
        "word_rnn.weight_ih_l0": (4 * H, E + H),
        "word_rnn.weight_hh_l0": (4 * H, H),
        "word_rnn.bias_ih_l0": (4 * H,),
        "word_rnn.bias_hh_l0": (4 * H,),
        "feat_linear.weight": (H, F),
        "feat_linear.bias": (H,),
        "out_linear.weight": (V, H),
        "out_linear.bias": (V,),
        "embedding.weight": (V, E),
    }


def make_state_dict(V, F, H, E, seed=0, out_scale=1.0):
    """Seeded parameters with torch-default-like magnitudes.

    LSTM / Linear tensors: U(-k, k), k = 1/sqrt(fan) (fan = H for the LSTMs,
    in_features for the Linears); embedding: N(0, 1).  One generator per key
    (seed + index in sorted key order) so adding a key never shifts the others.
    ``out_scale`` widens the logits (a larger top-2 margin makes greedy token
    ids robust to fp32 summation order, SURVEY.md §7 "Bit-exact token ids").
    """
    shapes = param_shapes(V, F, H, E)
    sd = {}
    for idx, key in enumerate(sorted(shapes)):
        g = torch.Generator().manual_seed(1000003 * seed + idx)
        shape = shapes[key]
        if key == "embedding.weight":
            t = torch.randn(shape, generator=g)
        else:
            if key.startswith("feat_linear"):
                fan = F
            else:
                fan = H
            k = 1.0 / math.sqrt(fan)
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * k
            if key.startswith("out_linear"):
                t = t * out_scale
        sd[key] = t.float().contiguous()
    return sd


def make_batch(B, L, F, V, seed=1234, relu=False, min_words=5, max_words=15):
    """feats [B,L,F] f32, captions [B,L] i64, mask [B,L] f32."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, L, F, generator=g)
    if relu:
        feats = feats.relu_()
    caps = torch.zeros(B, L, dtype=torch.long)
    mask = torch.zeros(B, L)
    hi = max(min(max_words, L - 2), 1)
    lo = min(min_words, hi)
    n_words = torch.randint(lo, hi + 1, (B,), generator=g)
    low_tok = 5 if V > 5 else 0
    for b in range(B):
        n = int(n_words[b])
        caps[b, 0] = 3 if V > 3 else 0
        caps[b, 1:1 + n] = torch.randint(low_tok, V, (n,), generator=g)
        caps[b, 1 + n] = 4 if V > 4 else 0
        mask[b, :n + 2] = 1.0
    return feats, caps, mask


CONFIGS = {
    # name: dims of BASELINE.json configs
    "tiny": dict(B=3, L=8, F=64, H=32, E=24, V=50),
    "c1": dict(B=4, L=80, F=4096, H=500, E=500, V=100),
    "c2": dict(B=64, L=80, F=4096, H=1000, E=1000, V=12000),
    "c3": dict(B=256, L=80, F=4096, H=1000, E=1000, V=12000),
    "c4": dict(B=128, L=80, F=4096, H=1000, E=1000, V=12000),
    "c5": dict(B=128, L=80, F=4096, H=1000, E=1000, V=12000),
    # not a BASELINE config: B = 64 (split-precision / two-stream drivers) at dims small enough for long CPU reference runs
    "mid64": dict(B=64, L=24, F=512, H=256, E=256, V=1000),
}
