"""Drop-in `attention_baseline` module for Kamino666/S2VT-video-caption on MI355X: the network the reference's committed
train.py instantiates (train.py:86-87, attention_baseline.py:9-105), with the same module / class / sub-module names and
constructor, so state_dicts and full-module pickles of the reference load unchanged.

`forward` does not call nn.LSTM / nn.Linear: the projections are s2vt_gemm_f32, the three recurrences (encoder forward,
encoder reverse, decoder) the fused LSTM timestep kernels of the S2VT path with their BPTT, the greedy loop
s2vt_lstm_step_fwd + s2vt_decode_step_argmax - see s2vt-video-caption_amd/att_functional.py, which also explains why the
"attention" of this network is a plain sum over the frames (a softmax over a dimension of size one).  The nn.* sub-modules only
hold the parameters.  HIP tensors only: there is no CPU fallback.
"""
from torch import nn

import s2vt_video_caption_amd  # noqa: F401  (registers the package alias)
from s2vt_video_caption_amd import att_functional as _A
from s2vt_video_caption_amd.functional import require_hip as _require_hip


class Att_Baseline(nn.Module):
    def __init__(self, vocab_size, dim_feat, length, dim_hid=500, dim_embed=500, feat_dropout=0, out_dropout=0, sos_ix=3,
                 eos_ix=4):
        super().__init__()
        self.vocab_size, self.dim_feat, self.length = vocab_size, dim_feat, length
        self.dim_hid, self.dim_embed = dim_hid, dim_embed
        self.sos_ix, self.eos_ix = sos_ix, eos_ix
        # parameter containers, created in the reference's order (attention_baseline.py:23-33) so that a seeded default
        # initialisation draws the same numbers
        self.encoder = nn.LSTM(dim_hid, dim_hid, batch_first=True, bidirectional=True)
        self.decoder = nn.LSTM(2 * dim_hid + dim_embed, dim_hid, batch_first=True)
        self.feat_linear = nn.Linear(dim_feat, dim_hid)
        self.feat_drop = nn.Dropout(p=feat_dropout)
        self.embedding = nn.Embedding(vocab_size, dim_embed, padding_idx=0)
        self.out_linear = nn.Linear(dim_hid, vocab_size)
        self.out_drop = nn.Dropout(p=out_dropout)
        self.att_enc = nn.Linear(2 * dim_hid, dim_hid, bias=True)
        self.att_prev_hid = nn.Linear(dim_hid, dim_hid, bias=True)
        self.att_apply = nn.Linear(dim_hid, 1, bias=False)

    def forward(self, feats, targets=None, mode='train'):
        """feats [B, length, dim_feat] (HIP, fp32).  mode 'train': targets [B, length-1] int64 -> logits [B, length-1, vocab]
        (autograd-connected to every parameter); mode 'test': greedy ids [B, length] int64."""
        _require_hip(feats, "feats")
        if feats.dim() != 3 or feats.shape[1] != self.length or feats.shape[2] != self.dim_feat:
            raise ValueError("feats must be [B, %d, %d], got %s" % (self.length, self.dim_feat, tuple(feats.shape)))
        if mode == 'train':
            if targets is None:
                raise ValueError("mode='train' needs targets [B, length-1]")
            return _A.train_forward(self, feats.float(), targets)
        if mode == 'test':
            return _A.greedy_decode(self, feats.float())
        raise ValueError("mode must be 'train' or 'test', got %r" % (mode,))
