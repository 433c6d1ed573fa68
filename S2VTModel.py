"""Drop-in `S2VTModel` module for Kamino666/S2VT-video-caption, MI355X-native.

Same module name, class name, constructor and `forward` signature, sub-module names/types and
attributes as the reference (S2VTModel.py:10-37), so `torch.load` of reference full-module pickles,
`load_state_dict`, optimisers and the reference's train/eval scripts work unchanged.  The arithmetic
of `forward` does not go through `nn.LSTM`/`nn.Linear`: it is executed by hand-written HIP kernels
behind the C ABI in `include/s2vt_hip.h`.  The `nn.*` sub-modules are parameter
containers only.  HIP tensors only — there is no CPU fallback.
"""
import torch
from torch import nn

import s2vt_video_caption_amd  # noqa: F401  (registers the package alias)
from s2vt_video_caption_amd import capi as _capi
from s2vt_video_caption_amd import functional as _F
from s2vt_video_caption_amd import beam as _beam


class S2VT(nn.Module):
    def __init__(self, vocab_size, feat_dim, length, dim_hid=500, dim_embed=500, feat_dropout=0, rnn_dropout=0,
                 out_dropout=0, num_layers=1, bidirectional=False, rnn_type='lstm', sos_ix=3, eos_ix=4):
        super(S2VT, self).__init__()
        # construction order = the reference's (S2VTModel.py:19-28) so seeded default init matches
        rnn_cell = nn.LSTM if rnn_type.lower() == 'lstm' else nn.GRU
        self.vid_rnn = rnn_cell(dim_hid, dim_hid, batch_first=True, num_layers=num_layers,
                                bidirectional=bidirectional, dropout=rnn_dropout)
        self.word_rnn = rnn_cell(dim_hid + dim_embed, dim_hid, batch_first=True, num_layers=num_layers,
                                 bidirectional=bidirectional, dropout=rnn_dropout)
        self.feat_drop = nn.Dropout(p=feat_dropout)
        self.out_drop = nn.Dropout(p=out_dropout)
        self.feat_linear = nn.Linear(feat_dim, dim_hid)
        self.out_linear = nn.Linear(dim_hid, vocab_size)
        self.embedding = nn.Embedding(vocab_size, dim_embed)
        self.feat_dim = feat_dim
        self.length = length
        self.dim_hid = dim_hid
        self.dim_embed = dim_embed
        self.sos_ix = sos_ix
        self.eos_ix = eos_ix
        self.vocab_size = vocab_size
        self.rnn_type = rnn_type

    # -- the 13 tensors in include/s2vt_hip.h order
    def _hip_params(self):
        self._check_supported()
        return (self.vid_rnn.weight_ih_l0, self.vid_rnn.weight_hh_l0, self.vid_rnn.bias_ih_l0,
                self.vid_rnn.bias_hh_l0, self.word_rnn.weight_ih_l0, self.word_rnn.weight_hh_l0,
                self.word_rnn.bias_ih_l0, self.word_rnn.bias_hh_l0, self.feat_linear.weight,
                self.feat_linear.bias, self.out_linear.weight, self.out_linear.bias, self.embedding.weight)

    def _check_supported(self):
        for rnn in (self.vid_rnn, self.word_rnn):
            if not isinstance(rnn, nn.LSTM) or rnn.num_layers != 1 or rnn.bidirectional or not rnn.bias:
                raise NotImplementedError(
                    "the HIP S2VT path implements the reference configuration (1-layer unidirectional LSTM); "
                    "GRU / num_layers>1 / bidirectional are outside the hot path (SURVEY.md §8)")

    def forward(self, feats, targets=None, mode='train', beam_width=3, max_beam_depth=30):
        """
        :param feats: [B, L, feat_dim]
        :param targets: [B, L-1] word ids (train mode)
        :param mode: 'train' -> logits [B, L-1, V]; 'test' -> greedy ids [B, L-1] (int64);
                     'beam_search' -> list of id sequences (each starting with <sos>)
        """
        _F.require_hip(feats, "feats")
        if feats.dim() != 3 or feats.shape[1] != self.length or feats.shape[2] != self.feat_dim:
            raise ValueError("feats must be [B, %d, %d], got %s" % (self.length, self.feat_dim, tuple(feats.shape)))
        params = self._hip_params()
        feats = self.feat_drop(feats)                      # identity at the reference's p=0 (S2VTModel.py:52)
        if mode == 'beam_search':
            return _beam.beam_search(self, feats, params, beam_width=beam_width, max_depth=max_beam_depth)
        if mode == 'train':
            if targets is None:
                raise ValueError("mode='train' needs targets")
            out_mask = None
            if self.training and self.out_drop.p > 0:
                # S2VTModel.py:79 applies nn.Dropout to the [B, L-1, H] decode-step hidden states: the same call on a ones
                # tensor of that shape draws the same mask from torch's generator; the kernels want it time-major
                B, H = feats.shape[0], self.dim_hid
                keep = self.out_drop(torch.ones(B, self.length - 1, H, dtype=torch.float32, device=feats.device))
                out_mask = keep.transpose(0, 1).reshape((self.length - 1) * B, H).contiguous()
            return _F.train_forward(feats, targets, params, grad_sink=_F.grad_sink_for(self), out_mask=out_mask)
        elif mode == 'test':
            return _F.greedy_decode(feats, params, self.sos_ix, owner=self)
        return None                                        # the reference falls through for unknown modes

    @staticmethod
    def _get_word2embed_from_glove(glove_path, ix2word):
        """{word: [floats]} for the vocabulary words found in a GloVe text file (one 'word v1 v2 ...' line per word)."""
        wanted = set(ix2word.values())
        table = {}
        with open(glove_path, encoding='utf-8') as f:
            for line in f:
                word, _, rest = line.rstrip('\n').partition(' ')
                if word in wanted:
                    table[word] = [float(x) for x in rest.split(' ') if x]
        return table

    def load_glove_weights(self, glove_path, glove_dim, ix2word, word2embed='./data/word2embed.json'):
        """Initialise the embedding from GloVe vectors (S2VTModel.py:112-147; train.py:88 has the call commented out).
        `word2embed` None: parse `glove_path` and cache the {word: vector} table as ./data/word2embed.json; otherwise the
        path of such a cache.  Words without a vector keep a Xavier-normal row; the embedding stays trainable.  Host-side
        torch glue: the new `nn.Embedding` is an ordinary parameter of the HIP path."""
        import json
        import os
        assert glove_dim == self.dim_embed
        if word2embed is None:
            table = self._get_word2embed_from_glove(glove_path, ix2word)
            os.makedirs('./data', exist_ok=True)
            with open('./data/word2embed.json', 'w+', encoding='utf-8') as fp:
                json.dump(table, fp)
        else:
            with open(word2embed, encoding='utf-8') as fp:
                table = json.load(fp)
        print('get {} word2embed'.format(len(table)))
        dev = self.embedding.weight.device
        weights = torch.zeros([self.vocab_size, glove_dim], dtype=torch.float, device=dev)
        torch.nn.init.xavier_normal_(weights)
        for ix, word in ix2word.items():
            if word in table:
                weights[int(ix)] = torch.tensor(table[word], dtype=torch.float, device=dev)
        self.embedding = nn.Embedding.from_pretrained(weights, freeze=False)


BeamSearchNode = _beam.BeamSearchNode
