"""CPU oracle for the S2VT hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain torch-CPU tensor arithmetic, fp32 by
default, fp64 on request) of the reference's algorithm for the path SURVEY.md §8
names.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product (``S2VTModel.S2VT`` and the HIP
library) never does.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), and the arithmetic lives in PyTorch (third-party, unpinned by
the reference; 2.10.0+rocm7.0 CPU here).  The oracle is therefore pinned against
outputs of the reference itself, imported read-only in the build container by
``oracle/make_golden.py``; the vectors live in ``tests/golden/`` and
``tests/test_oracle_golden.py`` re-checks the oracle against them everywhere
(including on the GPU box, where the reference is absent).

Every function cites the reference lines it restates (paths relative to the
reference root).
"""
import heapq
import math

import torch

KEYS = (
    "vid_rnn.weight_ih_l0", "vid_rnn.weight_hh_l0", "vid_rnn.bias_ih_l0", "vid_rnn.bias_hh_l0",
    "word_rnn.weight_ih_l0", "word_rnn.weight_hh_l0", "word_rnn.bias_ih_l0", "word_rnn.bias_hh_l0",
    "feat_linear.weight", "feat_linear.bias", "out_linear.weight", "out_linear.bias",
    "embedding.weight",
)


def _cast(params, dtype):
    return {k: v.to(dtype) for k, v in params.items()}


# --------------------------------------------------------------------------- cell
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One LSTM step, torch gate order i,f,g,o.

    Restates what ``nn.LSTM`` (S2VTModel.py:19-22, called at :67/:77/:86/:93/:103)
    computes per timestep: g = x W_ih^T + b_ih + h W_hh^T + b_hh;
    c' = sigmoid(f) c + sigmoid(i) tanh(g~); h' = sigmoid(o) tanh(c').
    ``x`` may be None for an all-zero input (the reference multiplies the zeros).
    """
    g = h @ w_hh.t() + b_hh + b_ih
    if x is not None:
        g = g + x @ w_ih.t()
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def _vid_layer(p, x1, n_steps):
    """vid_rnn over ``n_steps`` steps; input is x1[:, t] for t < L, zero after
    (S2VTModel.py:64-67).  Returns h for every step [B, n_steps, H] and final (h, c)."""
    B, L, H = x1.shape
    h = x1.new_zeros(B, H)
    c = x1.new_zeros(B, H)
    outs = []
    for t in range(n_steps):
        x = x1[:, t] if t < L else None
        h, c = lstm_cell(x, h, c, p["vid_rnn.weight_ih_l0"], p["vid_rnn.weight_hh_l0"],
                         p["vid_rnn.bias_ih_l0"], p["vid_rnn.bias_hh_l0"])
        outs.append(h)
    return torch.stack(outs, dim=1), (h, c)


def _word_step(p, emb, vid_h, h, c):
    """word_rnn step on the concatenation [embed ‖ vid_out] (S2VTModel.py:75,85,91,101).
    ``emb`` None means the zero padding of the encode phase."""
    E = p["embedding.weight"].shape[1]
    w_ih = p["word_rnn.weight_ih_l0"]
    if emb is None:
        x_part = vid_h @ w_ih[:, E:].t()
    else:
        x_part = torch.cat([emb, vid_h], dim=1) @ w_ih.t()
    g = x_part + p["word_rnn.bias_ih_l0"] + h @ p["word_rnn.weight_hh_l0"].t() + p["word_rnn.bias_hh_l0"]
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


# ------------------------------------------------------------------------ forward
def forward_train(params, feats, targets, dtype=torch.float32, out_mask=None):
    """``S2VT.forward(feats, targets, mode='train')`` (S2VTModel.py:48-54, 63-81).

    feats [B, L, F]; targets [B, L-1] int64 (= caption[:, :-1]); returns logits
    [B, L-1, V].  Dropouts are p=0 (identity), one layer, unidirectional.
    """
    p = _cast(params, dtype)
    feats = feats.to(dtype)
    B, L, _ = feats.shape
    T = 2 * L - 1
    x1 = feats @ p["feat_linear.weight"].t() + p["feat_linear.bias"]          # :54
    out1, _ = _vid_layer(p, x1, T)                                            # :64-67
    emb = p["embedding.weight"][targets]                                      # :71
    H = x1.shape[2]
    h = x1.new_zeros(B, H)
    c = x1.new_zeros(B, H)
    outs = []
    for t in range(T):                                                        # :72-77
        e = emb[:, t - L] if t >= L else None
        h, c = _word_step(p, e, out1[:, t], h, c)
        if t >= L:
            outs.append(h)                                                    # :78
    res = torch.stack(outs, dim=1)
    if out_mask is not None:      # out_drop (:79) with a given keep mask [B, L-1, H], entries 0 or 1/(1-p)
        res = res * out_mask.to(dtype)
    return res @ p["out_linear.weight"].t() + p["out_linear.bias"]            # :80


def mask_criterion(logits, target, mask):
    """``MaskCriterion.forward`` (utils.py:13-26): mean CE over all B*(L-1) rows
    against target[:, 1:]; the scalar is then multiplied by the mask vector,
    summed and divided by the mask sum (which cancels)."""
    n = logits.shape[0] * logits.shape[1]
    tgt = target[:, 1:].contiguous().view(-1)
    msk = mask[:, 1:].contiguous().view(-1).to(logits.dtype)
    lg = logits.contiguous().view(n, -1)
    lse = torch.logsumexp(lg, dim=1)
    loss = (lse - lg.gather(1, tgt.view(-1, 1)).squeeze(1)).mean()            # utils.py:11,22
    return torch.sum(loss * msk) / torch.sum(msk)                             # utils.py:24-25


def greedy_decode(params, feats, sos_ix=3, dtype=torch.float32, return_margins=False):
    """``S2VT.forward(feats, mode='test')`` (S2VTModel.py:82-110): returns int64
    ids [B, L-1]; never stops at <eos>; argmax takes the lowest index on ties."""
    p = _cast(params, dtype)
    feats = feats.to(dtype)
    B, L, _ = feats.shape
    T = 2 * L - 1
    x1 = feats @ p["feat_linear.weight"].t() + p["feat_linear.bias"]
    out1, _ = _vid_layer(p, x1, T)
    H = x1.shape[2]
    h = x1.new_zeros(B, H)
    c = x1.new_zeros(B, H)
    for t in range(L):                                                        # :84-86
        h, c = _word_step(p, None, out1[:, t], h, c)
    tok = torch.full((B,), sos_ix, dtype=torch.long)                          # :89
    preds, margins = [], []
    for i in range(L - 1):                                                    # :91-107
        e = p["embedding.weight"][tok]
        h, c = _word_step(p, e, out1[:, L + i], h, c)
        logits = h @ p["out_linear.weight"].t() + p["out_linear.bias"]
        tok = torch.argmax(logits, dim=1)
        preds.append(tok)
        if return_margins:
            top2 = logits.topk(2, dim=1).values
            margins.append(top2[:, 0] - top2[:, 1])
    ids = torch.stack(preds, dim=1)                                           # :108-110
    if return_margins:
        return ids, torch.stack(margins, dim=1)
    return ids


# --------------------------------------------------------------------- beam search
class _Node:
    """``BeamSearchNode`` (S2VTModel.py:243-274): score = logp / len**0.7 where logp
    is the log-prob of the LAST token only (not cumulative, :220)."""
    __slots__ = ("vid", "word", "prev", "tok", "logp", "leng", "_score")

    def __init__(self, vid, word, prev, tok, logp, leng):
        self.vid, self.word, self.prev, self.tok, self.logp, self.leng = vid, word, prev, tok, logp, leng
        self._score = None

    def score(self):
        if self._score is None:
            self._score = self.logp / pow(float(self.leng), 0.7)              # :267
        return self._score

    def __lt__(self, other):
        # The reference only defines __gt__ (:271-274); `a < b` falls back to b.__gt__(a),
        # which is True iff a.score > b.score.
        return bool(self.score() > other.score())


def beam_search(params, feats, beam_width=3, max_depth=30, sos_ix=3, eos_ix=4, fanout=20,
                dtype=torch.float32, return_gap=False):
    """``S2VT.forward(mode='beam_search')`` + ``beam_search`` (S2VTModel.py:56-61,
    149-240).  Returns list[B] of python int lists starting with <sos>.

    Semantics kept: vid_rnn runs over the L real frames only (:57); per depth pop up
    to ``beam_width`` entries and CLEAR the queue (:190-194); a popped <eos> node is
    re-inserted unchanged (:200-202); otherwise one zero-input vid step (:208-210),
    one word step (:211-212), log_softmax (:213-214), push the top-20 tokens in
    ascending token order (:216-223); stop when the queue has <= beam_width entries
    (:227-228); answer = best entry, back-traced (:231-238).

    ``return_gap``: also return the smallest score gap any decision of the search rested on (per depth: between the
    last entry popped and the best entry discarded by the clear; at the end: between the winner and the runner-up).
    Fixture screening only (oracle/make_golden.py): a reimplementation whose log-probs differ by less than this gap
    takes the same decisions.  ``return_gap="per_sample"`` returns the list of per-sample gaps instead of their minimum.
    """
    p = _cast(params, dtype)
    feats = feats.to(dtype)
    B, L, _ = feats.shape
    x1 = feats @ p["feat_linear.weight"].t() + p["feat_linear.bias"]
    out1, (h1, c1) = _vid_layer(p, x1, L)
    H = x1.shape[2]
    h2 = x1.new_zeros(B, H)
    c2 = x1.new_zeros(B, H)
    for t in range(L):                                                        # :58-60
        h2, c2 = _word_step(p, None, out1[:, t], h2, c2)

    sentences = []
    min_gap = float("inf")
    sample_gaps = []
    for b in range(B):                                                        # :170
        min_gap = float("inf")
        root = _Node((h1[b:b + 1], c1[b:b + 1]), (h2[b:b + 1], c2[b:b + 1]), None, sos_ix, 0, 1)
        heap = []
        heapq.heappush(heap, (-root.score(), root))                           # :182
        depth = 0
        while depth < max_depth:                                              # :186
            depth += 1
            beam = []
            for _ in range(beam_width):                                       # :191-193
                if heap:
                    beam.append(heapq.heappop(heap))
            if heap and beam:
                min_gap = min(min_gap, float(heap[0][0]) - float(beam[-1][0]))
            heap = []                                                         # :194
            for key, n in beam:
                if n.tok == eos_ix and n.prev is not None:                    # :200-202
                    heapq.heappush(heap, (key, n))
                    continue
                vh, vc = lstm_cell(None, n.vid[0], n.vid[1], p["vid_rnn.weight_ih_l0"],
                                   p["vid_rnn.weight_hh_l0"], p["vid_rnn.bias_ih_l0"],
                                   p["vid_rnn.bias_hh_l0"])                    # :208-210
                e = p["embedding.weight"][torch.tensor([n.tok])]
                wh, wc = _word_step(p, e, vh, n.word[0], n.word[1])           # :211-212
                logits = (wh @ p["out_linear.weight"].t() + p["out_linear.bias"]).view(-1)
                logp = torch.log_softmax(logits, dim=0)                       # :214
                top = sorted(int(i) for i in logp.topk(fanout).indices)      # :216-219
                for i in top:
                    child = _Node((vh, vc), (wh, wc), n, i, logp[i], n.leng + 1)
                    heapq.heappush(heap, (-child.score(), child))             # :223
            if len(heap) <= beam_width:                                       # :227-228
                break
        k_fin, fin = heapq.heappop(heap)                                      # :231
        if heap:
            min_gap = min(min_gap, float(heap[0][0]) - float(k_fin))
        sent = [fin.tok]
        while fin.prev is not None:                                           # :234-236
            fin = fin.prev
            sent.append(fin.tok)
        sentences.append(sent[::-1])
        sample_gaps.append(min_gap)
    if return_gap == "per_sample":
        return sentences, sample_gaps
    if return_gap:
        return sentences, min(sample_gaps)
    return sentences


# ---------------------------------------------------------------------- train step
class OracleModel(torch.nn.Module):
    """Parameters under the reference's state_dict names; forward = forward_train.
    Used to restate the train step (train.py:114-128) with torch.optim.Adam(lr=1e-4)
    (train.py:89-93)."""

    def __init__(self, state_dict, dtype=torch.float32):
        super().__init__()
        self.dtype = dtype
        self.names = list(KEYS)
        self.params = torch.nn.ParameterList(
            [torch.nn.Parameter(state_dict[k].detach().clone().to(dtype)) for k in KEYS])

    def as_dict(self):
        return {k: p for k, p in zip(self.names, self.params)}

    def forward(self, feats, targets, out_mask=None):
        return forward_train(self.as_dict(), feats, targets, dtype=self.dtype, out_mask=out_mask)


def train_steps(state_dict, feats, caps, mask, n_steps, lr=1e-4, dtype=torch.float32):
    """n optimisation steps of train.py:116-127 on one fixed batch; returns the list of
    losses, the grads of the FIRST step (dict) and the final state_dict."""
    model = OracleModel(state_dict, dtype)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses, first_grads = [], None
    for s in range(n_steps):
        opt.zero_grad()
        logits = model(feats, caps[:, :-1])
        loss = mask_criterion(logits, caps, mask)
        loss.backward()
        if s == 0:
            first_grads = {k: p.grad.detach().clone() for k, p in model.as_dict().items()}
        opt.step()
        losses.append(float(loss))
    final = {k: p.detach().clone() for k, p in model.as_dict().items()}
    return losses, first_grads, final


# ------------------------------------------------- reference-shaped CPU baseline
class ReferenceShapedCPUModel(torch.nn.Module):
    """The same path expressed with the torch modules the reference itself calls on CPU
    (nn.LSTM / nn.Linear / nn.Embedding -> oneDNN), so that ``bench.py``'s
    ``cpu_baseline`` times what the reference's CPU run executes (S2VTModel.py:19-28,
    48-81) rather than the slower explicit-cell loop above.  Checked equal to
    ``forward_train`` in tests/test_oracle_golden.py."""

    def __init__(self, state_dict):
        super().__init__()
        V, E = state_dict["embedding.weight"].shape
        H, F = state_dict["feat_linear.weight"].shape
        self.L_vid = torch.nn.LSTM(H, H, batch_first=True)
        self.L_word = torch.nn.LSTM(H + E, H, batch_first=True)
        self.proj = torch.nn.Linear(F, H)
        self.out = torch.nn.Linear(H, V)
        self.emb = torch.nn.Embedding(V, E)
        remap = {"vid_rnn": "L_vid", "word_rnn": "L_word", "feat_linear": "proj",
                 "out_linear": "out", "embedding": "emb"}
        sd = {}
        for k, v in state_dict.items():
            head, tail = k.split(".", 1)
            sd[remap[head] + "." + tail] = v.detach().clone().float()
        self.load_state_dict(sd)
        self.E, self.H = E, H

    def forward(self, feats, targets):
        B, L, _ = feats.shape
        x1 = self.proj(feats)
        x1 = torch.cat([x1, x1.new_zeros(B, L - 1, self.H)], dim=1)
        o1, _ = self.L_vid(x1)
        e = torch.cat([x1.new_zeros(B, L, self.E), self.emb(targets)], dim=1)
        o2, _ = self.L_word(torch.cat([e, o1], dim=2))
        return self.out(o2[:, L:, :])

    @torch.no_grad()
    def greedy(self, feats, sos_ix=3):
        B, L, _ = feats.shape
        x1 = self.proj(feats)
        x1 = torch.cat([x1, x1.new_zeros(B, L - 1, self.H)], dim=1)
        o1, _ = self.L_vid(x1)
        _, st = self.L_word(torch.cat([x1.new_zeros(B, L, self.E), o1[:, :L]], dim=2))
        tok = torch.full((B,), sos_ix, dtype=torch.long)
        out = []
        for i in range(L - 1):
            x = torch.cat([self.emb(tok), o1[:, L + i]], dim=1).unsqueeze(1)
            o2, st = self.L_word(x, st)
            tok = torch.argmax(self.out(o2.squeeze(1)), dim=1)
            out.append(tok)
        return torch.stack(out, dim=1)
