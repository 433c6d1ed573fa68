"""Beam search of the S2VT decoder (`S2VT.forward(mode='beam_search')` + `S2VT.beam_search`,
S2VTModel.py:56-61, 149-240), batched on the GPU.

The reference walks every sample and every node in Python and launches three tiny torch ops per node.
Here the priority-queue bookkeeping stays on the host (it is what defines the result, including its
quirks), but all samples advance one depth at a time and every LSTM step / vocabulary projection of
that depth runs as ONE batched call into the HIP library:
  * all expandable nodes of a sample at one depth share the vid_rnn state (zero-input recurrence from
    the encoder state), so the vid step runs once per depth for the whole batch;
  * the word step and the out_linear run over all (sample, beam slot) rows at once.
Semantics kept exactly: score = log-prob of the LAST token / len**0.7 (not
cumulative), queue cleared after popping `beam_width` entries, finished (<eos>) entries re-inserted
unchanged, fan-out = top-20 tokens pushed in ascending token order, stop when the queue holds
<= beam_width entries, answer = best queue entry back-traced, first element the [[<sos>]] tensor.
"""
import heapq

import numpy as np
import torch

from . import ops

FANOUT = 20  # hard-coded topk(20) of the reference (S2VTModel.py:216)


class BeamSearchNode(object):
    """API-compatible with the reference's BeamSearchNode (S2VTModel.py:243-274)."""

    def __init__(self, vid_hid, word_hid, previousNode, wordId, logProb, length):
        self.vid_hid = vid_hid
        self.word_hid = word_hid
        self.prevNode = previousNode
        self.wordid = wordId
        self.logp = logProb
        self.leng = length
        self.score = None

    def eval(self, alpha=0.7):
        if self.score is None:
            self.score = self.logp / pow(float(self.leng), alpha)
        return self.score

    def __gt__(self, other):
        return bool(other.eval() > self.eval())

    def __lt__(self, other):
        # the reference defines only __gt__; `a < b` therefore resolves to b.__gt__(a)
        return bool(self.eval() > other.eval())


@torch.no_grad()
def beam_search(model, feats, params, beam_width=3, max_depth=30):
    (w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2, w_f, b_f, w_o, b_o, emb) = [p.detach() for p in params]
    dev = feats.device
    B, L, _ = feats.shape
    H, E, V = model.dim_hid, model.dim_embed, model.vocab_size
    if V < FANOUT:
        raise RuntimeError("beam search needs vocab_size >= %d (topk(20), S2VTModel.py:216)" % FANOUT)
    sos, eos = int(model.sos_ix), int(model.eos_ix)
    bsum1 = (b_ih1 + b_hh1).contiguous()
    bsum2 = (b_ih2 + b_hh2).contiguous()
    w_v = w_ih2[:, E:]          # [4H, H] view, row stride E+H
    w_e = w_ih2[:, :E]

    # ---- encoder: vid_rnn over the L real frames only, word_rnn with a zero embedding (S2VTModel.py:57-60)
    x1 = ops.feat_proj_fwd(feats.contiguous(), w_f, b_f)                       # [L*B, H] time-major
    gx1 = _gemm_strided(x1, w_ih1, bsum1)
    h1_all, c1_all, _ = ops.lstm_seq_fwd(L, B, gx1, L, None, w_hh1)
    gx2 = _gemm_strided(h1_all, w_v, bsum2)
    h2_all, c2_all, _ = ops.lstm_seq_fwd(L, B, gx2, L, None, w_hh2)
    vid_h, vid_c = h1_all[(L - 1) * B:], c1_all[(L - 1) * B:]
    word_h, word_c = h2_all[(L - 1) * B:].clone(), c2_all[(L - 1) * B:].clone()

    # ---- per-sample queues: entries (key, node); node.word_hid = (depth table index, row)
    states = [(word_h, word_c)]
    heaps = []
    for b in range(B):
        root = BeamSearchNode(None, (0, b), None, sos, 0, 1)
        heaps.append([(-root.eval(), root)])
    done = [False] * B
    depth = 0
    while depth < max_depth and not all(done):
        depth += 1
        beams = [None] * B
        rows_b, rows_state, rows_tok, rows_node = [], [], [], []
        for b in range(B):
            if done[b]:
                continue
            heap = heaps[b]
            beam = [heapq.heappop(heap) for _ in range(min(beam_width, len(heap)))]
            heaps[b] = []                                                      # queue cleared (:194)
            beams[b] = beam
            for key, n in beam:
                if n.wordid == eos and n.prevNode is not None:
                    continue
                rows_b.append(b)
                rows_state.append(n.word_hid)
                rows_tok.append(n.wordid)
                rows_node.append(n)
        # one zero-input vid step for the whole batch (:208-210)
        vid_h, vid_c = ops.lstm_step_fwd(None, bsum1, w_hh1, vid_h.contiguous(), vid_c.contiguous())
        R = len(rows_b)
        if R:
            bidx = torch.tensor(rows_b, device=dev)
            tok = torch.tensor(rows_tok, device=dev)
            # every expandable node of this depth was created at the previous depth -> one state table
            assert all(d == rows_state[0][0] for d, _ in rows_state)
            sh, sc = states[rows_state[0][0]]
            ridx = torch.tensor([r for _, r in rows_state], device=dev)
            ph, pc = sh[ridx], sc[ridx]
            gx = _gemm_strided(vid_h[bidx].contiguous(), w_v, bsum2)             # vid_out half + biases
            gx = _gemm_strided(emb[tok].contiguous(), w_e, None, out=gx, accumulate=True)   # embedded word half
            wh, wc = ops.lstm_step_fwd(gx, None, w_hh2, ph.contiguous(), pc.contiguous())   # (:211-212)
            logits = ops.gemm(wh, w_o, bias=b_o)                                # (:213)
            logp = torch.log_softmax(logits, dim=1)                             # (:214)
            top = logp.topk(FANOUT, dim=1).indices.sort(dim=1).values          # ascending token order (:216-219)
            top_lp = logp.gather(1, top).cpu().numpy()
            top_ix = top.cpu().numpy()
            states[-1] = None                      # parents of the next depth live in the new table only
            states.append((wh, wc))
        # push phase, in the reference's order
        r = 0
        for b in range(B):
            if beams[b] is None:
                continue
            heap = heaps[b]
            for key, n in beams[b]:
                if n.wordid == eos and n.prevNode is not None:
                    heapq.heappush(heap, (key, n))                             # (:200-202)
                    continue
                leng = n.leng + 1
                for j in range(FANOUT):
                    child = BeamSearchNode(None, (len(states) - 1, r), n, int(top_ix[r, j]), top_lp[r, j], leng)
                    heapq.heappush(heap, (-child.eval(), child))               # (:220-223)
                r += 1
            if len(heap) <= beam_width:                                        # (:227-228)
                done[b] = True
    sentences = []
    for b in range(B):
        _, node = heapq.heappop(heaps[b])                                      # (:231)
        seq = [node.wordid]
        while node.prevNode is not None:                                       # (:234-236)
            node = node.prevNode
            seq.append(node.wordid)
        seq = seq[::-1]
        out = [torch.tensor([[seq[0]]], dtype=torch.long, device=dev)]
        out += [torch.tensor(i, device=dev) for i in seq[1:]]
        sentences.append(out)
    return sentences


def _gemm_strided(a, w, bias, out=None, accumulate=False):
    """a[M,K]·w[N,K]^T where w may be a column slice (row stride > K) of a larger matrix."""
    import ctypes
    from . import capi
    from .functional import _ptr, _stream
    lib = capi.load()
    M, K = a.shape
    N = w.shape[0]
    dev = a.device
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(M, N, dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_gemm_f32(1, 1, M, N, K, _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out),
                                     out.stride(0), _ptr(bias), int(accumulate), _stream(dev)), "s2vt_gemm_f32")
    return out
