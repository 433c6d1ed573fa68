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

    # ---- per-sample queues (host): see BeamQueues
    states = (word_h, word_c)
    queues = (BeamQueues if FAST_QUEUES else HeapQueues)(B, beam_width, sos, eos)
    depth = 0
    while depth < max_depth and not queues.all_done():
        depth += 1
        rows_b, rows_state, rows_tok = queues.pop()
        # one zero-input vid step for the whole batch (:208-210)
        vid_h, vid_c = ops.lstm_step_fwd(None, bsum1, w_hh1, vid_h.contiguous(), vid_c.contiguous())
        top_ix = top_lp = None
        if len(rows_b):
            bidx = torch.as_tensor(np.asarray(rows_b, dtype=np.int64), device=dev)
            tok = torch.as_tensor(np.asarray(rows_tok, dtype=np.int64), device=dev)
            ridx = torch.as_tensor(np.asarray(rows_state, dtype=np.int64), device=dev)
            # every expandable node of this depth was created at the previous depth -> one state table
            sh, sc = states
            ph, pc = sh[ridx], sc[ridx]
            gx = _gemm_strided(vid_h[bidx].contiguous(), w_v, bsum2)             # vid_out half + biases
            gx = _gemm_strided(emb[tok].contiguous(), w_e, None, out=gx, accumulate=True)   # embedded word half
            wh, wc = ops.lstm_step_fwd(gx, None, w_hh2, ph.contiguous(), pc.contiguous())   # (:211-212)
            logits = ops.gemm(wh, w_o, bias=b_o)                                # (:213)
            logp = torch.log_softmax(logits, dim=1)                             # (:214)
            top = logp.topk(FANOUT, dim=1).indices.sort(dim=1).values          # ascending token order (:216-219)
            both = torch.cat([logp.gather(1, top), top.to(torch.float32)], dim=1).cpu().numpy()   # one D2H copy
            top_lp = np.ascontiguousarray(both[:, :FANOUT])
            top_ix = both[:, FANOUT:].astype(np.int64)                          # exact: V < 2**24
            states = (wh, wc)
        queues.push(top_ix, top_lp)
    sentences = []
    for seq in queues.finish():
        out = [torch.tensor([[seq[0]]], dtype=torch.long, device=dev)]
        out += [torch.tensor(i, device=dev) for i in seq[1:]]
        sentences.append(out)
    return sentences


FAST_QUEUES = True


class HeapQueues(object):
    """The reference's queue bookkeeping, literally: one Python heap of (key, node) tuples per sample
    (S2VTModel.py:186-236).  Kept as the definition the vectorised BeamQueues is tested against and as the
    tie fallback's model; ~30 ms per depth at B=128, beam 5."""

    def __init__(self, B, beam_width, sos, eos):
        self.B, self.bw, self.eos = B, beam_width, eos
        self.heaps = []
        for b in range(B):
            root = BeamSearchNode(None, b, None, sos, 0, 1)
            self.heaps.append([(-root.eval(), root)])
        self.done = [False] * B
        self.beams = [None] * B

    def all_done(self):
        return all(self.done)

    def pop(self):
        rows_b, rows_state, rows_tok = [], [], []
        self.beams = [None] * self.B
        for b in range(self.B):
            if self.done[b]:
                continue
            heap = self.heaps[b]
            beam = [heapq.heappop(heap) for _ in range(min(self.bw, len(heap)))]
            self.heaps[b] = []                                                 # queue cleared (:194)
            self.beams[b] = beam
            for key, n in beam:
                if n.wordid == self.eos and n.prevNode is not None:
                    continue
                rows_b.append(b)
                rows_state.append(n.word_hid)
                rows_tok.append(n.wordid)
        return rows_b, rows_state, rows_tok

    def push(self, top_ix, top_lp):
        r = 0
        for b in range(self.B):
            if self.beams[b] is None:
                continue
            heap = self.heaps[b]
            for key, n in self.beams[b]:
                if n.wordid == self.eos and n.prevNode is not None:
                    heapq.heappush(heap, (key, n))                             # (:200-202)
                    continue
                leng = n.leng + 1
                for j in range(FANOUT):
                    child = BeamSearchNode(None, r, n, int(top_ix[r, j]), top_lp[r, j], leng)
                    heapq.heappush(heap, (-child.eval(), child))               # (:220-223)
                r += 1
            if len(heap) <= self.bw:                                           # (:227-228)
                self.done[b] = True

    def finish(self):
        out = []
        for b in range(self.B):
            _, node = heapq.heappop(self.heaps[b])                             # (:231)
            seq = [node.wordid]
            while node.prevNode is not None:                                   # (:234-236)
                node = node.prevNode
                seq.append(node.wordid)
            out.append(seq[::-1])
        return out


class _Slot(object):
    """heap payload whose comparison answers what BeamSearchNode's does for EQUAL scores: never less."""
    __slots__ = ("i",)

    def __init__(self, i):
        self.i = i

    def __lt__(self, other):
        return False


def _heap_order(keys, m):
    """Indices the reference's heap would pop first, m of them: push in order, pop m times (ties included)."""
    heap = []
    for i, k in enumerate(keys):
        heapq.heappush(heap, (k, _Slot(i)))
    return [heapq.heappop(heap)[1].i for _ in range(m)]


class BeamQueues(object):
    """Same bookkeeping as HeapQueues with the per-depth candidate sets held as numpy arrays.

    Because the queue is emptied after every pop phase (:194), a sample's heap at depth d is just "the
    candidates pushed at depth d-1, in push order"; popping beam_width entries is a partial sort by key.
    Distinct keys sort the same way in any heap, so the fast path is one stable argsort per sample; when two
    of the beam_width+1 smallest keys are EQUAL the pop order depends on the heap's internal layout, and the
    sample falls back to replaying the pushes into a real heap (_heap_order).  Nodes are materialised only
    when popped (<= beam_width per sample and depth instead of 20 x beam_width).
    """

    def __init__(self, B, beam_width, sos, eos):
        self.B, self.bw, self.eos = B, beam_width, eos
        # node table: token and parent of every popped node (back-trace), plus the root per sample
        self.tok = [sos] * B
        self.prev = [-1] * B
        # candidates per sample, in push order.  A candidate is (key, token, parent node, length, state row,
        # node id or -1 when not materialised yet).
        self.c_key = [np.array([-0.0], dtype=np.float64) for _ in range(B)]
        self.c_tok = [np.array([sos]) for _ in range(B)]
        self.c_par = [np.array([-1]) for _ in range(B)]
        self.c_len = [np.array([1]) for _ in range(B)]
        self.c_row = [np.array([b]) for b in range(B)]
        self.c_nid = [np.array([b]) for b in range(B)]
        self.done = [False] * B
        self.beams = [None] * B
        self.tie_fallbacks = 0

    def all_done(self):
        return all(self.done)

    def _order(self, keys, m):
        n = len(keys)
        if n == 1:
            return [0]
        idx = np.argsort(keys, kind="stable")[:m + 1]
        ks = keys[idx]
        if np.any(ks[1:] == ks[:-1]):
            self.tie_fallbacks += 1
            return _heap_order(keys.tolist(), m)
        return idx[:m].tolist()

    def pop(self):
        rows_b, rows_state, rows_tok = [], [], []
        self.beams = [None] * self.B
        for b in range(self.B):
            if self.done[b]:
                continue
            keys = self.c_key[b]
            order = self._order(keys, min(self.bw, len(keys)))
            toks, pars, lens, rows, nids = self.c_tok[b], self.c_par[b], self.c_len[b], self.c_row[b], self.c_nid[b]
            beam = []
            for i in order:
                nid = int(nids[i])
                t = int(toks[i])
                if nid < 0:
                    nid = len(self.tok)
                    self.tok.append(t)
                    self.prev.append(int(pars[i]))
                fin = (t == self.eos and self.prev[nid] >= 0)
                beam.append((keys[i], nid, int(lens[i]), fin))
                if not fin:
                    rows_b.append(b)
                    rows_state.append(int(rows[i]))
                    rows_tok.append(t)
            self.beams[b] = beam
        return rows_b, rows_state, rows_tok

    def push(self, top_ix, top_lp):
        r = 0
        F = FANOUT
        for b in range(self.B):
            beam = self.beams[b]
            if beam is None:
                continue
            n = sum(1 if fin else F for _, _, _, fin in beam)
            key = np.empty(n, dtype=np.float64)
            tok = np.empty(n, dtype=np.int64)
            par = np.empty(n, dtype=np.int64)
            ln = np.empty(n, dtype=np.int64)
            row = np.empty(n, dtype=np.int64)
            nid = np.empty(n, dtype=np.int64)
            o = 0
            for k, node, leng, fin in beam:
                if fin:                                                        # re-inserted unchanged (:200-202)
                    key[o], tok[o], par[o], ln[o], row[o], nid[o] = k, self.eos, self.prev[node], leng, -1, node
                    o += 1
                    continue
                # score = fp32 log-prob / python float len**0.7, evaluated in fp32 like the reference's
                # 0-dim tensor / float division (S2VTModel.py:262-266); key = -score
                sc = top_lp[r] / np.float32(pow(float(leng + 1), 0.7))
                key[o:o + F] = -sc.astype(np.float64)
                tok[o:o + F] = top_ix[r]
                par[o:o + F] = node
                ln[o:o + F] = leng + 1
                row[o:o + F] = r
                nid[o:o + F] = -1
                o += F
                r += 1
            self.c_key[b], self.c_tok[b], self.c_par[b], self.c_len[b], self.c_row[b], self.c_nid[b] = \
                key, tok, par, ln, row, nid
            if n <= self.bw:                                                   # (:227-228)
                self.done[b] = True

    def finish(self):
        out = []
        for b in range(self.B):
            keys = self.c_key[b]
            i = self._order(keys, 1)[0]                                        # (:231)
            seq = [int(self.c_tok[b][i])]
            node = int(self.c_nid[b][i])
            node = int(self.c_par[b][i]) if node < 0 else self.prev[node]
            while node >= 0:                                                   # (:234-236)
                seq.append(self.tok[node])
                node = self.prev[node]
            out.append(seq[::-1])
        return out


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
