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

from . import capi, ops

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


PERSISTENT_ENCODER = True        # A/B switch: encoder recurrences on the persistent split-precision kernels
PLANE_ENCODER = True             # A/B switch: the whole encode phase by the library (s2vt_decode_encode_cached), GEMMs on the plane path
VID_PRECOMPUTE = True            # A/B switch: vid_rnn's token-independent decode steps and their gate-input GEMM once, in front of the depth loop


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
    lib0 = capi.load()
    if (DEVICE_QUEUES and PLANE_STEP and PLANE_ENCODER and lib0.s2vt_beam_queue_bytes(B, beam_width, max_depth) > 0):
        # the library's own encode phase (what mode='test' runs: plane-path GEMMs, paired persistent recurrence launches); it
        # fills the weight-image cache the plane-path depth step reads
        from . import functional
        enc = functional.decode_encode(feats, params, model, depth=max_depth if VID_PRECOMPUTE else 0)
        if enc is not None:
            return _beam_search_device_queues(lib0, model, feats, params, B, H, beam_width, max_depth, sos, eos, *enc)
    x1 = ops.feat_proj_fwd(feats.contiguous(), w_f, b_f)                       # [L*B, H] time-major
    gx1 = _gemm_strided(x1, w_ih1, bsum1)
    # (the persistent split-precision recurrence, lstm_persist_x3.hip, where the shape is supported; else launches per timestep)
    px = capi.load().s2vt_lstm_seq_x3_workspace_bytes(L, B, H) > 0 and PERSISTENT_ENCODER

    def layer(gx, w_hh):
        if px:
            return ops.lstm_seq_fwd_persist(L, B, gx, L, None, w_hh, poison=False)
        return ops.lstm_seq_fwd(L, B, gx, L, None, w_hh)
    h1_all, c1_all, _ = layer(gx1, w_hh1)
    gx2 = _gemm_strided(h1_all, w_v, bsum2)
    h2_all, c2_all, _ = layer(gx2, w_hh2)
    vid_h, vid_c = h1_all[(L - 1) * B:], c1_all[(L - 1) * B:]
    word_h, word_c = h2_all[(L - 1) * B:].clone(), c2_all[(L - 1) * B:].clone()

    # ---- per-sample queues (host, BeamQueues) + one library call per depth (s2vt_beam_step): vid step for the batch,
    # word step / out_linear / log_softmax / top-20 for all expandable (sample, beam slot) rows
    import ctypes
    from .functional import _ptr, _stream, _dims, _params_struct
    lib = capi.load()
    max_rows = max(B * beam_width, B)
    if DEVICE_QUEUES and lib.s2vt_beam_queue_bytes(B, beam_width, max_depth) > 0:
        return _beam_search_device_queues(lib, model, feats, params, B, H, beam_width, max_depth, sos, eos, vid_h, vid_c, word_h, word_c)
    global LAST_PATH
    LAST_PATH = "host queues"
    queues = (BeamQueues if FAST_QUEUES else HeapQueues)(B, beam_width, sos, eos)
    pcs = tuple(p.contiguous() for p in (w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2, w_f, b_f, w_o, b_o, emb))
    d = _dims(feats, pcs)
    ps = _params_struct(capi.Params, pcs)
    with torch.cuda.device(dev):
        nbytes = lib.s2vt_beam_workspace_bytes(ctypes.byref(d), max_rows)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        vid = [(vid_h.contiguous(), vid_c.contiguous()), (torch.empty(B, H, device=dev), torch.empty(B, H, device=dev))]
        tab = [(torch.empty(max_rows, H, device=dev), torch.empty(max_rows, H, device=dev)) for _ in range(2)]
        tab[0][0][:B].copy_(word_h)
        tab[0][1][:B].copy_(word_c)
        rows_dev = torch.empty(3, max_rows, dtype=torch.int32, device=dev)
        top_dev = torch.empty(2, max_rows, FANOUT, dtype=torch.int32, device=dev)      # [0] ids, [1] fp32 log-prob bits
        depth = 0
        while depth < max_depth and not queues.all_done():
            depth += 1
            rows_b, rows_state, rows_tok = queues.pop()
            R = len(rows_b)
            if R:
                rows_dev[:, :R].copy_(torch.from_numpy(np.stack([np.asarray(rows_b), np.asarray(rows_state),
                                                                  np.asarray(rows_tok)]).astype(np.int32)))
            (vh_in, vc_in), (vh_out, vc_out) = vid[(depth - 1) & 1], vid[depth & 1]
            (wh_in, wc_in), (wh_out, wc_out) = tab[(depth - 1) & 1], tab[depth & 1]
            capi.check(lib.s2vt_beam_step(ctypes.byref(d), ctypes.byref(ps), R, _ptr(rows_dev[0]), _ptr(rows_dev[1]),
                                          _ptr(rows_dev[2]), _ptr(vh_in), _ptr(vc_in), _ptr(vh_out), _ptr(vc_out),
                                          _ptr(wh_in), _ptr(wc_in), _ptr(wh_out), _ptr(wc_out), _ptr(top_dev[0]),
                                          _ptr(top_dev[1]), _ptr(ws), nbytes, _stream(dev)), "s2vt_beam_step")
            top_ix = top_lp = None
            if R:
                both = top_dev[:, :R].cpu().numpy()                                     # one D2H copy per depth
                top_ix = both[0].astype(np.int64)
                top_lp = both[1].view(np.float32)
            queues.push(top_ix, top_lp)
    # list[list[Tensor]] like the reference (first element the [[<sos>]] tensor, then 0-dim ids): one H2D copy for all
    # sequences, the elements are views of it
    seqs = queues.finish()
    flat = torch.as_tensor(np.concatenate([np.asarray(q, dtype=np.int64) for q in seqs]), device=dev)
    sentences, o = [], 0
    for q in seqs:
        t = flat[o:o + len(q)]
        o += len(q)
        sentences.append([t[:1].view(1, 1)] + list(t[1:].unbind(0)))
    return sentences


FAST_QUEUES = True
DEVICE_QUEUES = True         # the queues on the device (csrc/beam_queue.hip): no host work and no transfer per depth


PLANE_STEP = True            # s2vt_beam_step_cached: the depth's GEMMs on the plane path, from the decode cache of the same weights
LAST_PATH = None             # which path the last beam_search call took (bench.py reports it beside the rate)


def _beam_search_device_queues(lib, model, feats, params, B, H, beam_width, max_depth, sos, eos, vid_h, vid_c, word_h, word_c,
                               gx_dec=None):
    """The depth loop with the queue bookkeeping on the device: per depth ONE s2vt_beam_queue_step (push the children of the
    depth before, freeze finished samples, pop the next beam with heapq's own sift order, write the rows of the step) and ONE
    s2vt_beam_step over the fixed rows r = b * beam_width + slot; nothing crosses PCIe until the back-traced sequences at the
    end.  The reference's early exit (all samples stopped, S2VTModel.py:189) is polled without blocking: the frozen-sample
    counter is copied to pinned memory after every depth and read one depth late - extra depths change nothing."""
    import ctypes
    from .functional import _ptr, _stream, _dims, _params_struct
    dev = feats.device
    R = B * beam_width
    pcs = tuple(p.detach().contiguous() for p in params)
    d = _dims(feats, pcs)
    ps = _params_struct(capi.Params, pcs)
    # the weight-derived images of mode='test' (plane images of W_v / W_o, the per-token gate table) serve the beam's depth step
    # as well; a model that has not decoded with these weights yet runs ONE greedy decode of this batch to fill them (7 ms, once
    # per weight version - eval.py decodes the whole validation set with one set of weights)
    from . import functional
    cache = None
    if PLANE_STEP:
        cache, valid = functional.decode_cache_entry(model, params, d, dev, lib)
        if cache is not None and not valid:
            functional.greedy_decode(feats, params, sos, owner=model)
            cache, valid = functional.decode_cache_entry(model, params, d, dev, lib)
            if not valid:
                cache = None
    global LAST_PATH
    if cache is None:
        gx_dec = None
    LAST_PATH = ("device queues + plane-path depth step (decode cache)" + (", vid_rnn steps precomputed" if gx_dec is not None else "")) \
        if cache is not None else "device queues + fp32-MFMA depth step"
    with torch.cuda.device(dev):
        nbytes = lib.s2vt_beam_workspace_bytes(ctypes.byref(d), R)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        qbytes = lib.s2vt_beam_queue_bytes(B, beam_width, max_depth)
        qs = torch.empty(qbytes, dtype=torch.uint8, device=dev)
        vid = [(vid_h.contiguous(), vid_c.contiguous()), (torch.empty(B, H, device=dev), torch.empty(B, H, device=dev))]
        tab = [(torch.zeros(R, H, device=dev), torch.zeros(R, H, device=dev)) for _ in range(2)]
        tab[0][0][:B].copy_(word_h)
        tab[0][1][:B].copy_(word_c)
        rows = torch.zeros(3, R, dtype=torch.int32, device=dev)
        top_ix = torch.zeros(R, FANOUT, dtype=torch.int32, device=dev)
        top_lp = torch.zeros(R, FANOUT, dtype=torch.float32, device=dev)
        frozen = torch.zeros(max_depth + 2, dtype=torch.int32).pin_memory()
        frozen_dev = qs[:4].view(torch.int32)
        events = []
        st = _stream(dev)

        def qstep(depth):
            capi.check(lib.s2vt_beam_queue_step(B, beam_width, max_depth, sos, eos, depth, _ptr(qs), qbytes, _ptr(top_ix), _ptr(top_lp),
                                                _ptr(rows[0]), _ptr(rows[1]), _ptr(rows[2]), st), "s2vt_beam_queue_step")
        depth = 0
        while depth < max_depth:
            if depth >= 2 and events[depth - 2].query() and int(frozen[depth - 1]) == B:
                break                                   # every sample had stopped two depths ago: the reference's loop ended there
            depth += 1
            qstep(depth)
            if depth > 1:
                frozen[depth].copy_(frozen_dev[0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            events.append(ev)
            (vh_in, vc_in), (vh_out, vc_out) = vid[(depth - 1) & 1], vid[depth & 1]
            (wh_in, wc_in), (wh_out, wc_out) = tab[(depth - 1) & 1], tab[depth & 1]
            if gx_dec is not None:
                capi.check(lib.s2vt_beam_step_gx(ctypes.byref(d), ctypes.byref(ps), R, _ptr(rows[0]), _ptr(rows[1]), _ptr(rows[2]),
                                                 _ptr(gx_dec[depth - 1]), _ptr(wh_in), _ptr(wc_in), _ptr(wh_out), _ptr(wc_out),
                                                 _ptr(top_ix), _ptr(top_lp), _ptr(ws), nbytes, _ptr(cache), cache.numel(), st),
                           "s2vt_beam_step_gx")
            elif cache is not None:
                capi.check(lib.s2vt_beam_step_cached(ctypes.byref(d), ctypes.byref(ps), R, _ptr(rows[0]), _ptr(rows[1]), _ptr(rows[2]),
                                                     _ptr(vh_in), _ptr(vc_in), _ptr(vh_out), _ptr(vc_out), _ptr(wh_in), _ptr(wc_in),
                                                     _ptr(wh_out), _ptr(wc_out), _ptr(top_ix), _ptr(top_lp), _ptr(ws), nbytes,
                                                     _ptr(cache), cache.numel(), st), "s2vt_beam_step_cached")
            else:
                capi.check(lib.s2vt_beam_step(ctypes.byref(d), ctypes.byref(ps), R, _ptr(rows[0]), _ptr(rows[1]), _ptr(rows[2]),
                                              _ptr(vh_in), _ptr(vc_in), _ptr(vh_out), _ptr(vc_out), _ptr(wh_in), _ptr(wc_in),
                                              _ptr(wh_out), _ptr(wc_out), _ptr(top_ix), _ptr(top_lp), _ptr(ws), nbytes, st), "s2vt_beam_step")
        qstep(0)                                        # the last depth's push
        cap = max_depth + 2
        out = torch.empty(B, cap, dtype=torch.int32, device=dev)
        out_len = torch.empty(B, dtype=torch.int32, device=dev)
        capi.check(lib.s2vt_beam_queue_result(B, beam_width, max_depth, _ptr(qs), qbytes, _ptr(out), cap, _ptr(out_len), st),
                   "s2vt_beam_queue_result")
        host = torch.cat([out, out_len[:, None]], dim=1).cpu().numpy()        # (the one synchronisation of the search)
        lens = host[:, cap].tolist()
        flat = torch.as_tensor(np.concatenate([host[b, :lens[b]] for b in range(B)]).astype(np.int64), device=dev)
    sentences, o = [], 0
    for b in range(B):
        t = flat[o:o + lens[b]]
        o += lens[b]
        sentences.append([t[:1].view(1, 1)] + list(t[1:].unbind(0)))
    return sentences


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
    """Same bookkeeping as HeapQueues, vectorised over samples AND candidates (numpy).

    Because the queue is emptied after every pop phase (:194), a sample's heap at depth d is just "the candidates
    pushed at depth d-1, in push order"; popping beam_width entries is a partial sort by key.  Distinct keys sort the
    same way in any heap, so the fast path is one stable argsort over a [B, beam_width*20] key matrix; when two of a
    sample's beam_width+1 smallest keys are EQUAL the pop order depends on the heap's internal layout, and that sample
    falls back to replaying its pushes into a real heap (_heap_order).  Nodes are materialised only when popped
    (<= beam_width per sample and depth instead of 20 x beam_width).
    """

    def __init__(self, B, beam_width, sos, eos):
        self.B, self.bw, self.eos = B, beam_width, eos
        self.NC = max(beam_width * FANOUT, 1)
        # node table (back-trace): token and parent of every popped node; the B roots come first
        self.tok_tab = [np.full(B, sos, dtype=np.int64)]
        self.prev_tab = [np.full(B, -1, dtype=np.int64)]
        self.n_nodes = B
        NC = self.NC
        # candidates of every sample in push order, padded to NC: key (+inf pad), token, parent node, length, state row,
        # node id (-1: not materialised yet)
        self.key = np.full((B, NC), np.inf)
        self.tok = np.zeros((B, NC), dtype=np.int64)
        self.par = np.full((B, NC), -1, dtype=np.int64)
        self.len = np.ones((B, NC), dtype=np.int64)
        self.row = np.zeros((B, NC), dtype=np.int64)
        self.nid = np.full((B, NC), -1, dtype=np.int64)
        self.n = np.ones(B, dtype=np.int64)
        self.key[:, 0] = -0.0
        self.tok[:, 0] = sos
        self.row[:, 0] = np.arange(B)
        self.nid[:, 0] = np.arange(B)
        self.done = np.zeros(B, dtype=bool)
        self.tie_fallbacks = 0
        # score divisor len**0.7 exactly as the reference evaluates it: Python float pow, then fp32 (S2VTModel.py:262-266)
        self.pow07 = np.array([np.float32(pow(float(l), 0.7)) if l > 0 else np.float32(1) for l in range(4096)],
                              dtype=np.float32)

    def all_done(self):
        return bool(self.done.all())

    def _tab(self):
        if len(self.tok_tab) > 1:
            self.tok_tab = [np.concatenate(self.tok_tab)]
            self.prev_tab = [np.concatenate(self.prev_tab)]
        return self.tok_tab[0], self.prev_tab[0]

    def _pop_order(self, m_want):
        """[B, m_want] candidate indices in pop order (entries j >= min(m_want, n_b) are meaningless)."""
        B = self.B
        k = min(m_want + 1, self.NC)
        order = np.argsort(self.key, axis=1, kind="stable")[:, :k]
        ks = np.take_along_axis(self.key, order, axis=1)
        # a tie matters only among the first min(m_want, n) + 1 VALID candidates
        pos = np.arange(1, k)[None, :]
        tie = ((ks[:, 1:] == ks[:, :-1]) & (pos < self.n[:, None]) & (pos <= m_want)).any(axis=1) & ~self.done
        order = order[:, :m_want] if order.shape[1] >= m_want else np.pad(order, ((0, 0), (0, m_want - order.shape[1])))
        for b in np.nonzero(tie)[0]:
            self.tie_fallbacks += 1
            nb = int(self.n[b])
            o = _heap_order(self.key[b, :nb].tolist(), min(m_want, nb))
            order[b, :len(o)] = o
        return order

    def pop(self):
        B, bw = self.B, self.bw
        act = ~self.done
        order = self._pop_order(bw)                                         # [B, bw]
        valid = (np.arange(bw)[None, :] < np.minimum(self.n, bw)[:, None]) & act[:, None]
        g = lambda arr: np.take_along_axis(arr, order, axis=1)
        key, tok, par, ln, row, nid = g(self.key), g(self.tok), g(self.par), g(self.len), g(self.row), g(self.nid)
        # materialise the popped candidates that are not nodes yet, in (sample, beam slot) order
        new = valid & (nid < 0)
        nnew = int(new.sum())
        if nnew:
            nid = nid.copy()
            nid[new] = self.n_nodes + np.arange(nnew)
            self.tok_tab.append(tok[new])
            self.prev_tab.append(par[new])
            self.n_nodes += nnew
        tok_tab, prev_tab = self._tab()
        prev = np.where(valid, prev_tab[np.where(valid, nid, 0)], -1)
        fin = valid & (tok == self.eos) & (prev >= 0)
        exp = valid & ~fin
        self.beam = (key, nid, ln, fin, exp, valid)
        bidx = np.broadcast_to(np.arange(B)[:, None], (B, bw))
        return bidx[exp], row[exp], tok[exp]

    def push(self, top_ix, top_lp):
        B, bw, NC, F = self.B, self.bw, self.NC, FANOUT
        key, nid, ln, fin, exp, valid = self.beam
        _, prev_tab = self._tab()
        size = np.where(exp, F, 0) + fin.astype(np.int64)                   # candidates each beam entry pushes
        off = np.cumsum(size, axis=1) - size                                # their first slot, in beam order
        n_new = size.sum(axis=1)
        act = valid.any(axis=1)
        nk = np.full((B, NC), np.inf)
        nt = np.zeros((B, NC), dtype=np.int64)
        npar = np.full((B, NC), -1, dtype=np.int64)
        nl = np.ones((B, NC), dtype=np.int64)
        nrow = np.zeros((B, NC), dtype=np.int64)
        nnid = np.full((B, NC), -1, dtype=np.int64)
        bidx = np.broadcast_to(np.arange(B)[:, None], (B, bw))
        if fin.any():                                                       # finished entries re-inserted unchanged (:200-202)
            fb, fo = bidx[fin], off[fin]
            nk[fb, fo] = key[fin]
            nt[fb, fo] = self.eos
            npar[fb, fo] = prev_tab[nid[fin]]
            nl[fb, fo] = ln[fin]
            nrow[fb, fo] = -1
            nnid[fb, fo] = nid[fin]
        if exp.any():                                                       # 20 children per expanded entry (:216-223)
            eb, eo, el, en = bidx[exp], off[exp], ln[exp] + 1, nid[exp]
            R = len(eb)
            cols = eo[:, None] + np.arange(F)[None, :]
            rows = eb[:, None]
            # score = fp32 log-prob / fp32(len**0.7); key = -score
            sc = np.asarray(top_lp, dtype=np.float32)[:R] / self.pow07[el][:, None]
            nk[rows, cols] = -sc.astype(np.float64)
            nt[rows, cols] = np.asarray(top_ix)[:R]
            npar[rows, cols] = en[:, None]
            nl[rows, cols] = el[:, None]
            nrow[rows, cols] = np.arange(R)[:, None]
        for dst, src in ((self.key, nk), (self.tok, nt), (self.par, npar), (self.len, nl), (self.row, nrow), (self.nid, nnid)):
            dst[act] = src[act]
        self.n[act] = n_new[act]
        self.done |= act & (n_new <= bw)                                    # (:227-228)

    def finish(self):
        tok_tab, prev_tab = self._tab()
        first = self._pop_order_final()
        out = []
        for b in range(self.B):
            i = int(first[b])                                               # (:231)
            seq = [int(self.tok[b, i])]
            node = int(self.nid[b, i])
            node = int(self.par[b, i]) if node < 0 else int(prev_tab[node])
            while node >= 0:                                                # (:234-236)
                seq.append(int(tok_tab[node]))
                node = int(prev_tab[node])
            out.append(seq[::-1])
        return out

    def _pop_order_final(self):
        saved, self.done = self.done, np.zeros(self.B, dtype=bool)          # frozen samples still pop their best entry
        try:
            return self._pop_order(1)[:, 0]
        finally:
            self.done = saved


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
