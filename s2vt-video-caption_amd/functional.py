"""Autograd glue between torch tensors and the C-ABI library (host side of the hot path).

Mirrors the reference's operator surface for this path: `train_forward` is what
`S2VT.forward(mode='train')` computes (S2VTModel.py:48-81), `greedy_decode` is `mode='test'`
(S2VTModel.py:82-110), `mean_cross_entropy` is the `nn.CrossEntropyLoss()` inside `MaskCriterion`
(utils.py:11,22).  Everything here requires HIP tensors; CPU tensors raise.
"""
import ctypes
import weakref

import torch

from . import capi


# MaskCriterion's backward fused into the model's (s2vt_mean_ce_backward_fused): on by default wherever the plane drivers run
# (B % 64 == 0).  The fp32 dlogits tensor then never exists - `logits.retain_grad()` / tensor hooks on the logits see a stride-0
# tensor of zeros; set this to False to get the materialised gradient back.
FUSE_CE = True


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_hip(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise capi.S2VTHipError(
            "%s must be a HIP (cuda) tensor: this build runs the S2VT hot path on MI355X only and has no "
            "CPU fallback (got %s)" % (name, getattr(t, "device", type(t))))


def _f32c(t, name):
    require_hip(t, name)
    if t.dtype != torch.float32:
        raise capi.S2VTHipError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _params_struct(cls, tensors):
    s = cls()
    for f, t in zip(capi.PARAM_FIELDS, tensors):
        setattr(s, f, t.data_ptr())
    return s


def _dims(feats, params):
    B, L, F = feats.shape
    H = params[8].shape[0]           # feat_linear.weight [H, F]
    V, E = params[12].shape          # embedding.weight   [V, E]
    if params[8].shape[1] != F:
        raise ValueError("feats last dim %d != feat_dim %d" % (F, params[8].shape[1]))
    return capi.Dims(B, L, F, H, E, V)


class _TrainForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, targets, grad_sink, out_mask, *params):
        lib = capi.load()
        feats = _f32c(feats, "feats")
        params = tuple(_f32c(p, "parameter") for p in params)
        require_hip(targets, "targets")
        if targets.dtype != torch.int64:
            targets = targets.long()
        if targets.dim() != 2 or targets.stride(1) != 1:
            targets = targets.reshape(targets.shape[0], -1).contiguous()
        d = _dims(feats, params)
        if targets.shape[0] != d.B or targets.shape[1] != d.L - 1:
            raise ValueError("targets must be [B, L-1] = [%d, %d], got %s" % (d.B, d.L - 1, tuple(targets.shape)))
        dev = feats.device
        with torch.cuda.device(dev):
            nbytes = lib.s2vt_train_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            logits = torch.empty(d.B, d.L - 1, d.V, dtype=torch.float32, device=dev)
            ps = _params_struct(capi.Params, params)
            if out_mask is None:
                capi.check(lib.s2vt_train_forward(ctypes.byref(d), ctypes.byref(ps), _ptr(feats), _ptr(targets),
                                                  targets.stride(0), _ptr(logits), _ptr(ws), nbytes, _stream(dev)),
                           "s2vt_train_forward")
            else:       # out_dropout > 0: mask [(L-1)*B, H] time-major, entries 0 or 1/(1-p)
                out_mask = _f32c(out_mask, "out_mask")
                if tuple(out_mask.shape) != ((d.L - 1) * d.B, d.H):
                    raise ValueError("out_mask must be [(L-1)*B, H] = [%d, %d], got %s" % ((d.L - 1) * d.B, d.H, tuple(out_mask.shape)))
                capi.check(lib.s2vt_train_forward_dropout(ctypes.byref(d), ctypes.byref(ps), _ptr(feats), _ptr(targets),
                                                          targets.stride(0), _ptr(out_mask), _ptr(logits), _ptr(ws), nbytes,
                                                          _stream(dev)), "s2vt_train_forward_dropout")
        ctx.out_mask = out_mask
        ctx.save_for_backward(feats, *params)
        ctx.ws, ctx.d, ctx.used, ctx.grad_sink = ws, d, False, grad_sink
        ctx.dlog_fused = False      # set by _MeanCE.backward when it wrote the dlogits planes into ctx.ws itself (fused CE)
        ctx.fusable = bool(FUSE_CE and d.B % 64 == 0 and lib.s2vt_set_gemm_mode(-1) in (1, 3))
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.used:
            raise capi.S2VTHipError("S2VT backward ran twice on one forward: the saved activations are consumed "
                                    "in place (retain_graph is not supported)")
        ctx.used = True
        lib = capi.load()
        feats, *params = ctx.saved_tensors
        d, ws = ctx.d, ctx.ws
        dev = feats.device
        if ctx.dlog_fused:
            # the criterion's backward already turned the logits into dlogits operand planes inside ctx.ws; what arrives here is
            # its stride-0 placeholder.  Anything else means a second consumer of the logits added its own gradient
            # (stride 0 everywhere).  A tensor hook on the logits, anomaly mode or a future autograd that makes incoming
            # gradients contiguous hands over a MATERIALISED copy of those zeros: still the hand-over, recognised by its
            # content (one reduction over [B, L-1, V], only on that rare route), not by its strides
            placeholder = dlogits.dim() == 3 and all(st == 0 for st in dlogits.stride())
            if not placeholder and bool((dlogits != 0).any()):
                raise capi.S2VTHipError("the logits of this forward fed the fused MaskCriterion backward AND another consumer: set "
                                        "s2vt_video_caption_amd.functional.FUSE_CE = False to materialise dlogits")
            dlogits = None
        else:
            dlogits = _f32c(dlogits, "dlogits")
        with torch.cuda.device(dev):
            sink = ctx.grad_sink
            if sink is not None:
                # data-parallel mode (dp.FlatGradAllReducer.attach): the 13 gradients are WRITTEN (not accumulated)
                # straight into the views of the flat all-reduce buffer; autograd gets None and leaves .grad alone
                if len(sink) != len(params) or any(g.shape != p.shape or g.dtype != torch.float32 or not g.is_contiguous()
                                                   or g.device != p.device for g, p in zip(sink, params)):
                    raise capi.S2VTHipError("gradient sink does not match the 13 S2VT parameters")
                grads = list(sink)
            else:
                grads = [torch.empty_like(p) for p in params]
            dfeats = torch.empty_like(feats) if ctx.needs_input_grad[0] else None
            ps = _params_struct(capi.Params, params)
            gs = _params_struct(capi.Grads, grads)
            if ctx.out_mask is None:
                capi.check(lib.s2vt_train_backward(ctypes.byref(d), ctypes.byref(ps), _ptr(feats), _ptr(dlogits),
                                                   ctypes.byref(gs), _ptr(dfeats), _ptr(ws), ws.numel(), _stream(dev)),
                           "s2vt_train_backward")
            else:
                capi.check(lib.s2vt_train_backward_dropout(ctypes.byref(d), ctypes.byref(ps), _ptr(feats), _ptr(dlogits),
                                                           _ptr(ctx.out_mask), ctypes.byref(gs), _ptr(dfeats), _ptr(ws),
                                                           ws.numel(), _stream(dev)), "s2vt_train_backward_dropout")
        ctx.ws = None
        if sink is not None:
            return (dfeats, None, None, None) + (None,) * len(params)
        return (dfeats, None, None, None) + tuple(grads)


# model -> 13 gradient tensors (capi.PARAM_KEYS order) the backward writes into; kept OUT of the module's __dict__ so
# that torch.save(model) checkpoints stay exactly the reference's layout
_GRAD_SINKS = weakref.WeakKeyDictionary()


def set_grad_sink(model, tensors):
    if tensors is None:
        _GRAD_SINKS.pop(model, None)
    else:
        _GRAD_SINKS[model] = tuple(tensors)


def grad_sink_for(model):
    return _GRAD_SINKS.get(model)


def train_forward(feats, targets, params, grad_sink=None, out_mask=None):
    """logits [B, L-1, V] of S2VT.forward(mode='train'); `params` in capi.PARAM_KEYS order.  `grad_sink`: optional 13
    tensors (same order) the backward writes the parameter gradients into instead of returning them to autograd.
    `out_mask`: the out_drop mask (S2VTModel.py:79) as [(L-1)*B, H] time-major, entries 0 or 1/(1-p); None = no dropout."""
    return _TrainForward.apply(feats, targets, grad_sink, out_mask, *params)


# Weight-derived images of the decode (plane images of four weight matrices, the per-token gate-input table) kept between
# mode='test' calls of one model while its parameters stand: key = every parameter's (data_ptr, _version) + dims + library
# modes + stream.  In-place updates through autograd-visible ops (optimizers, load_state_dict, .to()) change the key; writes
# through `param.data` do NOT bump a version counter - call clear_decode_cache(model) after such a write, or set
# DECODE_CACHE = False.  Kept OUT of the module (a WeakKeyDictionary) so that torch.save(model) stays the reference's layout.
DECODE_CACHE = True
_DECODE_CACHES = weakref.WeakKeyDictionary()


def clear_decode_cache(model=None):
    if model is None:
        _DECODE_CACHES.clear()
    else:
        _DECODE_CACHES.pop(model, None)


def decode_cache_entry(owner, raw_params, d, dev, lib):
    """(cache tensor, valid) of `owner`'s weight-derived decode images for the parameters as they stand, or (None, False) where
    the cache does not apply (no owner, DECODE_CACHE off, fp32-MFMA mode; any batch size - the library pads to a multiple of 64).  valid = False:
    the tensor is fresh and the next s2vt_greedy_decode_cached call must fill it (cache_valid = 0)."""
    if owner is None or not DECODE_CACHE or lib.s2vt_set_gemm_mode(-1) == 0:
        return None, False
    # (the batch size, the recurrence mode and the pipeline block are NOT in the key: a filling call writes every
    # image the cache holds, whichever of them its own batch / modes read - s2vt_greedy_decode_cached)
    key = (tuple((p.data_ptr(), p._version) for p in raw_params), (d.L, d.F, d.H, d.E, d.V), lib.s2vt_set_gemm_mode(-1),
           str(dev), torch.cuda.current_stream(dev).cuda_stream)
    entry = _DECODE_CACHES.get(owner)
    cbytes = lib.s2vt_decode_cache_bytes(ctypes.byref(d))
    # entry = [key, tensor, filled]: "the images are there" is recorded by the call that WROTE them (greedy_decode below), not
    # inferred from the entry's existence - a lookup that only allocates (the beam search asks before it decides to fill) must
    # not make the next lookup believe the fresh tensor holds anything
    valid = entry is not None and entry[0] == key and entry[1].numel() >= cbytes and entry[2]
    if entry is None or entry[0] != key or entry[1].numel() < cbytes:
        entry = [key, torch.empty(cbytes, dtype=torch.uint8, device=dev), False]
        _DECODE_CACHES[owner] = entry
    return entry[1], valid


@torch.no_grad()
def greedy_decode(feats, params, sos_ix, owner=None):
    """ids int64 [B, L-1] of S2VT.forward(mode='test').  `owner`: the module the parameters belong to (enables the
    weight-image cache above)."""
    lib = capi.load()
    feats = _f32c(feats, "feats")
    raw = params
    params = tuple(_f32c(p.detach(), "parameter") for p in params)
    d = _dims(feats, params)
    dev = feats.device
    with torch.cuda.device(dev):
        nbytes = lib.s2vt_decode_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ids = torch.empty(d.B, d.L - 1, dtype=torch.int64, device=dev)
        ps = _params_struct(capi.Params, params)
        cache, valid = decode_cache_entry(owner, raw, d, dev, lib)
        if cache is not None:
            capi.check(lib.s2vt_greedy_decode_cached(ctypes.byref(d), ctypes.byref(ps), _ptr(feats), int(sos_ix), _ptr(ids),
                                                     _ptr(ws), nbytes, _ptr(cache), cache.numel(), 1 if valid else 0,
                                                     _stream(dev)), "s2vt_greedy_decode_cached")
            entry = _DECODE_CACHES.get(owner)
            if entry is not None and entry[1] is cache and lib.s2vt_decode_uses_cache(ctypes.byref(d)):
                entry[2] = True                     # (stream-ordered: later calls on this stream see the filled images; a batch of
                                                    #  at most 16 clips decodes on the launch-per-timestep path and fills nothing)
        else:
            capi.check(lib.s2vt_greedy_decode(ctypes.byref(d), ctypes.byref(ps), _ptr(feats), int(sos_ix), _ptr(ids),
                                              _ptr(ws), nbytes, _stream(dev)), "s2vt_greedy_decode")
    return ids


@torch.no_grad()
def decode_encode(feats, params, owner, depth=0):
    """The encode phase of a decode on the plane path (s2vt_decode_encode_cached): (vid_h, vid_c, word_h, word_c, gx_dec), the
    states [B, H] S2VT.beam_search starts from and - for 0 < depth <= L-1, else None - vid_rnn's half of word_rnn's gate input for
    the first `depth` decode steps [depth, B, 4H]; fills `owner`'s weight-image cache when the weights moved.  None where the cache
    or the persistent split-precision recurrence does not apply (the caller runs its own encoder)."""
    lib = capi.load()
    feats = _f32c(feats, "feats")
    raw = params
    params = tuple(_f32c(p.detach(), "parameter") for p in params)
    d = _dims(feats, params)
    dev = feats.device
    if lib.s2vt_lstm_seq_x3_workspace_bytes(d.L, (d.B + 63) // 64 * 64, d.H) == 0 or lib.s2vt_set_recurrence_mode(-1) == 0:
        return None                                  # (the library pads the batch to a multiple of 64 itself)
    with torch.cuda.device(dev):
        cache, valid = decode_cache_entry(owner, raw, d, dev, lib)
        if cache is None:
            return None
        nbytes = lib.s2vt_decode_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty(4, d.B, d.H, dtype=torch.float32, device=dev)
        depth = int(depth) if 0 < int(depth) <= d.L - 1 else 0
        gx_dec = torch.empty(depth, d.B, 4 * d.H, dtype=torch.float32, device=dev) if depth else None
        ps = _params_struct(capi.Params, params)
        rc = lib.s2vt_decode_encode_cached(ctypes.byref(d), ctypes.byref(ps), _ptr(feats), _ptr(ws), nbytes, _ptr(cache), cache.numel(),
                                           1 if valid else 0, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(gx_dec),
                                           depth, _stream(dev))
        if rc == -1:
            return None                              # (S2VT_ERR_ARG: a shape the library's own predicate refused - before anything was
                                                     #  enqueued - capi.last_error() says why; the caller runs its own encoder)
        capi.check(rc, "s2vt_decode_encode_cached")  # anything else is a real error (HIP, time-out, bad token id): raise
        entry = _DECODE_CACHES.get(owner)
        if entry is not None and entry[1] is cache:
            entry[2] = True
    return out[0], out[1], out[2], out[3], gx_dec


def _ce_inputs(logits, target):
    logits = _f32c(logits, "logits")
    require_hip(target, "target")
    if target.dtype != torch.int64:
        target = target.long()
    if target.stride(1) != 1:
        target = target.contiguous()
    B, Lm1, V = logits.shape
    if target.shape[0] != B or target.shape[1] != Lm1 + 1:
        raise ValueError("target must be [B, L] = [%d, %d], got %s" % (B, Lm1 + 1, tuple(target.shape)))
    return logits, target


def _fusable_train_node(logits):
    """The _TrainForward node behind `logits` when the criterion's backward may write the dlogits operand planes into that
    forward's workspace instead of a [B, L-1, V] fp32 tensor (fused route), else None."""
    node = getattr(logits, "grad_fn", None)
    return node if (node is not None and getattr(node, "fusable", False) and getattr(node, "ws", None) is not None
                    and logits.is_contiguous()) else None


def _ce_backward(lib, node, logits, target, lse, g):
    """d(mean CE)/d(logits) * g (g: one device float) - fused into the train workspace of `node` where that applies."""
    B, Lm1, V = logits.shape
    dev = logits.device
    if node is not None and node.ws is not None and not node.used and not node.dlog_fused and FUSE_CE:
        with torch.cuda.device(dev):
            capi.check(lib.s2vt_mean_ce_backward_fused(ctypes.byref(node.d), _ptr(logits), _ptr(target), target.stride(0),
                                                       _ptr(lse), _ptr(g), _ptr(node.ws), node.ws.numel(), _stream(dev)),
                       "s2vt_mean_ce_backward_fused")
        node.dlog_fused = True
        return torch.zeros((), dtype=logits.dtype, device=dev).expand(B, Lm1, V)     # placeholder: no memory behind it
    with torch.cuda.device(dev):
        dlogits = torch.empty_like(logits)
        capi.check(lib.s2vt_mean_ce_backward(B, Lm1, V, _ptr(logits), _ptr(target), target.stride(0), _ptr(lse),
                                             _ptr(g), _ptr(dlogits), _stream(dev)), "s2vt_mean_ce_backward")
    return dlogits


class _MeanCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        lib = capi.load()
        logits, target = _ce_inputs(logits, target)
        B, Lm1, V = logits.shape
        dev = logits.device
        with torch.cuda.device(dev):
            scratch = torch.empty(2 * B * Lm1 + 1, dtype=torch.float32, device=dev)
            lse, rowloss, loss = scratch[:B * Lm1], scratch[B * Lm1:2 * B * Lm1], scratch[2 * B * Lm1:]
            capi.check(lib.s2vt_mean_ce_forward(B, Lm1, V, _ptr(logits), _ptr(target), target.stride(0), _ptr(lse),
                                                _ptr(rowloss), _ptr(loss), _stream(dev)), "s2vt_mean_ce_forward")
        ctx.save_for_backward(logits, target, lse)
        ctx.train_node = _fusable_train_node(logits)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits, target, lse = ctx.saved_tensors
        gout = _f32c(gout.reshape(1), "grad_output")
        return _ce_backward(capi.load(), ctx.train_node, logits, target, lse, gout), None


class _MaskCriterion(torch.autograd.Function):
    """MaskCriterion.forward (utils.py:13-26) in two launches: per-row CE, then mean / mask weighting / division."""

    @staticmethod
    def forward(ctx, logits, target, mask):
        lib = capi.load()
        logits, target = _ce_inputs(logits, target)
        require_hip(mask, "mask")
        if mask.dtype != torch.float32:
            mask = mask.float()
        if mask.dim() != 2 or mask.stride(1) != 1:
            mask = mask.reshape(mask.shape[0], -1).contiguous()
        B, Lm1, V = logits.shape
        if mask.shape[0] != B or mask.shape[1] != Lm1 + 1:
            raise ValueError("mask must be [B, L] = [%d, %d], got %s" % (B, Lm1 + 1, tuple(mask.shape)))
        dev = logits.device
        with torch.cuda.device(dev):
            scratch = torch.empty(2 * B * Lm1 + 4, dtype=torch.float32, device=dev)
            lse, rowloss, out3 = scratch[:B * Lm1], scratch[B * Lm1:2 * B * Lm1], scratch[2 * B * Lm1:]
            capi.check(lib.s2vt_mask_criterion_forward(B, Lm1, V, _ptr(logits), _ptr(target), target.stride(0), _ptr(mask),
                                                       mask.stride(0), _ptr(lse), _ptr(rowloss), _ptr(out3), _stream(dev)),
                       "s2vt_mask_criterion_forward")
        ctx.save_for_backward(logits, target, lse, mask, out3)
        ctx.train_node = _fusable_train_node(logits)
        return out3[0].reshape(())

    @staticmethod
    def backward(ctx, gout):
        lib = capi.load()
        logits, target, lse, mask, out3 = ctx.saved_tensors
        B, Lm1, V = logits.shape
        dev = logits.device
        gout = _f32c(gout.reshape(1), "grad_output")
        with torch.cuda.device(dev):
            g_ce = torch.empty(1, dtype=torch.float32, device=dev)
            capi.check(lib.s2vt_mask_criterion_backward(B, Lm1, _ptr(mask), mask.stride(0), _ptr(out3), _ptr(gout), _ptr(g_ce),
                                                        _stream(dev)), "s2vt_mask_criterion_backward")
        return _ce_backward(lib, ctx.train_node, logits, target, lse, g_ce), None, None


def mean_cross_entropy(logits, target):
    """Mean CE of logits [B, L-1, V] against target[:, 1:] (target int64 [B, L]) — utils.py:11,22."""
    return _MeanCE.apply(logits, target)


def mask_criterion(logits, target, mask):
    """MaskCriterion()(logits, target, mask) of the reference (utils.py:13-26): mean CE against target[:, 1:], weighted with
    mask[:, 1:] and divided by its sum (NaN for an all-zero mask, as upstream).  No gradient flows to the mask."""
    return _MaskCriterion.apply(logits, target, mask)
