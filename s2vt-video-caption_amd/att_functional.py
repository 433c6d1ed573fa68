"""Autograd glue for the reference's second network, `Att_Baseline` (attention_baseline.py:9-105), over the per-op C-ABI entry
points of libs2vt_hip.so: the projections and their gradients are the split-precision plane GEMMs of the S2VT path (s2vt_split_planes +
s2vt_gemm_bf16_nt / _tt; s2vt_gemm_f32 outside gemm mode 3), every recurrence s2vt_lstm_seq_fwd / s2vt_lstm_seq_bwd (the fused
timestep kernels of the S2VT path), the greedy loop s2vt_lstm_step_fwd + s2vt_decode_step_argmax.  PyTorch only holds the
tensors and wires the autograd graph (the callers / model variants either side of the hot path).

What the reference network computes (and what therefore is built here): a bidirectional LSTM encoder over the projected
frames, a context vector, an LSTM decoder over [Emb[word] ‖ context], out_linear.  Its attention weights are
`softmax(et, dim=2)` of a [B, L, 1] tensor (attention_baseline.py:53-55): a softmax over a dimension of size ONE, i.e. every
weight is exactly 1, so the context is the SUM of the encoder outputs over the frames, the same vector at every decode step,
and the three attention layers receive exactly-zero gradients.  The drop-in reproduces that arithmetic, not the intention.
"""
import torch

from . import capi, ops


PLANE_GEMMS = True       # A/B switch: the projections on the split-precision plane path (fp32-equivalent, bf16 matrix cores)


def _plane_mode():
    """The library's batched-GEMM arithmetic is split precision (s2vt_set_gemm_mode 3: three bf16 planes per operand, six plane
    products, fp32-equivalent); modes 0 (exact-fp32 MFMA) and 1 (bf16 operands) keep s2vt_gemm_f32 here."""
    return PLANE_GEMMS and capi.load().s2vt_set_gemm_mode(-1) == 3


def _dw_planes(dy, x_planes, N, K):
    """dW [N, K] = dy^T x from the ROW plane images of dy [M, N] and x [M, K] (transposed fragment reads: no transposed copy);
    M is padded to the images' 64-row blocks, whose padding rows are zero."""
    pdy = ops.split_planes(dy)
    mpad = pdy[0].shape[0]
    return ops.gemm_planes_tt(pdy, x_planes, N, K, mpad), pdy


class _Affine(torch.autograd.Function):
    """y = x W^T (+ b) with x [M, K], W [N, K]: forward and both gradients on the split-precision plane GEMMs of the S2VT path
    (gemm_x3.hip: fp32-equivalent) when the library is in that mode, else on s2vt_gemm_f32 (exact-fp32 MFMA GEMM)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = x.contiguous(), w.contiguous()
        ctx.has_bias = b is not None
        ctx.planes = _plane_mode()
        if ctx.planes:
            px = ops.split_planes(x)
            y = ops.gemm_planes(px, ops.split_planes(w), x.shape[0], w.shape[0], bias=b.contiguous() if b is not None else None)
            ctx.save_for_backward(w, px[0])
            ctx.px_meta, ctx.xshape = px[1:], tuple(x.shape)
            return y
        ctx.save_for_backward(x, w)
        return ops.gemm(x, w, bias=b.contiguous() if b is not None else None)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if ctx.planes:
            w, px0 = ctx.saved_tensors
            (M, K), N = ctx.xshape, w.shape[0]
            px = (px0,) + tuple(ctx.px_meta)
            pdy = None
            dw = None
            if ctx.needs_input_grad[1]:
                dw, pdy = _dw_planes(dy, px, N, K)
            dx = None
            if ctx.needs_input_grad[0]:
                pdy = pdy if pdy is not None else ops.split_planes(dy)
                dx = ops.gemm_planes(pdy, ops.split_planes(w, transpose=True), M, K)            # [M,N]·[N,K]
            return dx, dw, db
        x, w = ctx.saved_tensors
        dx = ops.gemm(dy, w, b_kmajor=False) if ctx.needs_input_grad[0] else None            # [M,N]·[N,K]
        dw = ops.gemm(dy, x, a_kmajor=False, b_kmajor=False) if ctx.needs_input_grad[1] else None   # [N,M]·[M,K]
        return dx, dw, db


class _LstmLayer(torch.autograd.Function):
    """h_all [T*B, H] = one LSTM layer from the zero state over time-major gate inputs gx [T*B, 4H] (= x W_ih^T + b_ih + b_hh):
    the launch-per-timestep kernels of the S2VT path; BPTT returns dG = d gx, dW_hh = sum_t dG_t^T h_{t-1}."""

    @staticmethod
    def forward(ctx, gx, w_hh, T, B):
        w_hh = w_hh.contiguous()
        h, c, stash = ops.lstm_seq_fwd(T, B, gx.contiguous(), T, None, w_hh, want_stash=True)
        ctx.save_for_backward(w_hh, h, c, stash)
        ctx.T, ctx.B, ctx.used = T, B, False
        return h

    @staticmethod
    def backward(ctx, dh):
        if ctx.used:
            raise RuntimeError("the saved gate stash of this LSTM layer was consumed by an earlier backward (retain_graph is not supported)")
        ctx.used = True
        w_hh, h, c, stash = ctx.saved_tensors
        T, B = ctx.T, ctx.B
        dg = ops.lstm_seq_bwd(T, B, w_hh, dh.contiguous(), 0, c, stash)          # in place of the stash
        if T > 1:
            dw = ops.gemm(dg[B:], h[:-B], a_kmajor=False, b_kmajor=False)         # [4H, (T-1)B]·[(T-1)B, H]
        else:
            dw = torch.zeros_like(w_hh)
        return dg, dw, None, None


def affine(x, w, b=None):
    return _Affine.apply(x, w, b)


def lstm_layer(gx, w_hh, T, B):
    return _LstmLayer.apply(gx, w_hh, T, B)


def _rnn_params(rnn, suffix=""):
    return (getattr(rnn, "weight_ih_l0" + suffix), getattr(rnn, "weight_hh_l0" + suffix),
            getattr(rnn, "bias_ih_l0" + suffix), getattr(rnn, "bias_hh_l0" + suffix))


def encode(model, feats):
    """feat_linear + bidirectional encoder (attention_baseline.py:63-66) -> time-major encoder outputs [L, B, 2H]."""
    B, L, F = feats.shape
    H = model.dim_hid
    x = model.feat_drop(feats)
    x1 = affine(x.reshape(B * L, F), model.feat_linear.weight, model.feat_linear.bias)        # batch-major rows
    x_tm = x1.view(B, L, H).transpose(0, 1).reshape(L * B, H)                                   # time-major
    outs = []
    for suffix, flip in (("", False), ("_reverse", True)):
        w_ih, w_hh, b_ih, b_hh = _rnn_params(model.encoder, suffix)
        xin = x_tm.view(L, B, H).flip(0).reshape(L * B, H) if flip else x_tm
        h = lstm_layer(affine(xin, w_ih, b_ih + b_hh), w_hh, L, B).view(L, B, H)
        outs.append(h.flip(0) if flip else h)                       # the reverse direction's output at frame l: state after l..L-1
    return torch.cat(outs, dim=2)


def context_of(model, enc_tm):
    """attention_baseline.py:35-57 as it computes: softmax over a size-one dimension -> all weights 1 -> the sum over the frames;
    the attention layers take part with an exactly-zero contribution so that their gradients are zeros, not None, as in the
    reference's autograd graph."""
    zero = sum(p.sum() for p in (model.att_enc.weight, model.att_enc.bias, model.att_prev_hid.weight, model.att_prev_hid.bias,
                                 model.att_apply.weight)) * 0.0
    return enc_tm.sum(0) + zero


def train_forward(model, feats, targets):
    """mode='train' (attention_baseline.py:69-84): logits [B, L-1, V]."""
    B, L, _ = feats.shape
    H, E = model.dim_hid, model.dim_embed
    T = L - 1
    if targets.shape[0] != B or targets.shape[1] < T:
        raise ValueError("targets must be [B, length-1]")
    ctxv = context_of(model, encode(model, feats))                                             # [B, 2H]
    w_ih, w_hh, b_ih, b_hh = _rnn_params(model.decoder)
    emb_tm = model.embedding(targets[:, :T]).transpose(0, 1).reshape(T * B, E)                 # time-major rows
    gx = affine(emb_tm, w_ih[:, :E], b_ih + b_hh).view(T, B, 4 * H) + affine(ctxv, w_ih[:, E:], None).unsqueeze(0)
    h = lstm_layer(gx.reshape(T * B, 4 * H), w_hh, T, B)
    logits = affine(model.out_drop(h), model.out_linear.weight, model.out_linear.bias)          # [(L-1)B, V] time-major
    return logits.view(T, B, -1).transpose(0, 1)


@torch.no_grad()
def greedy_decode(model, feats):
    """mode='test' (attention_baseline.py:85-104): ids [B, L] - L steps, the first input is <sos>."""
    B, L, _ = feats.shape
    H, E = model.dim_hid, model.dim_embed
    ctxv = context_of(model, encode(model, feats))
    w_ih, w_hh, b_ih, b_hh = _rnn_params(model.decoder)
    gctx = ops.gemm(ctxv.contiguous(), w_ih[:, E:].contiguous(), bias=(b_ih + b_hh).contiguous())   # constant over the steps
    w_e = w_ih[:, :E].contiguous()
    w_hh = w_hh.contiguous()
    wo, bo = model.out_linear.weight.contiguous(), model.out_linear.bias.contiguous()
    tok = torch.full((B,), int(model.sos_ix), dtype=torch.long, device=feats.device)
    h = c = None
    preds = []
    for _ in range(L):
        gx = ops.gemm(model.embedding(tok).contiguous(), w_e) + gctx
        h, c = ops.lstm_step_fwd(gx, None, w_hh, h, c)
        tok = ops.decode_step_argmax(h, wo, bo)
        preds.append(tok)
    capi.check_async_error(wait=False)
    return torch.stack(preds, dim=1)
