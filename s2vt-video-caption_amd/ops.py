"""Thin tensor-level wrappers over the per-op C-ABI entry points (include/s2vt_hip.h).

Used by the beam-search driver and by the GPU parity tests; no autograd here.  All tensors must be
contiguous float32 HIP tensors unless noted; outputs are allocated by torch.
"""
import ctypes

import torch

from . import capi
from .functional import _ptr, _stream, _f32c


def gemm(a, b, bias=None, a_kmajor=True, b_kmajor=True, out=None, accumulate=False):
    """out[M,N] (+)= op(a)·op(b) (+bias).  a: [M,K] if a_kmajor else [K,M]; b: [N,K] if b_kmajor else [K,N]."""
    lib = capi.load()
    a, b = _f32c(a, "a"), _f32c(b, "b")
    M, K = (a.shape if a_kmajor else (a.shape[1], a.shape[0]))
    N = b.shape[0] if b_kmajor else b.shape[1]
    Kb = b.shape[1] if b_kmajor else b.shape[0]
    if K != Kb:
        raise ValueError("gemm: inner dims differ (%d vs %d)" % (K, Kb))
    dev = a.device
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(M, N, dtype=torch.float32, device=dev)
            if accumulate:
                out.zero_()
        capi.check(lib.s2vt_gemm_f32(int(a_kmajor), int(b_kmajor), M, N, K, _ptr(a), a.stride(0), _ptr(b),
                                     b.stride(0), _ptr(out), out.stride(0), _ptr(bias), int(accumulate),
                                     _stream(dev)), "s2vt_gemm_f32")
    return out


def split_planes(x, nplanes=3, transpose=False):
    """fp32 [rows, cols] -> bf16 planes of a k-major GEMM operand (include/s2vt_hip.h: blocked layout for 3 planes, plain
    rows for 1).  Returns (planes int16 [ceil64(operand rows), nplanes*kpad], ldo, kpad)."""
    lib = capi.load()
    x = _f32c(x, "x")
    rows, cols = x.shape
    orows, k = (cols, rows) if transpose else (rows, cols)
    kpad = (k + 63) // 64 * 64
    ldo = nplanes * kpad
    dev = x.device
    with torch.cuda.device(dev):
        out = torch.zeros((orows + 63) // 64 * 64, ldo, dtype=torch.int16, device=dev)
        capi.check(lib.s2vt_split_planes(nplanes, int(transpose), _ptr(x), x.stride(0), rows, cols, _ptr(out), ldo, kpad,
                                         orows, _stream(dev)), "s2vt_split_planes")
    return out, ldo, kpad


def gemm_planes_tt(pa, pb, M, N, K, bias=None, out=None, accumulate=False, splitk_ws=None, nplanes=3):
    """out[M,N] (+)= X_A^T X_B from the ROW plane images (split_planes(x, nplanes)) of X_A [K, M] and X_B [K, N]; K % 64 == 0."""
    lib = capi.load()
    (a, lda, _), (b, ldb, _) = pa, pb
    dev = a.device
    with torch.cuda.device(dev):
        if out is None:
            out = torch.zeros(M, N, dtype=torch.float32, device=dev) if accumulate else \
                torch.empty(M, N, dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_gemm_bf16_tt(nplanes, M, N, K, _ptr(a), lda, _ptr(b), ldb, _ptr(out), out.stride(0), _ptr(bias),
                                         int(accumulate), _ptr(splitk_ws),
                                         ctypes.c_size_t(splitk_ws.numel() if splitk_ws is not None else 0), _stream(dev)),
                   "s2vt_gemm_bf16_tt")
    return out


def gemm_planes(pa, pb, M, N, nplanes=3, bias=None, out=None, accumulate=False, splitk_ws=None):
    """out[M,N] (+)= A·B^T (+bias) from operands written by split_planes (same nplanes, same kpad)."""
    lib = capi.load()
    (a, lda, ka), (b, ldb, kb) = pa, pb
    if ka != kb:
        raise ValueError("gemm_planes: operands have different padded k (%d vs %d)" % (ka, kb))
    dev = a.device
    with torch.cuda.device(dev):
        if out is None:
            out = torch.zeros(M, N, dtype=torch.float32, device=dev) if accumulate else \
                torch.empty(M, N, dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_gemm_bf16_nt(nplanes, M, N, ka, _ptr(a), lda, _ptr(b), ldb, _ptr(out), out.stride(0), _ptr(bias),
                                         int(accumulate), _ptr(splitk_ws),
                                         ctypes.c_size_t(splitk_ws.numel() if splitk_ws is not None else 0), _stream(dev)),
                   "s2vt_gemm_bf16_nt")
    return out


def feat_proj_fwd(feats, w, bias):
    """x1 time-major [L*B, H] = feats[B,L,F]·w^T + bias."""
    lib = capi.load()
    feats, w = _f32c(feats, "feats"), _f32c(w, "w")
    B, L, F = feats.shape
    H = w.shape[0]
    d = capi.Dims(B, L, F, H, 1, 1)
    dev = feats.device
    with torch.cuda.device(dev):
        x1 = torch.empty(L * B, H, dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_feat_proj_fwd(ctypes.byref(d), _ptr(feats), _ptr(w), _ptr(bias), _ptr(x1), _stream(dev)),
                   "s2vt_feat_proj_fwd")
    return x1


def feat_proj_bwd(feats, w, dx1, need_dfeats=False):
    lib = capi.load()
    feats, w, dx1 = _f32c(feats, "feats"), _f32c(w, "w"), _f32c(dx1, "dx1")
    B, L, F = feats.shape
    H = w.shape[0]
    d = capi.Dims(B, L, F, H, 1, 1)
    dev = feats.device
    with torch.cuda.device(dev):
        dw = torch.empty_like(w)
        db = torch.empty(H, dtype=torch.float32, device=dev)
        dfe = torch.empty_like(feats) if need_dfeats else None
        ws = torch.empty(max(1, lib.s2vt_colsum_ws_floats(L * B, H)), dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_feat_proj_bwd(ctypes.byref(d), _ptr(feats), _ptr(w), _ptr(dx1), _ptr(dw), _ptr(db),
                                          _ptr(dfe), _ptr(ws), _stream(dev)), "s2vt_feat_proj_bwd")
    return dw, db, dfe


def lstm_step_fwd(gx, bias, w_hh, h_prev, c_prev, want_stash=False):
    """One LSTM step; gx [B,4H] or None (then bias [4H]); h_prev/c_prev [B,H] or None."""
    lib = capi.load()
    w_hh = _f32c(w_hh, "w_hh")
    H = w_hh.shape[1]
    B = gx.shape[0] if gx is not None else h_prev.shape[0]
    dev = w_hh.device
    with torch.cuda.device(dev):
        h = torch.empty(B, H, dtype=torch.float32, device=dev)
        c = torch.empty(B, H, dtype=torch.float32, device=dev)
        stash = torch.empty(B, 4 * H, dtype=torch.float32, device=dev) if want_stash else None
        capi.check(lib.s2vt_lstm_step_fwd(B, H, _ptr(gx), _ptr(bias), _ptr(w_hh), _ptr(h_prev), _ptr(c_prev), _ptr(h),
                                          _ptr(c), _ptr(stash), _stream(dev)), "s2vt_lstm_step_fwd")
    return (h, c, stash) if want_stash else (h, c)


def lstm_step_fwd_token(gx, w_hh, h_prev, c_prev, emb, w_ih, tok=None, tok_packed=None, tok_const=0):
    """One decode step of word_rnn (S2VTModel.py:100-103): gx [B,4H] (vid_out half of the gate input + biases), emb [V,E],
    w_ih [4H, E+H] (its first E columns multiply the embedded word).  Token per row: `tok` int32 [B], else `tok_packed` (the
    packed argmax words of decode_step_argmax), else `tok_const`.  Ids outside [0, V) raise IndexError at the next
    capi.check_async_error()."""
    lib = capi.load()
    gx, w_hh, emb, w_ih = _f32c(gx, "gx"), _f32c(w_hh, "w_hh"), _f32c(emb, "emb"), _f32c(w_ih, "w_ih")
    B, H = gx.shape[0], w_hh.shape[1]
    V, E = emb.shape
    dev = gx.device
    with torch.cuda.device(dev):
        h = torch.empty(B, H, dtype=torch.float32, device=dev)
        c = torch.empty(B, H, dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_lstm_step_fwd_token(B, H, E, V, _ptr(gx), _ptr(w_hh), _ptr(h_prev), _ptr(c_prev), _ptr(emb), _ptr(w_ih),
                                                w_ih.stride(0), _ptr(tok), _ptr(tok_packed), int(tok_const), _ptr(h), _ptr(c),
                                                _stream(dev)), "s2vt_lstm_step_fwd_token")
    return h, c


def lstm_step_bwd(dg_next, w_hh_t, dh_out, stash, c, c_prev, dc, dc_is_zero):
    lib = capi.load()
    B, H4 = stash.shape
    H = H4 // 4
    dev = stash.device
    with torch.cuda.device(dev):
        dg = torch.empty_like(stash)
        capi.check(lib.s2vt_lstm_step_bwd(B, H, _ptr(dg_next), _ptr(w_hh_t), _ptr(dh_out), _ptr(stash), _ptr(c),
                                          _ptr(c_prev), _ptr(dc), int(dc_is_zero), _ptr(dg), _stream(dev)),
                   "s2vt_lstm_step_bwd")
    return dg


def lstm_seq_fwd(T, B, gx, n_gx, bias, w_hh, want_stash=False):
    """Whole layer from zero state; gx [n_gx*B, 4H] time-major (overwritten by the stash if want_stash)."""
    lib = capi.load()
    w_hh = _f32c(w_hh, "w_hh")
    H = w_hh.shape[1]
    dev = w_hh.device
    with torch.cuda.device(dev):
        h_all = torch.empty(T * B, H, dtype=torch.float32, device=dev)
        c_all = torch.empty(T * B, H, dtype=torch.float32, device=dev)
        stash = None
        if want_stash:
            stash = torch.empty(T * B, 4 * H, dtype=torch.float32, device=dev)
            if n_gx:
                stash[:n_gx * B].copy_(gx)
            gx = stash
        capi.check(lib.s2vt_lstm_seq_fwd(T, B, H, _ptr(gx), n_gx, _ptr(bias), _ptr(w_hh), _ptr(h_all), _ptr(c_all),
                                         _ptr(stash), _stream(dev)), "s2vt_lstm_seq_fwd")
    return h_all, c_all, stash


def lstm_seq_bwd(T, B, w_hh, dh_out, dh_first, c_all, stash):
    """BPTT over a layer; returns dG [T*B,4H] (stash is consumed in place)."""
    lib = capi.load()
    w_hh = _f32c(w_hh, "w_hh")
    H = w_hh.shape[1]
    dev = w_hh.device
    with torch.cuda.device(dev):
        wt = torch.empty(H, 4 * H, dtype=torch.float32, device=dev)
        dc = torch.empty(B, H, dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_lstm_seq_bwd(T, B, H, _ptr(w_hh), _ptr(dh_out), dh_first, _ptr(c_all), _ptr(stash),
                                         _ptr(wt), _ptr(dc), _stream(dev)), "s2vt_lstm_seq_bwd")
    return stash


def decode_step_argmax(h, w_out, b_out, planes=False):
    """token ids int64 [B] = argmax_v (h·w_out^T + b_out), lowest index on ties.  planes: the bf16-matrix-core kernel on
    3-plane operands (s2vt_decode_step_argmax_x3) instead of the fp32-input MFMA one."""
    lib = capi.load()
    h, w_out = _f32c(h, "h"), _f32c(w_out, "w_out")
    B, H = h.shape
    V = w_out.shape[0]
    dev = h.device
    with torch.cuda.device(dev):
        packed = torch.zeros(B, dtype=torch.int64, device=dev)
        if planes:
            nbytes = lib.s2vt_decode_step_argmax_x3_workspace_bytes(B, H, V)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            capi.check(lib.s2vt_decode_step_argmax_x3(B, H, V, _ptr(h), _ptr(w_out), _ptr(b_out), _ptr(packed), _ptr(ws), nbytes,
                                                      _stream(dev)), "s2vt_decode_step_argmax_x3")
            return 0xFFFFFFFF - (packed & 0xFFFFFFFF)
        capi.check(lib.s2vt_decode_step_argmax(B, H, V, _ptr(h), _ptr(w_out), _ptr(b_out), _ptr(packed), _stream(dev)),
                   "s2vt_decode_step_argmax")
        return 0xFFFFFFFF - (packed & 0xFFFFFFFF)


def beam_step(params, dims, row_b, row_state, tok, vid_h, vid_c, word_h, word_c):
    """One s2vt_beam_step call (include/s2vt_hip.h).  params: 13 tensors in capi.PARAM_KEYS order; dims = (B,L,F,H,E,V);
    row_b / row_state / tok: int32 [R].  Returns (vid_h', vid_c', word_h' [R,H], word_c' [R,H], top_ix [R,20] int32,
    top_lp [R,20] fp32)."""
    from .functional import _params_struct
    lib = capi.load()
    d = capi.Dims(*dims)
    B, H = dims[0], dims[3]
    R = int(row_b.numel())
    dev = vid_h.device
    with torch.cuda.device(dev):
        nbytes = lib.s2vt_beam_workspace_bytes(ctypes.byref(d), max(R, 1))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        vh, vc = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
        wh, wc = torch.empty(max(R, 1), H, device=dev), torch.empty(max(R, 1), H, device=dev)
        tix = torch.empty(max(R, 1), 20, dtype=torch.int32, device=dev)
        tlp = torch.empty(max(R, 1), 20, dtype=torch.float32, device=dev)
        ps = _params_struct(capi.Params, tuple(_f32c(p, "parameter") for p in params))
        capi.check(lib.s2vt_beam_step(ctypes.byref(d), ctypes.byref(ps), R, _ptr(row_b), _ptr(row_state), _ptr(tok),
                                      _ptr(_f32c(vid_h, "vid_h")), _ptr(_f32c(vid_c, "vid_c")), _ptr(vh), _ptr(vc),
                                      _ptr(_f32c(word_h, "word_h")), _ptr(_f32c(word_c, "word_c")), _ptr(wh), _ptr(wc),
                                      _ptr(tix), _ptr(tlp), _ptr(ws), nbytes, _stream(dev)), "s2vt_beam_step")
    return vh, vc, wh[:R], wc[:R], tix[:R], tlp[:R]


def lstm_seq_fwd_bf16(gx, n_gx, bias, w_hh, T, B, H, persistent=False, block=0):
    """Config-3 arithmetic of one LSTM layer (bf16 operands, fp32 accumulate / cell state): returns (h_all, c_all, gates)
    [T*B,H], [T*B,H], [T*B,4H].  `gx` [T*B,4H] is left untouched (the kernels work on a copy: the gate stash is written
    in place of the gate input).  persistent: one launch per `block` steps (0 = all)."""
    lib = capi.load()
    dev = w_hh.device
    stash = _f32c(gx, "gx").clone()
    w_hh = _f32c(w_hh, "w_hh")
    with torch.cuda.device(dev):
        nbytes = lib.s2vt_lstm_seq_bf16_workspace_bytes(T, B, H)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        h_all = torch.empty(T * B, H, dtype=torch.float32, device=dev)
        c_all = torch.empty(T * B, H, dtype=torch.float32, device=dev)
        capi.check(lib.s2vt_lstm_seq_fwd_bf16(T, B, H, _ptr(stash), int(n_gx), _ptr(bias), _ptr(w_hh), _ptr(h_all),
                                              _ptr(c_all), _ptr(ws), nbytes, int(persistent), int(block), _stream(dev)),
                   "s2vt_lstm_seq_fwd_bf16")
        if int(ws[:4].view(torch.int32)[0].item()) != 0:
            raise capi.S2VTHipError("persistent recurrence: a hand-off wait timed out (workgroups not co-resident?)")
    return h_all, c_all, stash


def lstm_seq_fwd_bf16_pair(gx0, gx1, n_gx, bias0, bias1, w0, w1, T, B, H, block=0):
    """Two independent layers, every block of timesteps of both in ONE persistent launch.  Returns two (h_all, c_all, gates)."""
    lib = capi.load()
    dev = w0.device
    st0, st1 = _f32c(gx0, "gx0").clone(), _f32c(gx1, "gx1").clone()
    w0, w1 = _f32c(w0, "w0"), _f32c(w1, "w1")
    with torch.cuda.device(dev):
        nbytes = 2 * lib.s2vt_lstm_seq_bf16_workspace_bytes(T, B, H)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        outs = [torch.empty(T * B, H, dtype=torch.float32, device=dev) for _ in range(4)]
        capi.check(lib.s2vt_lstm_seq_fwd_bf16_pair(T, B, H, _ptr(st0), _ptr(st1), int(n_gx), _ptr(bias0), _ptr(bias1), _ptr(w0),
                                                   _ptr(w1), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]), _ptr(ws),
                                                   nbytes, int(block), _stream(dev)), "s2vt_lstm_seq_fwd_bf16_pair")
        if int(ws[:4].view(torch.int32)[0].item()) != 0:
            raise capi.S2VTHipError("persistent recurrence: a hand-off wait timed out (workgroups not co-resident?)")
    return (outs[0], outs[2], st0), (outs[1], outs[3], st1)


def _check_persist_err(ws):
    if int(ws[:4].view(torch.int32)[0].item()) != 0:
        raise capi.S2VTHipError("persistent recurrence: a hand-off wait timed out (workgroups not co-resident?)")


def lstm_seq_bwd_bf16(w_hh, dh_out, dh_first, c_all, gates, T, B, H, persistent=False, block=0):
    """Config-3 arithmetic of one layer's BPTT: returns fp32 dG [T*B,4H] (`gates` is left untouched)."""
    lib = capi.load()
    dev = w_hh.device
    dg = _f32c(gates, "gates").clone()
    with torch.cuda.device(dev):
        nbytes = lib.s2vt_lstm_seq_bwd_bf16_workspace_bytes(T, B, H)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        capi.check(lib.s2vt_lstm_seq_bwd_bf16(T, B, H, _ptr(_f32c(w_hh, "w_hh")), _ptr(dh_out), int(dh_first), _ptr(_f32c(c_all, "c_all")),
                                              _ptr(dg), _ptr(ws), nbytes, int(persistent), int(block), _stream(dev)),
                   "s2vt_lstm_seq_bwd_bf16")
        _check_persist_err(ws)
    return dg


def lstm_seq_bwd_bf16_pair(w0, w1, dh0, dh1, dh_first, c0, c1, gates0, gates1, T, B, H, block=0):
    lib = capi.load()
    dev = w0.device
    dg0, dg1 = _f32c(gates0, "gates0").clone(), _f32c(gates1, "gates1").clone()
    with torch.cuda.device(dev):
        nbytes = 2 * lib.s2vt_lstm_seq_bwd_bf16_workspace_bytes(T, B, H)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        capi.check(lib.s2vt_lstm_seq_bwd_bf16_pair(T, B, H, _ptr(_f32c(w0, "w0")), _ptr(_f32c(w1, "w1")), _ptr(dh0), _ptr(dh1),
                                                   int(dh_first), _ptr(_f32c(c0, "c0")), _ptr(_f32c(c1, "c1")), _ptr(dg0), _ptr(dg1),
                                                   _ptr(ws), nbytes, int(block), _stream(dev)), "s2vt_lstm_seq_bwd_bf16_pair")
        _check_persist_err(ws)
    return dg0, dg1


def lstm_seq_fwd_persist(T, B, gx, n_gx, bias, w_hh, block=0, second=None, poison=True):
    """fp32-equivalent layer forward with the persistent split-precision kernel (three bf16 planes per operand,
    lstm_persist_x3.hip): returns (h_all, c_all, gates).  `second` = (gx, bias, w_hh) of another layer of the same shape that
    shares every launch: then a pair of result tuples is returned."""
    lib = capi.load()
    w_hh = _f32c(w_hh, "w_hh")
    H = w_hh.shape[1]
    dev = w_hh.device
    with torch.cuda.device(dev):
        # (0xFF bytes: bf16 NaN patterns - the kernel must never read a plane element it has not written)
        n = lib.s2vt_lstm_seq_x3_workspace_bytes(T, B, H)
        if n == 0:
            raise capi.S2VTHipError("lstm_seq_fwd_persist: shape B=%d H=%d is not supported on this device" % (B, H))
        ws = torch.full((n,), 0xFF, dtype=torch.uint8, device=dev) if poison else torch.empty(n, dtype=torch.uint8, device=dev)
        sets = []
        for g, b, w in [(gx, bias, w_hh)] + ([second] if second is not None else []):
            stash = torch.empty(T * B, 4 * H, dtype=torch.float32, device=dev)
            if n_gx:
                stash[:n_gx * B].copy_(g)
            sets.append((stash, b, _f32c(w, "w_hh"), torch.empty(T * B, H, dtype=torch.float32, device=dev),
                         torch.empty(T * B, H, dtype=torch.float32, device=dev)))
        a, b2 = sets[0], (sets[1] if len(sets) > 1 else (None,) * 5)
        capi.check(lib.s2vt_lstm_seq_fwd_x3_persist(T, B, H, _ptr(a[0]), _ptr(b2[0]), int(n_gx), _ptr(a[1]), _ptr(b2[1]), _ptr(a[2]),
                                                    _ptr(b2[2]), _ptr(a[3]), _ptr(b2[3]), _ptr(a[4]), _ptr(b2[4]), int(block),
                                                    _ptr(ws), n, _stream(dev)), "s2vt_lstm_seq_fwd_x3_persist")
        _check_persist_err(ws)
    outs = [(x[3], x[4], x[0]) for x in sets]
    return outs[0] if second is None else tuple(outs)


def lstm_seq_bwd_persist(T, B, w_hh, dh_out, dh_first, c_all, gates, block=0, second=None):
    """fp32-equivalent BPTT with the persistent split-precision reduce-scatter kernel (lstm_persist_x3.hip): returns dG (`gates`
    untouched).  `second` = (w_hh, dh_out, c_all, gates)."""
    lib = capi.load()
    w_hh = _f32c(w_hh, "w_hh")
    H = w_hh.shape[1]
    dev = w_hh.device
    with torch.cuda.device(dev):
        n = lib.s2vt_lstm_seq_bwd_x3_workspace_bytes(T, B, H, int(block))
        ws = torch.full((n,), 0xFF, dtype=torch.uint8, device=dev)     # (NaN patterns: nothing may be read before it is written)
        sets = [(_f32c(w, "w_hh"), dh, _f32c(c, "c_all"), _f32c(g, "gates").clone())
                for w, dh, c, g in [(w_hh, dh_out, c_all, gates)] + ([second] if second is not None else [])]
        a, b2 = sets[0], (sets[1] if len(sets) > 1 else (None,) * 4)
        capi.check(lib.s2vt_lstm_seq_bwd_x3_persist(T, B, H, _ptr(a[0]), _ptr(b2[0]), _ptr(a[1]), _ptr(b2[1]), int(dh_first),
                                                    _ptr(a[2]), _ptr(b2[2]), _ptr(a[3]), _ptr(b2[3]), int(block), _ptr(ws), n,
                                                    _stream(dev)), "s2vt_lstm_seq_bwd_x3_persist")
        _check_persist_err(ws)
    return sets[0][3] if second is None else (sets[0][3], sets[1][3])
