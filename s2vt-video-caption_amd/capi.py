"""ctypes binding of include/s2vt_hip.h (libs2vt_hip.so).

This is plumbing only: torch owns every tensor, this module passes raw device pointers, sizes and the
current HIP stream.  There is no CPU fallback: if the library cannot be loaded the import of the
compute path fails loudly.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# S2VT_LIB: another build of the library (A/B timing of two builds on one GPU box: tools/bench_gemm_shapes.py); never set in production
LIB_PATH = os.environ.get("S2VT_LIB") or os.path.join(_HERE, "libs2vt_hip.so")


class Dims(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("B", "L", "F", "H", "E", "V")]


PARAM_FIELDS = ("vid_w_ih", "vid_w_hh", "vid_b_ih", "vid_b_hh", "word_w_ih", "word_w_hh", "word_b_ih", "word_b_hh",
                "feat_w", "feat_b", "out_w", "out_b", "emb_w")
# state_dict keys in the same order
PARAM_KEYS = ("vid_rnn.weight_ih_l0", "vid_rnn.weight_hh_l0", "vid_rnn.bias_ih_l0", "vid_rnn.bias_hh_l0",
              "word_rnn.weight_ih_l0", "word_rnn.weight_hh_l0", "word_rnn.bias_ih_l0", "word_rnn.bias_hh_l0",
              "feat_linear.weight", "feat_linear.bias", "out_linear.weight", "out_linear.bias", "embedding.weight")


class Params(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in PARAM_FIELDS]


class Grads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in PARAM_FIELDS]


# name -> (restype, argtypes): every symbol include/s2vt_hip.h declares
ABI_VERSION = 9          # S2VT_ABI_VERSION of include/s2vt_hip.h this binding was written against

SIGNATURES = {
    "s2vt_abi_version": (c_int32, []),
    "s2vt_last_error": (c_char_p, []),
    "s2vt_padded_batch": (c_int32, [c_int32]),
    "s2vt_train_workspace_bytes": (c_size_t, [POINTER(Dims)]),
    "s2vt_train_forward": (c_int32, [POINTER(Dims), POINTER(Params), c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "s2vt_train_forward_dropout": (c_int32, [POINTER(Dims), POINTER(Params), c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                             c_void_p, c_size_t, c_void_p]),
    "s2vt_train_backward_dropout": (c_int32, [POINTER(Dims), POINTER(Params), c_void_p, c_void_p, c_void_p, POINTER(Grads),
                                              c_void_p, c_void_p, c_size_t, c_void_p]),
    "s2vt_check_async_error": (c_int32, [c_int32]),
    "s2vt_train_backward": (c_int32, [POINTER(Dims), POINTER(Params), c_void_p, c_void_p, POINTER(Grads), c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "s2vt_backward_wait_grads": (c_int32, [c_int32, c_void_p]),
    "s2vt_backward_order": (c_int32, [POINTER(c_int32), POINTER(c_int32)]),
    "s2vt_beam_workspace_bytes": (c_size_t, [POINTER(Dims), c_int32]),
    "s2vt_beam_step": (c_int32, [POINTER(Dims), POINTER(Params), c_int32] + [c_void_p] * 14 + [c_size_t, c_void_p]),
    "s2vt_beam_step_cached": (c_int32, [POINTER(Dims), POINTER(Params), c_int32] + [c_void_p] * 14 + [c_size_t, c_void_p, c_size_t,
                                                                                                     c_void_p]),
    "s2vt_decode_workspace_bytes": (c_size_t, [POINTER(Dims)]),
    "s2vt_greedy_decode": (c_int32, [POINTER(Dims), POINTER(Params), c_void_p, c_int32, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "s2vt_decode_cache_bytes": (c_size_t, [POINTER(Dims)]),
    "s2vt_decode_uses_cache": (c_int32, [POINTER(Dims)]),
    "s2vt_greedy_decode_cached": (c_int32, [POINTER(Dims), POINTER(Params), c_void_p, c_int32, c_void_p, c_void_p, c_size_t,
                                            c_void_p, c_size_t, c_int32, c_void_p]),
    "s2vt_adam_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double, c_double, c_double, c_int64, c_void_p]),
    "s2vt_mean_ce_forward": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "s2vt_mean_ce_backward": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "s2vt_mask_criterion_forward": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    "s2vt_mask_criterion_backward": (c_int32, [c_int32, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "s2vt_gemm_f32": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64,
                                c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    "s2vt_gemm_f32_splitk": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64,
                                       c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "s2vt_split_planes": (c_int32, [c_int32, c_int32, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_int32,
                                    c_int32, c_void_p]),
    "s2vt_gemm_bf16_nt": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                    c_int64, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "s2vt_gemm_bf16_tt": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                    c_int64, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "s2vt_gemm_tune": (c_int32, [c_int32, c_int32, c_int32]),
    "s2vt_beam_queue_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "s2vt_beam_queue_step": (c_int32, [c_int32] * 6 + [c_void_p, c_size_t] + [c_void_p] * 6),
    "s2vt_beam_queue_result": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_int32, c_void_p, c_void_p]),
    "s2vt_feat_proj_fwd": (c_int32, [POINTER(Dims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "s2vt_feat_proj_bwd": (c_int32, [POINTER(Dims), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "s2vt_colsum_ws_floats": (c_size_t, [c_int64, c_int32]),
    "s2vt_lstm_step_fwd": (c_int32, [c_int32, c_int32] + [c_void_p] * 9),
    "s2vt_lstm_step_fwd_token": (c_int32, [c_int32, c_int32, c_int32, c_int32] + [c_void_p] * 6 + [c_int64, c_void_p, c_void_p,
                                                                                            c_int32, c_void_p, c_void_p, c_void_p]),
    "s2vt_lstm_step_bwd": (c_int32, [c_int32, c_int32] + [c_void_p] * 7 + [c_int32, c_void_p, c_void_p]),
    "s2vt_lstm_seq_fwd": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_int32] + [c_void_p] * 6),
    "s2vt_lstm_seq_bwd": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32] + [c_void_p] * 5),
    "s2vt_lstm_seq_bf16_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "s2vt_lstm_seq_fwd_bf16": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_void_p]),
    "s2vt_lstm_seq_fwd_bf16_pair": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32] + [c_void_p] * 9 +
                                    [c_size_t, c_int32, c_void_p]),
    "s2vt_lstm_seq_bwd_bf16_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "s2vt_lstm_seq_bwd_bf16": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                         c_size_t, c_int32, c_int32, c_void_p]),
    "s2vt_lstm_seq_bwd_bf16_pair": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p]),
    "s2vt_lstm_seq_x3_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "s2vt_lstm_seq_fwd_x3_persist": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32] + [c_void_p] * 8 +
                                  [c_int32, c_void_p, c_size_t, c_void_p]),
    "s2vt_lstm_seq_bwd_x3_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "s2vt_lstm_seq_bwd_x3_persist": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32] +
                                     [c_void_p] * 4 + [c_int32, c_void_p, c_size_t, c_void_p]),
    "s2vt_set_recurrence_mode": (c_int32, [c_int32]),
    "s2vt_recurrence_plan": (c_int32, [c_int32, c_int32, POINTER(c_int32), POINTER(c_int32)]),
    "s2vt_decode_step_argmax": (c_int32, [c_int32, c_int32, c_int32] + [c_void_p] * 5),
    "s2vt_decode_step_argmax_x3_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "s2vt_decode_step_argmax_x3": (c_int32, [c_int32, c_int32, c_int32] + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "s2vt_mean_ce_backward_fused": (c_int32, [POINTER(Dims), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t,
                                              c_void_p]),
    "s2vt_set_option": (c_int32, [c_char_p, c_int32]),
    "s2vt_option_count": (c_int32, []),
    "s2vt_option_name": (c_char_p, [c_int32]),
    "s2vt_set_gemm_mode": (c_int32, [c_int32]),
    "s2vt_set_pipeline_block": (c_int32, [c_int32]),
    "s2vt_set_decode_schedule": (c_int32, [c_int32]),
    "s2vt_decode_encode_cached": (c_int32, [POINTER(Dims), POINTER(Params), c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_int32,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "s2vt_beam_step_gx": (c_int32, [POINTER(Dims), POINTER(Params), c_int32] + [c_void_p] * 10 + [c_void_p, c_size_t, c_void_p, c_size_t,
                                                                                                  c_void_p]),
    "s2vt_pipeline_overlaps": (c_int32, []),
    "s2vt_test_occupy_cus": (c_int32, [c_int32, c_int32, c_int64, c_void_p]),
    "s2vt_set_graph_mode": (c_int32, [c_int32]),
    "s2vt_graph_stats": (c_int32, [POINTER(c_int64), POINTER(c_int64)]),
    "s2vt_prof_enable": (c_int32, [c_int32]),
    "s2vt_prof_read": (c_int32, [c_int32, POINTER(c_double), POINTER(c_int64)]),
    "s2vt_prof_read_busy": (c_int32, [c_int32, POINTER(c_double)]),
    "s2vt_prof_reset": (c_int32, []),
}

_lib = None


class S2VTHipError(RuntimeError):
    pass


def load():
    """dlopen libs2vt_hip.so and type every entry point.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise S2VTHipError(
            "libs2vt_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python s2vt-video-caption_amd/build.py`. There is no CPU fallback for the S2VT hot path." % LIB_PATH)
    # The library must share torch's HIP runtime (torch owns the device memory and the streams it is handed):
    # torch bundles its own libamdhip64.so.7, and the dynamic loader binds our DT_NEEDED of the same SONAME to
    # whichever copy is already mapped.  Loading torch (and its runtime) FIRST guarantees a single runtime; the
    # other order gives two runtimes in one process ("no ROCm-capable device is detected").
    import torch
    rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.s2vt_abi_version() != ABI_VERSION:
        raise S2VTHipError("libs2vt_hip.so ABI %d != %d" % (lib.s2vt_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


ERR_INDEX, ERR_TIMEOUT = -2, -3      # S2VT_ERR_INDEX / S2VT_ERR_TIMEOUT of include/s2vt_hip.h


def check(rc, what):
    if rc != 0:
        msg = load().s2vt_last_error()
        text = "%s failed (rc=%d): %s" % (what, rc, msg.decode(errors="replace") if msg else "?")
        if rc == ERR_INDEX:
            raise IndexError(text)           # what nn.Embedding raises in the reference (S2VTModel.py:71)
        raise S2VTHipError(text)


def check_async_error(wait=True):
    """Raise the device-side error (target id out of range -> IndexError, hand-off time-out) of the last
    s2vt_train_forward, if any.  Call after a stream synchronisation (loss.item()) to get it without delay."""
    check(load().s2vt_check_async_error(1 if wait else 0), "s2vt_check_async_error")


def prof_read_busy(kind):
    """wall-clock ms during which at least one bracket of `kind` was open (overlapping lanes counted once)"""
    ms = c_double(0.0)
    check(load().s2vt_prof_read_busy(kind, ctypes.byref(ms)), "s2vt_prof_read_busy")
    return ms.value


def prof_read(kind):
    ms, n = c_double(0.0), c_int64(0)
    check(load().s2vt_prof_read(kind, ctypes.byref(ms), ctypes.byref(n)), "s2vt_prof_read")
    return ms.value, n.value


def recurrence_plan(B, H):
    """(forward, backward) recurrence kernels the train drivers would run for a batch of B rows (as the caller hands it over: a ragged
    batch the library pads - s2vt_padded_batch - is planned at its padded size) and hidden size H under the current options:
    0 launch per timestep, 1 persistent bf16, 3 persistent split precision."""
    import ctypes
    f, b = c_int32(0), c_int32(0)
    lib = load()
    check(lib.s2vt_recurrence_plan(int(lib.s2vt_padded_batch(int(B))), int(H), ctypes.byref(f), ctypes.byref(b)), "s2vt_recurrence_plan")
    return f.value, b.value
