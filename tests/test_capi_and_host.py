"""CPU: the C-ABI library loads and exports every symbol include/s2vt_hip.h declares; host-side logic of
the drop-in modules (no compute calls without a GPU)."""
import numpy as np
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "s2vt_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(s2vt_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound(lib):
    from s2vt_video_caption_amd import capi
    declared = _declared_symbols()
    assert len(declared) >= 20
    raw = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "libs2vt_hip.so does not export %s" % name
        assert name in capi.SIGNATURES, "capi.py does not bind %s" % name
    assert sorted(capi.SIGNATURES) == declared
    assert lib.s2vt_abi_version() == capi.ABI_VERSION == 9


def test_workspace_queries_and_argument_errors(lib):
    from s2vt_video_caption_amd import capi
    d = capi.Dims(64, 80, 4096, 1000, 1000, 12000)
    tb = lib.s2vt_train_workspace_bytes(ctypes.byref(d))
    db = lib.s2vt_decode_workspace_bytes(ctypes.byref(d))
    assert 5e8 < tb < 6e9 and 1e8 < db < tb
    bad = capi.Dims(0, 80, 4096, 1000, 1000, 12000)
    assert lib.s2vt_train_workspace_bytes(ctypes.byref(bad)) == 0
    # null pointers are rejected before any GPU call, with a message
    rc = lib.s2vt_train_forward(ctypes.byref(d), None, None, None, 0, None, None, 0, None)
    assert rc == -1 and b"s2vt_train_forward" in lib.s2vt_last_error()
    with pytest.raises(capi.S2VTHipError):
        capi.check(rc, "s2vt_train_forward")


def test_option_table_set_query_clamp_and_the_padding_rule(lib):
    """s2vt_set_option / s2vt_option_name on the host alone (no device call behind them): a negative value queries, values are
    clamped to the option's range, an unknown name is refused; s2vt_padded_batch follows pad_min_batch and the arithmetic mode."""
    names = [lib.s2vt_option_name(i).decode() for i in range(lib.s2vt_option_count())]
    assert {"gemm_mode", "persist", "pipe_block", "corun", "bptt_solo", "pad_min_batch", "gemv", "cu_reserve"} <= set(names)
    assert lib.s2vt_set_option(b"nope", 1) == -(2 ** 31) and b"nope" in lib.s2vt_last_error()
    prev = {n: lib.s2vt_set_option(n.encode(), -1) for n in names}
    try:
        assert lib.s2vt_set_option(b"corun", 99) == prev["corun"] and lib.s2vt_set_option(b"corun", -1) == 5        # clamped to 0..5
        assert lib.s2vt_set_option(b"gemm_mode", 2) == prev["gemm_mode"] and lib.s2vt_set_option(b"gemm_mode", -1) == 0   # 0 | 1 | 3 only
        assert lib.s2vt_set_gemm_mode(3) == 0 and lib.s2vt_set_option(b"gemm_mode", -1) == 3       # the typed setters are views of the table
        lib.s2vt_set_option(b"pad_min_batch", 33)
        assert [lib.s2vt_padded_batch(b) for b in (1, 16, 32, 33, 64, 65, 100, 128)] == [1, 16, 32, 64, 64, 128, 128, 128]
        lib.s2vt_set_option(b"pad_min_batch", 1)
        assert [lib.s2vt_padded_batch(b) for b in (1, 16, 64)] == [64, 64, 64]
        lib.s2vt_set_gemm_mode(0)                                  # fp32-input MFMA: no plane path, nothing is padded
        assert lib.s2vt_padded_batch(100) == 100
    finally:
        for n, v in prev.items():
            lib.s2vt_set_option(n.encode(), v)


def test_dropin_module_layout_and_reference_pickle():
    import S2VTModel
    from s2vt_video_caption_amd import capi, synth
    d = synth.CONFIGS["tiny"]
    m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
    assert tuple(m.state_dict().keys()) == capi.PARAM_KEYS
    shapes = synth.param_shapes(d["V"], d["F"], d["H"], d["E"])
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == shapes[k]
    for attr in ("feat_dim", "length", "dim_hid", "dim_embed", "sos_ix", "eos_ix", "vocab_size", "rnn_type"):
        assert hasattr(m, attr)
    # a full-module pickle WRITTEN BY THE REFERENCE class loads into the drop-in class
    ref = torch.load(os.path.join(ROOT, "tests", "golden", "tiny_reference_module.pth"), weights_only=False)
    assert isinstance(ref, S2VTModel.S2VT)
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=7)
    for k, v in ref.state_dict().items():
        assert torch.equal(v, sd[k])
    m.load_state_dict(ref.state_dict())
    # same default-init RNG stream as the reference's construction order (S2VTModel.py:19-28)
    torch.manual_seed(0)
    a = S2VTModel.S2VT(30, 16, 4, dim_hid=8, dim_embed=8)
    torch.manual_seed(0)
    vid = torch.nn.LSTM(8, 8, batch_first=True)
    assert torch.equal(a.vid_rnn.weight_ih_l0, vid.weight_ih_l0)


def test_cpu_tensors_fail_loudly_no_fallback():
    import S2VTModel
    import utils
    from s2vt_video_caption_amd import capi
    m = S2VTModel.S2VT(50, 64, 8, dim_hid=32, dim_embed=24)
    with pytest.raises(capi.S2VTHipError):
        m(torch.randn(2, 8, 64), targets=torch.zeros(2, 7, dtype=torch.long), mode="train")
    with pytest.raises(capi.S2VTHipError):
        m(torch.randn(2, 8, 64), mode="test")
    with pytest.raises(capi.S2VTHipError):
        utils.MaskCriterion()(torch.randn(2, 7, 50), torch.zeros(2, 8, dtype=torch.long), torch.ones(2, 8))


def test_unsupported_configs_raise():
    import S2VTModel
    m = S2VTModel.S2VT(50, 64, 8, dim_hid=32, dim_embed=24, rnn_type="gru")
    with pytest.raises(NotImplementedError):
        m._hip_params()
    m = S2VTModel.S2VT(50, 64, 8, dim_hid=32, dim_embed=24, num_layers=2)
    with pytest.raises(NotImplementedError):
        m._hip_params()


def test_synth_recipe_is_deterministic_and_order_independent():
    from s2vt_video_caption_amd import synth
    a = synth.make_state_dict(50, 64, 32, 24, seed=3)
    b = synth.make_state_dict(50, 64, 32, 24, seed=3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    f1, c1, m1 = synth.make_batch(4, 8, 64, 50, seed=1)
    f2, c2, m2 = synth.make_batch(4, 8, 64, 50, seed=1)
    assert torch.equal(f1, f2) and torch.equal(c1, c2) and torch.equal(m1, m2)
    assert (c1[:, 0] == 3).all() and ((c1 == 4).sum(1) == 1).all() and c1.max() < 50
    assert torch.equal(m1.sum(1), (c1 != 0).sum(1).float())


def test_early_stopping_protocol(tmp_path):
    """utils.EarlyStopping: the reference's protocol (utils.py:29-80) — patience counter, full-module checkpoint on
    every improvement, val_loss_min bookkeeping."""
    import utils
    path = str(tmp_path / "stop.pth")
    m = torch.nn.Linear(2, 2)
    msgs = []
    es = utils.EarlyStopping(patience=2, verbose=True, path=path, trace_func=msgs.append)
    for v in (1.0, 0.9, 0.95):
        es(v, m)
    assert not es.early_stop and es.counter == 1 and abs(es.val_loss_min - 0.9) < 1e-12
    es(0.96, m)
    assert es.early_stop and es.counter == 2
    assert isinstance(torch.load(path, weights_only=False), torch.nn.Linear)
    assert any("Validation loss decreased" in s for s in msgs) and any("EarlyStopping counter: 2 out of 2" in s for s in msgs)


def test_ids_to_caption_rules():
    """eval.py:54-58 / :90-96: ids -> words, cut at the first <eos>; beam outputs drop their leading <sos>."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("s2vt_eval", os.path.join(ROOT, "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    ix2word = {"0": "<pad>", "3": "<sos>", "4": "<eos>", "5": "a", "6": "cat"}
    assert ev.ids_to_caption([5, 6, 4, 5, 5], ix2word) == "a cat"
    assert ev.ids_to_caption([5, 6, 5], ix2word) == "a cat a"
    assert ev.ids_to_caption([3, 5, 6, 4, 6], ix2word, drop_sos=True) == "a cat"


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("V", [25, 3000])
def test_beam_queues_match_reference_heap(ties, V):
    """The vectorised per-depth candidate sets (beam.BeamQueues) pop, re-insert, stop and back-trace exactly like
    the literal heap bookkeeping of S2VTModel.py:186-236 (beam.HeapQueues) - also when scores tie, where the pop
    order depends on the heap's internal layout."""
    from s2vt_video_caption_amd import beam

    def run(Q, seed, B=9, bw=5, eos=2):
        rng = np.random.default_rng(seed)
        q = Q(B, bw, 1, eos)
        d = 0
        while d < 12 and not q.all_done():
            d += 1
            rb, rs, rt = q.pop()
            ix = lp = None
            if len(rb):
                ix = np.sort(np.stack([rng.choice(V, 20, replace=False) for _ in rb]), axis=1).astype(np.int64)
                lp = (-rng.integers(1, 6, size=(len(rb), 20)).astype(np.float32) if ties
                      else -rng.random((len(rb), 20), dtype=np.float32) * 10)
            q.push(ix, lp)
        return q.finish(), list(rs), getattr(q, "tie_fallbacks", 0)

    for seed in range(6):
        a, rows_a, _ = run(beam.HeapQueues, seed)
        b, rows_b, nf = run(beam.BeamQueues, seed)
        assert a == b and rows_a == rows_b
        assert (nf > 0) == ties


def test_graft_entry_build_runs():
    """the driver's build check: compile (or find up to date) every HIP source, load the library, import the drop-ins"""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()


def test_dp_reducer_attach_groups_cover_the_flat_buffer():
    """FlatGradAllReducer.attach(): the three gradient groups (out_linear | word_rnn + embedding | vid_rnn + feat_linear)
    are disjoint ranges covering the whole flat buffer, and the backward's gradient sink is the 13 flat views in the
    library's parameter order (host logic only: no GPU call)."""
    import S2VTModel
    from s2vt_video_caption_amd import dp, functional, capi
    m = S2VTModel.S2VT(50, 64, 8, dim_hid=32, dim_embed=24)
    red = dp.FlatGradAllReducer(m.parameters()).attach(m)
    n = sum(p.numel() for p in m.parameters())
    spans = sorted(r for g in red.groups.values() for r in g)
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = {g: sum(hi - lo for lo, hi in r) for g, r in red.groups.items()}
    named = dict(m.named_parameters())
    assert sizes[0] == sum(p.numel() for k, p in named.items() if k.startswith("out_linear"))
    assert sizes[1] == sum(p.numel() for k, p in named.items() if k.startswith(("word_rnn", "embedding")))
    sink = functional.grad_sink_for(m)
    assert len(sink) == len(capi.PARAM_KEYS)
    for k, g in zip(capi.PARAM_KEYS, sink):
        assert g.data_ptr() == named[k].grad.data_ptr() and g.shape == named[k].shape
    # the sink lives outside the module: full-module pickles keep the reference's layout
    assert not any("sink" in k for k in m.__dict__)


def test_load_glove_weights_fills_known_words_and_keeps_the_embedding_trainable(tmp_path, monkeypatch):
    """S2VTModel.py:112-147: vocabulary words found in the GloVe file get its vectors, the others a Xavier-normal row; the
    table is cached as ./data/word2embed.json and can be loaded back from that cache."""
    import S2VTModel
    monkeypatch.chdir(tmp_path)
    ix2word = {"0": "<pad>", "1": "a", "2": "man", "3": "<sos>", "4": "<eos>", "5": "guitar"}
    with open(tmp_path / "glove.txt", "w", encoding="utf-8") as f:
        f.write("the 9 9 9 9\n")                       # not in the vocabulary: skipped
        f.write("man 0.5 -1.25 2 0.125\n")
        f.write("guitar 1 2 3 4\n")
    m = S2VTModel.S2VT(vocab_size=6, feat_dim=8, length=4, dim_hid=8, dim_embed=4)
    m.load_glove_weights(str(tmp_path / "glove.txt"), 4, ix2word, word2embed=None)
    w = m.embedding.weight
    assert w.requires_grad and tuple(w.shape) == (6, 4)
    assert torch.equal(w[2].detach(), torch.tensor([0.5, -1.25, 2.0, 0.125]))
    assert torch.equal(w[5].detach(), torch.tensor([1.0, 2.0, 3.0, 4.0]))
    assert float(w[1].detach().abs().sum()) > 0          # no vector for "a": random row, not zeros
    assert sorted(m.state_dict()) == sorted(S2VTModel.S2VT(6, 8, 4, dim_hid=8, dim_embed=4).state_dict())
    m2 = S2VTModel.S2VT(vocab_size=6, feat_dim=8, length=4, dim_hid=8, dim_embed=4)
    m2.load_glove_weights("unused", 4, ix2word, word2embed=str(tmp_path / "data" / "word2embed.json"))
    assert torch.equal(m2.embedding.weight[5].detach(), w[5].detach())
    with pytest.raises(AssertionError):
        m2.load_glove_weights("unused", 3, ix2word)


def test_att_baseline_dropin_layout_matches_reference_fixture():
    """attention_baseline.Att_Baseline: parameter names, order and shapes are the reference's (recorded
    in tests/golden/att_tiny.npz by oracle/make_golden.py from the reference class), and the seeded default initialisation
    reproduces the reference's numbers - its state_dicts and pickles load unchanged.  No compute here (CPU)."""
    import os
    import numpy as np
    import torch
    import attention_baseline
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "att_tiny.npz"))
    B, L, F, H, E, V = (int(x) for x in g["dims"])
    torch.manual_seed(int(g["seed"]))
    m = attention_baseline.Att_Baseline(V, F, L, dim_hid=H, dim_embed=E)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g["param/" + k]), k
    import pytest
    from s2vt_video_caption_amd import capi
    with pytest.raises(capi.S2VTHipError):        # no CPU fallback: the product path fails loudly off the GPU
        m(torch.zeros(B, L, F), targets=torch.zeros(B, L - 1, dtype=torch.long), mode="train")
