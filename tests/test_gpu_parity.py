"""GPU parity of the whole hot path, called through the drop-in API (S2VTModel.S2VT / utils.MaskCriterion ->
ctypes -> C ABI -> HIP kernels), against (1) the golden vectors produced by the reference itself and (2) the
oracle on the same seeded inputs.  Tolerances (fp32): logits 2e-5 abs, loss 1e-4 (north_star), greedy ids
bit-exact."""
import numpy as np
import pytest
import torch

from oracle import s2vt_oracle as orc
from s2vt_video_caption_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(d, sd):
    import S2VTModel
    m = S2VTModel.S2VT(d["V"], d["F"], d["L"], dim_hid=d["H"], dim_embed=d["E"])
    m.load_state_dict(sd)
    return m.to(DEV)


import contextlib


@contextlib.contextmanager
def _options(lib, **kv):
    """library options set for the block and put back afterwards: a test that asserts a plan or a path SETS the options that plan
    rests on - the suite must not depend on the S2VT_* environment it is run under"""
    prev = {k: lib.s2vt_set_option(k.encode(), v) for k, v in kv.items()}
    assert all(v != -(2 ** 31) for v in prev.values()), "unknown option"
    try:
        yield
    finally:
        for k, v in prev.items():
            lib.s2vt_set_option(k.encode(), v)


_PLAN_DEFAULTS = dict(persist=1, persist_x3_fwd=1, persist_x3_bwd=2, pipe_block=32)


def _setup(g, name):
    d = synth.CONFIGS[name]
    seed = int(g["seed"])
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=seed, out_scale=float(g["out_scale"]))
    feats, caps, mask = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=1234 + seed)
    return d, sd, feats, caps, mask


def _train(m, feats, caps, mask, n_steps):
    import utils
    crit = utils.MaskCriterion()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    losses, grads, logits0 = [], None, None
    f, c, k = feats.to(DEV), caps.to(DEV), mask.to(DEV)
    for s in range(n_steps):
        opt.zero_grad()
        m.train()
        probs = m(f, targets=c[:, :-1], mode="train")
        loss = crit(probs, c, k)
        loss.backward()
        if s == 0:
            logits0 = probs.detach().cpu()
            grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        opt.step()
        losses.append(float(loss))
    return losses, grads, logits0


def test_native_library_is_the_one_running(lib):
    from s2vt_video_caption_amd import capi
    maps = open("/proc/self/maps").read()
    assert capi.LIB_PATH in maps, "libs2vt_hip.so is not mapped into the test process"


def test_tiny_against_reference_golden(lib, golden):
    g = golden("tiny")
    d, sd, feats, caps, mask = _setup(g, "tiny")
    m = _model(d, sd)
    losses, grads, logits = _train(m, feats, caps, mask, int(g["n_steps"]))
    assert np.abs(logits.numpy() - g["logits"]).max() < 2e-5
    assert np.abs(np.array(losses) - g["losses"]).max() < 1e-4
    for k in orc.KEYS:
        ref = g["grad/" + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 1e-6 + 1e-4 * np.abs(ref).max(), k
    for k, v in m.state_dict().items():
        assert np.abs(v.cpu().numpy() - g["final/" + k]).max() < 2e-5, k
    m.eval()
    with torch.no_grad():
        ids = m(feats.to(DEV), mode="test")
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (d["B"], d["L"] - 1)
    np.testing.assert_array_equal(ids.cpu().numpy(), g["greedy_ids"])


def test_tiny_beam_search_against_reference_golden(lib, golden):
    g = golden("tiny")
    d, sd, feats, caps, mask = _setup(g, "tiny")
    m = _model(d, sd).eval()
    with torch.no_grad():
        out = m(feats.to(DEV), mode="beam_search", beam_width=int(g["beam_width"]), max_beam_depth=30)
    assert tuple(out[0][0].shape) == (1, 1)                    # leading [[<sos>]] tensor as in the reference
    for b, s in enumerate(out):
        assert [int(t.item()) for t in s] == [int(x) for x in g["beam_ids"][b] if x >= 0]


def test_c1_against_reference_golden(lib, golden):
    g = golden("c1")
    d, sd, feats, caps, mask = _setup(g, "c1")
    m = _model(d, sd)
    m.eval()
    with torch.no_grad():
        ids = m(feats.to(DEV), mode="test").cpu().numpy()
    np.testing.assert_array_equal(ids, g["greedy_ids"])          # min top-2 margin of this fixture: 2.2e-4
    m.train()
    losses, grads, logits = _train(m, feats, caps, mask, int(g["n_steps"]))
    assert np.abs(np.array(losses) - g["losses"]).max() < 1e-4
    assert np.abs(logits[:, ::13, :64].numpy() - g["logits_rows"]).max() < 2e-5
    for k in orc.KEYS:
        gn = float(g["gradnorm/" + k])
        assert abs(float(grads[k].double().norm()) - gn) <= 2e-4 * gn + 1e-7, k
        ref = g["gradhead/" + k]
        assert np.abs(grads[k].reshape(-1)[:32].numpy() - ref).max() <= 1e-6 + 2e-4 * np.abs(ref).max(), k


def test_c1_beam_search_against_reference_golden(lib, golden):
    g = golden("c1")
    d, sd, feats, caps, mask = _setup(g, "c1")
    m = _model(d, sd).eval()
    with torch.no_grad():
        out = m(feats[:2].to(DEV), mode="beam_search", beam_width=int(g["beam_width"]), max_beam_depth=30)
    for b, s in enumerate(out):
        assert [int(t.item()) for t in s] == [int(x) for x in g["beam_ids"][b] if x >= 0]


@pytest.mark.parametrize("gemm_mode", [3, 0])
def test_c2_full_size_against_reference_golden(lib, golden, gemm_mode):
    """BASELINE config 2 (B=64, H=E=1000, V=12000, fp32) in both arithmetic modes of the batched GEMMs (3 = split-precision
    bf16x3 on the bf16 matrix cores, the default at B % 64 == 0; 0 = fp32-input MFMA): ALL 64 x 79 greedy token ids equal
    the reference's (the fixture's weakest top-2 margin is 1.3e-3: screened seed, oracle/make_golden.py), loss within 1e-4
    over two Adam steps, logits slice, and every gradient by norm, by sum and element-wise on its first 32 entries."""
    g = golden("c2")
    assert float(g["greedy_margin"].min()) >= 1e-3          # the fixture itself leaves no near-tie
    d, sd, feats, caps, mask = _setup(g, "c2")
    prev = lib.s2vt_set_gemm_mode(gemm_mode)
    try:
        _c2_body(g, d, sd, feats, caps, mask)
    finally:
        lib.s2vt_set_gemm_mode(prev)


@pytest.mark.parametrize("option,value,plan", [("persist", 0, (0, 0)), ("persist_x3_fwd", 0, (0, None)), ("persist_x3_bwd", 0, (3, 0)),
                                               ("persist_x3_bwd", 1, (3, 3)), ("persist_x3_bwd", 2, (3, 3)), ("pipe_block", 0, None), ("pipe_block", 20, None),
                                               ("graph", 1, None), ("cu_reserve", 24, None), ("decode_fused", 0, None), ("corun", 0, None), ("bptt_solo", 0, None)])
def test_c2_train_and_decode_under_every_option(lib, golden, option, value, plan):
    """The config-2 fixture (greedy ids bit-exact, two Adam steps within 1e-4, every gradient) with ONE run-time option of the
    library moved off / onto its default through s2vt_set_option - the whole switch table of csrc/options.hip except the
    arithmetic mode (test_c2_full_size_against_reference_golden runs gemm modes 3 and 0, the c3 tests mode 1) and bptt_units (bf16
    kernels: test_gpu_kernels.py): launches per timestep everywhere, the split-precision persistent kernels per direction, no layer
    pipeline / another block length, hipGraph replay, grids planned for fewer compute units, the two-chain decode schedule."""
    from s2vt_video_caption_amd import capi
    g = golden("c2")
    d, sd, feats, caps, mask = _setup(g, "c2")
    with _options(lib, **{**_PLAN_DEFAULTS, option: value}):          # (the option under test on top of the plan's defaults)
        assert lib.s2vt_set_option(option.encode(), -1) == value
        if plan is not None:
            got = capi.recurrence_plan(d["B"], d["H"])
            assert got[0] == plan[0] and (plan[1] is None or got[1] == plan[1]), (got, plan)
        _c2_body(g, d, sd, feats, caps, mask)
        capi.check_async_error()


def test_every_option_of_the_library_is_named_in_the_header_and_covered(lib):
    """The option table (s2vt_option_count / s2vt_option_name) against include/s2vt_hip.h's list and this file's parametrisations:
    a switch nobody documents or tests cannot be added."""
    import os
    import re
    names = [lib.s2vt_option_name(i).decode() for i in range(lib.s2vt_option_count())]
    assert len(set(names)) == len(names) and lib.s2vt_option_name(len(names)) is None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "s2vt_hip.h")).read()
    tests = "".join(open(os.path.join(root, "tests", f)).read() for f in os.listdir(os.path.join(root, "tests")) if f.endswith(".py"))
    for n in names:
        assert re.search(r"^ \*   %s\s" % re.escape(n), header, re.M), "option %s is not documented in include/s2vt_hip.h" % n
        assert ('"%s"' % n) in tests or ("%s=" % n) in tests, "option %s has no test" % n
    assert lib.s2vt_set_option(b"no_such_option", 1) == -(2 ** 31)


def test_c4_shard_full_size_against_reference_golden(lib, golden):
    """BASELINE configs[3] per-GPU shard: B=128 at full dims (the batch one rank of the 8-way data-parallel run trains on),
    two fp32 train steps of the reference (tests/golden/c4.npz): loss within 1e-4, logits slice, every gradient by norm,
    sum and leading entries.  (Greedy / beam ids at B=128: test_c5_full_batch_greedy_and_beam_against_reference_golden;
    the all-reduce above the shard: tests/test_gpu_dp_two_ranks.py.)"""
    g = golden("c4")
    d, sd, feats, caps, mask = _setup(g, "c4")
    _c2_body(g, d, sd, feats, caps, mask, greedy=False)


def _c2_body(g, d, sd, feats, caps, mask, greedy=True):
    m = _model(d, sd)
    if greedy:
        m.eval()
        with torch.no_grad():
            ids = m(feats.to(DEV), mode="test").cpu().numpy()
        np.testing.assert_array_equal(ids, g["greedy_ids"])          # bit-exact, every row, every step
    m.train()
    losses, grads, logits = _train(m, feats, caps, mask, int(g["n_steps"]))
    assert np.abs(np.array(losses) - g["losses"]).max() < 1e-4, (losses, g["losses"])
    scale = float(g["out_scale"])
    assert np.abs(logits[:, ::13, :64].numpy() - g["logits_rows"]).max() < 5e-5 * scale
    for k in orc.KEYS:
        gn = float(g["gradnorm/" + k])
        assert abs(float(grads[k].double().norm()) - gn) <= 5e-4 * gn + 1e-7, k
        # direction, not only size: the sum of all entries (bound relative to the norm: sums cancel) and the first 32 entries
        n = grads[k].numel()
        assert abs(float(grads[k].double().sum()) - float(g["gradsum/" + k])) <= 2e-4 * gn * n ** 0.5 + 1e-7, k
        ref = g["gradhead/" + k]
        assert np.abs(grads[k].reshape(-1)[:32].numpy() - ref).max() <= 2e-6 + 5e-4 * np.abs(ref).max(), k


@pytest.mark.parametrize("name,cfg,gemm_mode,bound", [("c1long", "c1", 3, 1e-4), ("mid64long", "mid64", 3, 1e-4),
                                                      ("mid64long", "mid64", 0, 1e-4), ("mid64long", "mid64", 1, 5e-3),
                                                      ("c3long", "c3", 1, 5e-3), ("c2long", "c2", 3, 1e-4)])
def test_long_loss_trajectory_against_reference(lib, golden, name, cfg, gemm_mode, bound):
    """FORTY Adam steps on one fixed batch at ten times train.py's learning rate (the reference's loss falls from 4.6 to
    0.007 at c1 dims, from 7.0 to 2.1 at B=64): every step runs on weights that carry the rounding history of all earlier
    steps, Adam's division included.  north_star's bound - training loss within 1e-4 - must hold at EVERY step for the
    fp32 paths (measured 5e-7: B=4 takes the fp32-MFMA driver, B=64 the split-precision two-stream driver, mode 0 its
    fp32-MFMA twin); the bf16 configuration (mode 1, persistent recurrence kernels) stays within 5e-3 of the same fp32
    trajectory (measured 1.2e-3).  Final parameter norms within 1e-4 (fp32) / 2e-3 (bf16; measured 4e-4) relative.
    c3long: BASELINE configs[2] AT ITS OWN SIZE (B=256, H=E=1000, V=12000): TEN fp32 Adam steps of the reference (lr 1e-3,
    one batch; train.py:116-127) against the bf16 configuration - persistent recurrence, bf16 batched GEMMs - step by step:
    |loss - reference| < 5e-3 at every step (measured 1.5e-3, round 4; the reference's loss falls from 9.53 to 1.46), final norms
    within 0.2 % (measured 0.04 %).
    c2long: BASELINE configs[1] - the headline workload - AT ITS OWN SIZE (B=64, H=E=1000, V=12000): ten fp32 Adam steps of the
    reference against the timed configuration (split-precision GEMMs, split-precision persistent forward recurrence, two-lane
    BPTT): north_star's 1e-4 on the loss at every step, final parameter norms within 1e-4."""
    import utils
    g = golden(name)
    d = synth.CONFIGS[cfg]
    seed, n, lr = int(g["seed"]), int(g["n_steps"]), float(g["lr"])
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=seed)
    feats, caps, mask = (t.to(DEV) for t in synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=1234 + seed))
    prev = lib.s2vt_set_gemm_mode(gemm_mode)
    try:
        m = _model(d, sd)
        m.train()
        crit = utils.MaskCriterion()
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        losses = []
        for _ in range(n):
            opt.zero_grad()
            loss = crit(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        from s2vt_video_caption_amd import capi
        capi.check_async_error()
    finally:
        lib.s2vt_set_gemm_mode(prev)
    diff = np.abs(np.array(losses) - g["losses"])
    assert diff.max() < bound, (int(diff.argmax()), float(diff.max()))
    rel = 1e-4 if gemm_mode != 1 else 2e-3
    for k, v in m.state_dict().items():
        ref = float(g["finalnorm/" + k])
        assert abs(float(v.double().norm()) - ref) <= rel * ref + 1e-9, k


def test_against_oracle_on_fresh_seeds(lib):
    """Ragged / edge shapes the fixtures do not cover: B=1, B not a multiple of the tile, H not a multiple of
    8 or 4-aligned E, V not 4-aligned; the B=64 / B=128 cases take the split-precision, two-lane drivers (blocked plane
    layout with partial 64-row blocks, k padding, clamped 256x256 tiles).  The last case has a 12-word vocabulary under
    2816 caption rows: every token is "heavy" for the embedding gradient (more than 64 rows each: ten tokens go through the
    heavy-token kernel, where the BASELINE batches have two or three)."""
    for (B, L, Fd, H, E, V, seed, words) in [(1, 4, 20, 12, 8, 23, 1, (1, 2)), (19, 6, 33, 36, 20, 57, 2, (1, 2)),
                                             (33, 3, 64, 40, 44, 30, 3, (1, 2)), (64, 5, 70, 44, 28, 61, 4, (1, 2)),
                                             (128, 4, 36, 100, 52, 333, 5, (1, 2)), (256, 12, 24, 16, 8, 12, 6, (5, 9))]:
        sd = synth.make_state_dict(V, Fd, H, E, seed=seed)
        feats, caps, mask = synth.make_batch(B, L, Fd, V, seed=seed, min_words=words[0], max_words=words[1])
        import S2VTModel, utils
        m = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
        m.load_state_dict(sd)
        m.to(DEV)
        f = feats.to(DEV).requires_grad_()
        logits = m(f, targets=caps[:, :-1].to(DEV), mode="train")
        loss = utils.MaskCriterion()(logits, caps.to(DEV), mask.to(DEV))
        loss.backward()
        om = orc.OracleModel(sd)
        fo = feats.clone().requires_grad_()
        ologits = om(fo, caps[:, :-1])
        oloss = orc.mask_criterion(ologits, caps, mask)
        oloss.backward()
        assert (logits.detach().cpu() - ologits.detach()).abs().max().item() < 2e-5
        assert abs(float(loss) - float(oloss)) < 1e-5
        for (n, p), (k, q) in zip(m.named_parameters(), om.as_dict().items()):
            assert n == k
            assert (p.grad.cpu() - q.grad).abs().max().item() <= 1e-6 + 1e-4 * q.grad.abs().max().item(), (n, B, H)
        assert (f.grad.cpu() - fo.grad).abs().max().item() <= 1e-6 + 1e-4 * fo.grad.abs().max().item()
        with torch.no_grad():
            ids = m.eval()(feats.to(DEV), mode="test").cpu()
        oids, marg = orc.greedy_decode(sd, feats, return_margins=True)
        if marg.min().item() > 1e-4:
            assert torch.equal(ids, oids)


@pytest.mark.parametrize("B,pad_min", [(16, 1), (10, 1), (16, 33), (10, 33), (100, 33), (40, 33)])
def test_reference_default_sizes_and_ragged_batches_match_the_oracle(lib, B, pad_min):
    """The reference's own defaults - batch_size = 16, dim_hidden = dim_embed = 512 (train.py:27-28,37), eval batch 10 (eval.py:27) -
    and ragged batches below / above 64 at L = 80, F = 4096.  From option pad_min_batch rows on (default 33; greedy decode: 24) the library pads the
    batch to a multiple of 64 inside its workspace (s2vt_padded_batch) so that it runs the plane GEMMs, the persistent
    recurrence and the decode cache; smaller batches run as they are on the launch-per-timestep driver (measured faster there:
    profiles/round5_ragged_batches.txt) - pad_min_batch = 1 forces the padded path at the reference's sizes too.  Train step
    (logits, loss, all 13 gradients and dfeats), greedy ids and beam captions against the oracle; the pad rows must not leak
    into anything (bias sums, embedding gradient, CE mean)."""
    import S2VTModel, utils
    from s2vt_video_caption_amd import beam, functional
    L, Fd, H, E, V = 80, 4096, 512, 512, 3000
    prev_pad = lib.s2vt_set_option(b"pad_min_batch", pad_min)
    try:
        _ragged_body(lib, B, pad_min, L, Fd, H, E, V)
    finally:
        lib.s2vt_set_option(b"pad_min_batch", prev_pad)
        functional.clear_decode_cache()


def _ragged_body(lib, B, pad_min, L, Fd, H, E, V):
    import S2VTModel, utils
    from s2vt_video_caption_amd import beam, functional
    padded = B >= pad_min
    assert lib.s2vt_padded_batch(B) == ((B + 63) // 64 * 64 if padded else B) and lib.s2vt_set_gemm_mode(-1) == 3
    sd = synth.make_state_dict(V, Fd, H, E, seed=40 + B)
    feats, caps, mask = synth.make_batch(B, L, Fd, V, seed=41 + B)
    m = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
    m.load_state_dict(sd)
    m.to(DEV)
    f = feats.to(DEV).requires_grad_()
    logits = m(f, targets=caps[:, :-1].to(DEV), mode="train")
    loss = utils.MaskCriterion()(logits, caps.to(DEV), mask.to(DEV))
    loss.backward()
    om = orc.OracleModel(sd)
    fo = feats.clone().requires_grad_()
    ologits = om(fo, caps[:, :-1])
    oloss = orc.mask_criterion(ologits, caps, mask)
    oloss.backward()
    assert (logits.detach().cpu() - ologits.detach()).abs().max().item() < 2e-5
    assert abs(float(loss) - float(oloss)) < 1e-5
    for (n, p), (k, q) in zip(m.named_parameters(), om.as_dict().items()):
        assert n == k
        assert (p.grad.cpu() - q.grad).abs().max().item() <= 1e-6 + 1e-4 * q.grad.abs().max().item(), (n, B)
    assert (f.grad.cpu() - fo.grad).abs().max().item() <= 1e-6 + 1e-4 * fo.grad.abs().max().item()
    m.eval()
    with torch.no_grad():
        ids = m(feats.to(DEV), mode="test").cpu()
        # the decode cache serves padded batches too; an unpadded small batch decodes launch per timestep and fills nothing
        assert m in functional._DECODE_CACHES and functional._DECODE_CACHES[m][2] == (B >= pad_min * 3 // 4)
        again = m(feats.to(DEV), mode="test").cpu()
    assert torch.equal(ids, again)
    oids, marg = orc.greedy_decode(sd, feats, return_margins=True)
    rows = (marg.reshape(B, -1).min(dim=1).values >= 1e-4).nonzero().flatten()       # (random-init weights: many rows rest on near-ties)
    assert len(rows) >= B // 4
    assert torch.equal(ids[rows], oids[rows])
    if B <= 16:
        with torch.no_grad():
            caps_dev = [[int(t) for t in s] for s in m(feats.to(DEV), mode="beam_search", beam_width=3, max_beam_depth=8)]
        from s2vt_video_caption_amd import capi as _capi
        if _capi.recurrence_plan(64, H)[0] == 3:                         # (the persistent forward is available under the options of this run)
            assert "precomputed" in beam.LAST_PATH, beam.LAST_PATH       # the library's encode phase + plane-path depth step
        nb = min(B, 6)            # (the oracle's search is a Python loop per sample)
        ocaps, gaps = orc.beam_search(sd, feats[:nb], beam_width=3, max_depth=8, return_gap="per_sample")
        for b in range(nb):
            assert gaps[b] < 1e-5 or caps_dev[b] == [int(t) for t in ocaps[b]], b


def test_full_size_properties_c2_shape(lib):
    """Size-independent properties at BASELINE full size (no oracle needed):
    batch independence (a sample's logits/ids do not depend on its batch mates, bitwise), determinism,
    and exact homogeneity of the backward in dlogits (scaling by 2 is exact in fp32)."""
    d = synth.CONFIGS["c2"]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=33)
    feats, caps, mask = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=77)
    m = _model(d, sd)
    f, c = feats.to(DEV), caps.to(DEV)
    with torch.no_grad():
        full = m(f, targets=c[:, :-1], mode="train")
        again = m(f, targets=c[:, :-1], mode="train")
        part = m(f[8:24], targets=c[8:24, :-1], mode="train")
        ids_full = m.eval()(f, mode="test")
        ids_part = m(f[40:48], mode="test")
    assert torch.equal(full, again)                       # deterministic: fixed summation orders, no atomics
    # a sample does not see its batch mates; only the split-K factor of the batched GEMMs (chosen from the grid
    # size) may change the fp32 summation order between batch sizes
    assert (full[8:24] - part).abs().max().item() < 3e-5
    # (ids of a sub-batch are compared on the screened fixtures, where no decision is a near-tie; these weights are unscreened)
    assert ids_part.shape == (8, d["L"] - 1)
    assert torch.isfinite(full).all()
    assert int(ids_full.min()) >= 0 and int(ids_full.max()) < d["V"]
    m.train()
    outs = []
    for scale in (1.0, 2.0):
        m.zero_grad()
        logits = m(f, targets=c[:, :-1], mode="train")
        gen = torch.Generator().manual_seed(5)
        dl = (torch.randn(logits.shape, generator=gen) * 1e-3).to(DEV)
        logits.backward(dl * scale)
        outs.append({n: p.grad.clone() for n, p in m.named_parameters()})
    for n in outs[0]:
        assert torch.equal(outs[1][n], 2 * outs[0][n]), n      # every reduction has a fixed order (no atomics anywhere)


def test_c5_dims_beam_and_greedy_against_reference_golden(lib, golden):
    """BASELINE config 5 dims (H=E=1000, V=12000), beam_size 5, depth 30: ALL token ids of the batched on-GPU beam search
    equal the reference's own per-sample Python beam search, and all greedy ids equal the reference's (4 captions: the
    reference needs ~20 s per caption).  The fixture is screened (oracle/make_golden.py): weakest greedy top-2 margin and
    weakest beam decision gap are both >= 1e-3."""
    g = golden("c5beam")
    assert float(g["greedy_margin"].min()) >= 1e-3 and float(g["beam_min_gap"]) >= 1e-3
    d = dict(synth.CONFIGS["c5"]); d["B"] = int(g["dims"][0])
    seed = int(g["seed"])
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=seed, out_scale=float(g["out_scale"]))
    feats, caps, mask = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=1234 + seed)
    m = _model(d, sd).eval()
    with torch.no_grad():
        ids = m(feats.to(DEV), mode="test").cpu().numpy()
        out = m(feats.to(DEV), mode="beam_search", beam_width=int(g["beam_width"]), max_beam_depth=30)
    np.testing.assert_array_equal(ids, g["greedy_ids"])
    for b, s in enumerate(out):
        assert [int(t.item()) for t in s] == [int(x) for x in g["beam_ids"][b] if x >= 0], b


def test_c5_full_batch_greedy_and_beam_against_reference_golden(lib, golden):
    """BASELINE configs[4] AT ITS OWN SIZE: B=128, H=E=1000, V=12000, beam_size 5, depth 30 (S2VTModel.py:56-61,149-240;
    eval.py:81-96) and the greedy decode of the same 128 samples (S2VTModel.py:82-110), against the reference's own output
    for every row (tests/golden/c5full.npz, oracle/make_golden.py c5full: 128 per-sample Python beam searches of the
    reference, ~17 s each).
    Greedy: all 128 x 79 ids bit-exact; the fixture's weakest top-2 margin is asserted here (>= 1e-3 at its out_scale).
    Beam: 31 non-cumulative score decisions per sample; among 128 samples some rest on near-ties of two log-probs, which no
    fp32 implementation with another summation order can be asked to reproduce.  The fixture records every sample's
    weakest decision gap (oracle replay); rows whose gap is below GATE are excused, everything else must be bit-exact, and
    the number of excused rows is asserted (<= 2 % of the batch, as is the number of rows that really differ)."""
    g = golden("c5full")
    d = dict(synth.CONFIGS["c5"])
    assert int(g["dims"][0]) == d["B"] == 128
    seed, scale = int(g["seed"]), float(g["out_scale"])
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=seed, out_scale=scale)
    feats, _, _ = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=1234 + seed)
    m = _model(d, sd).eval()
    with torch.no_grad():
        ids = m(feats.to(DEV), mode="test").cpu().numpy()
        out = m(feats.to(DEV), mode="beam_search", beam_width=int(g["beam_width"]), max_beam_depth=30)
    assert float(g["greedy_margin"].min()) >= 1e-3
    np.testing.assert_array_equal(ids, g["greedy_ids"])
    GATE = 1e-5                                  # score units at out_scale 16 (|logits| ~ 20: 5e-7 relative)
    gap = g["beam_gap"]
    excused = gap < GATE
    assert int(excused.sum()) <= 2, int(excused.sum())                       # <= 2 % of 128, a property of the fixture
    differ = []
    for b, s in enumerate(out):
        if [int(t.item()) for t in s] != [int(x) for x in g["beam_ids"][b] if x >= 0]:
            differ.append(b)
    assert all(excused[b] for b in differ), [(b, float(gap[b])) for b in differ if not excused[b]]
    assert len(differ) <= 2



def test_c5_unscaled_greedy_batch_against_reference_golden(lib, golden):
    """The UNSCALED companion of c5full (out_scale 1: the logit margins random-init weights really give, weakest 2.6e-4 here
    against the 4.1e-3 of the scaled fixture): the reference's greedy decode of the same 128 samples, tests/golden/c5raw.npz.
    Every row whose weakest top-2 margin in the reference is >= 1e-4 must be bit-exact over all 79 steps; the number of rows
    compared is asserted (this seed: all 128)."""
    g = golden("c5raw")
    d = dict(synth.CONFIGS["c5"])
    seed = int(g["seed"])
    assert float(g["out_scale"]) == 1.0 and int(g["dims"][0]) == d["B"] == 128
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=seed, out_scale=1.0)
    feats, _, _ = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=1234 + seed)
    m = _model(d, sd).eval()
    with torch.no_grad():
        ids = m(feats.to(DEV), mode="test").cpu().numpy()
    rows = np.nonzero(g["greedy_margin"].min(axis=1) >= 1e-4)[0]
    assert len(rows) >= 120, len(rows)
    np.testing.assert_array_equal(ids[rows], g["greedy_ids"][rows])


def test_c5_unscaled_beam_batch_against_reference_golden(lib, golden):
    """The UNSCALED companion of c5full's beam half (tests/golden/c5rawbeam.npz, oracle/make_golden.py c5rawbeam: the reference's
    128 beam searches - beam 5, depth 30, S2VTModel.py:149-240 - with the recipe's weights as they are).  With out_scale 1 the
    log-probs of a step lie within ~1e-3 of each other, so the 31 non-cumulative score decisions of a search are close calls by
    construction: the fixture's weakest decision gap is below 1e-4 for ALL 128 samples and below 1e-6 (two fp32 ulps of a score
    of 3) for 42 - which is why c5full scales the logits.  What this fixture pins: every sample whose weakest gap is >= GATE (one
    part in 3e6 of a score) must come out bit-exact, the number of samples compared is asserted, and at most 8 of the excused ones
    may differ (measured: all 128 identical, on the device-queue + plane-path default as on the host-queue / fp32-step paths).
    The model is fresh: the beam search is its FIRST call, so the plane path has to fill the weight-image cache itself (a lookup
    that only allocated the cache once made the filling call believe it was filled: every caption came out as one token)."""
    g = golden("c5rawbeam")
    d = dict(synth.CONFIGS["c5"])
    seed = int(g["seed"])
    assert float(g["out_scale"]) == 1.0 and int(g["dims"][0]) == d["B"] == 128
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=seed, out_scale=1.0)
    feats, _, _ = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=1234 + seed)
    m = _model(d, sd).eval()
    with torch.no_grad():
        out = m(feats.to(DEV), mode="beam_search", beam_width=int(g["beam_width"]), max_beam_depth=30)
    GATE = 1e-6
    gap = g["beam_gap"]
    compared = np.nonzero(gap >= GATE)[0]
    assert len(compared) >= 85, len(compared)
    same = np.array([[int(t.item()) for t in s] == [int(x) for x in g["beam_ids"][b] if x >= 0] for b, s in enumerate(out)])
    print("c5rawbeam: identical rows %d/128; differing gaps %s" % (int(same.sum()), np.sort(gap[~same])[-5:]))
    assert same[compared].all(), [(int(b), float(gap[b])) for b in compared if not same[b]]
    assert int(same.sum()) >= 120, int(same.sum())
    from s2vt_video_caption_amd import beam as _beam
    assert _beam.LAST_PATH.startswith("device queues + plane-path"), _beam.LAST_PATH


@pytest.mark.parametrize("B", [64, 192])
def test_beam_plane_path_fills_and_follows_the_weight_cache(lib, B):
    """The plane-path depth step (s2vt_beam_step_cached) reads weight images that mode='test' normally leaves behind.  Scenarios in
    which nobody has: the beam search is a fresh model's first call; the weights moved in place since the last decode (an optimiser
    step: version counters); a second model with other weights decodes in between.  In each the default path must give what the
    path that reads the fp32 parameters directly gives (beam.PLANE_STEP = False: s2vt_beam_step; >= 95 % of the captions identical -
    the two round differently, so a 1-ulp score tie may fall the other way) - stale or unfilled images give 0 %."""
    from s2vt_video_caption_amd import beam
    d = dict(synth.CONFIGS["c5"])
    feats = synth.make_batch(B, d["L"], d["F"], d["V"], seed=77)[0].to(DEV)

    def captions(m, plane):
        keep = beam.PLANE_STEP
        beam.PLANE_STEP = plane
        try:
            with torch.no_grad():
                out = m(feats, mode="beam_search", beam_width=5, max_beam_depth=12)
        finally:
            beam.PLANE_STEP = keep
        assert ("plane-path" in beam.LAST_PATH) == plane, beam.LAST_PATH
        return [[int(t.item()) for t in s] for s in out]

    def agree(a, b):
        return sum(x == y for x, y in zip(a, b)) / len(a)
    ma = _model(d, synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=21)).eval()
    mb = _model(d, synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=22)).eval()
    a1 = captions(ma, True)                                    # first call of a fresh model
    assert agree(a1, captions(ma, False)) >= 0.95
    gen = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        for p_ in ma.parameters():                             # what an optimiser step does: in place, version counters move
            p_.add_(0.02 * p_.abs().mean() * torch.randn(p_.shape, generator=gen).to(DEV))
    b1 = captions(mb, True)                                    # another model's images in between
    a2 = captions(ma, True)
    assert agree(a2, captions(ma, False)) >= 0.95
    assert agree(a2, a1) < 0.5                                 # (the weights did move)
    assert agree(b1, captions(mb, False)) >= 0.95
    assert captions(ma, True) == a2                            # and the cached images serve the next call


def test_library_encode_phase_is_the_beam_searchs_python_encoder(lib):
    """s2vt_decode_encode_cached (the encode phase of mode='test' handed out for the beam search, plus vid_rnn's token-independent
    decode steps): its four states and its per-depth gate inputs against the per-op encoder beam.py used before (s2vt_feat_proj_fwd,
    s2vt_gemm_f32, the sequence kernels) and an explicit vid_rnn roll-out in torch, to fp32-rounding bounds; and the beam search
    gives the same captions with the switches on and off (/root/reference/S2VTModel.py:56-60, 208-212)."""
    from s2vt_video_caption_amd import beam, functional, ops
    d = dict(synth.CONFIGS["c5"]); B = 64
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=31)
    feats = synth.make_batch(B, d["L"], d["F"], d["V"], seed=32)[0].to(DEV)
    m = _model(d, sd).eval()
    with _options(lib, **_PLAN_DEFAULTS), torch.no_grad():             # (the library's encode phase is the persistent one)
        caps_on = [[int(t.item()) for t in s] for s in m(feats, mode="beam_search", beam_width=5, max_beam_depth=10)]
        assert "precomputed" in beam.LAST_PATH
        keep = beam.PLANE_ENCODER
        beam.PLANE_ENCODER = False
        try:
            caps_off = [[int(t.item()) for t in s] for s in m(feats, mode="beam_search", beam_width=5, max_beam_depth=10)]
            assert "precomputed" not in beam.LAST_PATH
        finally:
            beam.PLANE_ENCODER = keep
    assert sum(a == b for a, b in zip(caps_on, caps_off)) >= int(0.95 * B)
    # states and gate inputs against torch fp64 on the same weights
    p64 = {k: v.double().to(DEV) for k, v in sd.items()}
    H, E, L = d["H"], d["E"], d["L"]
    x1 = feats.double() @ p64["feat_linear.weight"].T + p64["feat_linear.bias"]                      # [B, L, H]

    def run(x, wi, wh, bi, bh, h, c):
        hs = []
        for t in range(x.shape[1]):
            g = x[:, t] @ wi.T + h @ wh.T + bi + bh
            i, f, gg, o = g.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs.append(h)
        return torch.stack(hs, 1), h, c
    z = torch.zeros(B, H, dtype=torch.float64, device=DEV)
    h1s, h1, c1 = run(x1, p64["vid_rnn.weight_ih_l0"], p64["vid_rnn.weight_hh_l0"], p64["vid_rnn.bias_ih_l0"], p64["vid_rnn.bias_hh_l0"], z, z)
    x2 = torch.cat([torch.zeros(B, L, E, dtype=torch.float64, device=DEV), h1s], dim=2)
    _, h2, c2 = run(x2, p64["word_rnn.weight_ih_l0"], p64["word_rnn.weight_hh_l0"], p64["word_rnn.bias_ih_l0"], p64["word_rnn.bias_hh_l0"], z, z)
    depth = 6
    pad = torch.zeros(B, depth, H, dtype=torch.float64, device=DEV)
    hd, _, _ = run(pad, p64["vid_rnn.weight_ih_l0"], p64["vid_rnn.weight_hh_l0"], p64["vid_rnn.bias_ih_l0"], p64["vid_rnn.bias_hh_l0"], h1, c1)
    gx_ref = hd @ p64["word_rnn.weight_ih_l0"][:, E:].T + p64["word_rnn.bias_ih_l0"] + p64["word_rnn.bias_hh_l0"]   # [B, depth, 4H]
    plist = tuple(m.state_dict()[k] for k in ("vid_rnn.weight_ih_l0", "vid_rnn.weight_hh_l0", "vid_rnn.bias_ih_l0", "vid_rnn.bias_hh_l0",
                                              "word_rnn.weight_ih_l0", "word_rnn.weight_hh_l0", "word_rnn.bias_ih_l0", "word_rnn.bias_hh_l0",
                                              "feat_linear.weight", "feat_linear.bias", "out_linear.weight", "out_linear.bias",
                                              "embedding.weight"))
    with _options(lib, **_PLAN_DEFAULTS):
        out = functional.decode_encode(feats, plist, m, depth=depth)
    assert out is not None
    vh, vc, wh, wc, gx = out
    for got, ref in ((vh, h1), (vc, c1), (wh, h2), (wc, c2)):
        assert (got.double() - ref).abs().max().item() < 2e-6
    assert (gx.double() - gx_ref.transpose(0, 1)).abs().max().item() < 1e-5


@pytest.mark.parametrize("B", [64, 128, 192])
def test_decode_schedules_give_the_same_ids(lib, golden, B):
    """s2vt_set_decode_schedule: the fused schedule (h_t W_hh^T of step t+1 as extra row blocks of step t's argmax launch, then a
    cell-update launch - /root/reference/S2VTModel.py:98-107 reordered around the one data dependence the token has) and the
    step-kernel + argmax-kernel schedule decode the same ids, and both are the reference's on the rows of the unscaled fixture
    whose margins are resolvable (B <= 128: the first B samples of c5raw's batch)."""
    g = golden("c5raw")
    d = dict(synth.CONFIGS["c5"])
    seed = int(g["seed"])
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=seed, out_scale=1.0)
    feats128, _, _ = synth.make_batch(128, d["L"], d["F"], d["V"], seed=1234 + seed)
    feats = feats128[:B] if B <= 128 else torch.cat([feats128, feats128[:B - 128].flip(1)], dim=0)
    m = _model(d, sd).eval()
    prev = lib.s2vt_set_decode_schedule(-1)
    out = {}
    try:
        for sched in (1, 0):
            lib.s2vt_set_decode_schedule(sched)
            assert lib.s2vt_set_decode_schedule(-1) == sched
            with torch.no_grad():
                out[sched] = m(feats.to(DEV), mode="test").cpu().numpy()
    finally:
        lib.s2vt_set_decode_schedule(prev)
    np.testing.assert_array_equal(out[0], out[1])
    n = min(B, 128)
    rows = np.nonzero(g["greedy_margin"][:n].min(axis=1) >= 1e-4)[0]
    assert len(rows) >= n - 8
    np.testing.assert_array_equal(out[1][rows], g["greedy_ids"][rows])


@pytest.mark.parametrize("dims", [(64, 5, 70, 44, 28, 61), (128, 4, 36, 100, 52, 333), (192, 7, 24, 72, 40, 130), (256, 12, 24, 16, 8, 12)])
def test_decode_schedules_on_ragged_shapes_against_the_oracle(lib, dims):
    """The fused decode schedule at shapes where nothing is a multiple of a tile: 4H = 176 / 400 / 288 / 64 rows of W_hh planes (a
    partial last 64-row block, or exactly one), V = 61 / 333 / 130 / 12 vocabulary rows, H = 44 (k padded 44 -> 64).  Both
    schedules give the same ids, and on every row whose weakest top-2 margin in the oracle is >= 1e-4 they are the oracle's
    (oracle/s2vt_oracle.py::greedy_decode, the restatement of /root/reference/S2VTModel.py:82-110)."""
    B, L, Fd, H, E, V = dims
    sd = synth.make_state_dict(V, Fd, H, E, seed=11)
    feats, _, _ = synth.make_batch(B, L, Fd, V, seed=12)
    import S2VTModel
    m = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
    m.load_state_dict(sd)
    m.to(DEV).eval()
    prev = lib.s2vt_set_decode_schedule(-1)
    out = {}
    try:
        for sched in (1, 0):
            lib.s2vt_set_decode_schedule(sched)
            with torch.no_grad():
                out[sched] = m(feats.to(DEV), mode="test").cpu()
    finally:
        lib.s2vt_set_decode_schedule(prev)
    assert torch.equal(out[0], out[1])
    oids, marg = orc.greedy_decode(sd, feats, return_margins=True)
    rows = (marg.reshape(B, -1).min(dim=1).values >= 1e-4).nonzero().flatten()
    assert len(rows) >= B // 2, len(rows)
    assert torch.equal(out[1][rows], oids[rows])


def test_c3_full_size_bf16_against_reference_golden(lib, golden):
    """BASELINE configs[2] at its own size: B=256, L=80, F=4096, H=E=1000, V=12000 with s2vt_set_gemm_mode(1) (bf16
    operands for the batched GEMMs and the recurrence - k padded 1000 -> 1024, 4000 -> 4032 - fp32 accumulation, cell
    state and gradients), against ONE fp32 train step of the reference on the same seeded inputs (tests/golden/c3.npz).
    Stated bf16 bounds (round 4: three times what was measured): loss within 1e-3 (measured 2.7e-4: the mean over 20 224 rows
    averages the rounding out), logits slice within 2e-2 of the largest logit (6e-3), every gradient's norm within 0.06 % (0.02 % since the
    mean-CE scale's mantissa stays out of the bf16 planes: CeGradArgs::alpha_out; 0.21 % before), its first 32 entries within 2.5 % of
    their largest (0.7 %), and every FULL gradient within 1.5 % (relative L2; 0.48 %) and
    cosine 0.9999 (1 - 1e-5) of the fp32-equivalent arithmetic's, which is itself held to the reference at fp32 bounds.  Plus the size-independent properties: finite, deterministic (bitwise), batch-independent rows, and the
    persistent recurrence schedule equal to the launch-per-timestep one within bf16 re-rounding."""
    import utils
    g = golden("c3")
    d, sd, feats, caps, mask = _setup(g, "c3")
    f, c, k = feats.to(DEV), caps.to(DEV), mask.to(DEV)
    crit = utils.MaskCriterion()
    prev = lib.s2vt_set_gemm_mode(1)
    try:
        m = _model(d, sd)
        m.train()
        logits = m(f, targets=c[:, :-1], mode="train")
        loss = crit(logits, c, k)
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(logits).all()
        assert abs(float(loss) - float(g["losses"][0])) < 1e-3, (float(loss), float(g["losses"][0]))
        ref_rows = g["logits_rows"]
        assert np.abs(logits.detach()[:, ::13, :64].cpu().numpy() - ref_rows).max() < 2e-2 * np.abs(ref_rows).max()
        for key, p in m.named_parameters():
            gn = float(g["gradnorm/" + key])
            assert torch.isfinite(p.grad).all(), key
            assert abs(float(p.grad.double().norm()) - gn) <= 6e-4 * gn, (key, float(p.grad.double().norm()), gn)
            ref = g["gradhead/" + key]
            assert np.abs(p.grad.reshape(-1)[:32].cpu().numpy() - ref).max() <= 2.5e-2 * np.abs(ref).max() + 1e-9, key
        # DIRECTION of every full gradient (a dropped plane, a mis-scaled tile or a skipped k range of one GEMM moves a norm by
        # less than a per cent): the same step in the fp32-equivalent arithmetic (gemm mode 3) - itself held to the
        # reference's norms and leading entries at fp32 bounds here - and the cosine between the two, per parameter
        bf_grads = {key: p.grad.detach().clone() for key, p in m.named_parameters()}
        lib.s2vt_set_gemm_mode(3)
        m3 = _model(d, sd)
        m3.train()
        crit(m3(f, targets=c[:, :-1], mode="train"), c, k).backward()
        for key, p in m3.named_parameters():
            gn = float(g["gradnorm/" + key])
            assert abs(float(p.grad.double().norm()) - gn) <= 5e-4 * gn + 1e-7, key
            ref = g["gradhead/" + key]
            assert np.abs(p.grad.reshape(-1)[:32].cpu().numpy() - ref).max() <= 2e-6 + 5e-4 * np.abs(ref).max(), key
            a, b = bf_grads[key].double().reshape(-1), p.grad.double().reshape(-1)
            cos = float((a @ b) / (a.norm() * b.norm()))
            assert cos >= 0.9999, (key, cos)
            assert float((a - b).norm() / b.norm()) <= 1.5e-2, (key, float((a - b).norm() / b.norm()))
        del m3, bf_grads
        lib.s2vt_set_gemm_mode(1)
        with torch.no_grad():
            again = m(f, targets=c[:, :-1], mode="train")
            part = m(f[64:128], targets=c[64:128, :-1], mode="train")
            assert torch.equal(again, logits.detach())                   # deterministic
            # a sample does not see its batch mates (only the split-K factor of a batched GEMM may change the summation
            # order between batch sizes, and a re-rounded bf16 activation moves a logit by ~1e-3 of its scale)
            assert (part - logits.detach()[64:128]).abs().max().item() < 2e-2 * logits.detach().abs().max().item()
            before = lib.s2vt_set_recurrence_mode(0)
            try:
                per_step = m(f, targets=c[:, :-1], mode="train")
            finally:
                lib.s2vt_set_recurrence_mode(before)
            assert (per_step - logits.detach()).abs().max().item() < 2e-2 * logits.detach().abs().max().item()
            from s2vt_video_caption_amd import capi
            capi.check_async_error()
    finally:
        lib.s2vt_set_gemm_mode(prev)


def test_out_of_range_target_raises_index_error(lib):
    """The reference's nn.Embedding raises IndexError for a caption id outside the vocabulary (S2VTModel.py:71).  Here the
    check runs on the device: the error surfaces at the next synchronisation point (capi.check_async_error) or, at the
    latest, from the next forward / backward."""
    import utils
    from s2vt_video_caption_amd import capi
    d = synth.CONFIGS["tiny"]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=3)
    feats, caps, mask = synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=3)
    m = _model(d, sd)
    bad = caps.clone()
    bad[1, 2] = d["V"]                       # one id past the vocabulary
    m(feats.to(DEV), targets=bad[:, :-1].to(DEV), mode="train")
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        capi.check_async_error()
    capi.check_async_error()                 # reported once
    neg = caps.clone()
    neg[0, 1] = -1
    m(feats.to(DEV), targets=neg[:, :-1].to(DEV), mode="train")
    torch.cuda.synchronize()
    with pytest.raises(IndexError):          # ... or by the next forward
        m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")
    logits = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")       # a clean call works again
    capi.check_async_error()
    assert torch.isfinite(logits).all()
    # dp.train_step(check_errors=True), what train.py runs: the error is raised BEFORE optimizer.step() - a bad batch never
    # reaches the weights, as in the reference, whose nn.Embedding raises in the forward (S2VTModel.py:71, train.py:120-125)
    from s2vt_video_caption_amd import dp
    crit = utils.MaskCriterion()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with pytest.raises(IndexError):
        dp.train_step(m, crit, opt, feats.to(DEV), bad.to(DEV), mask.to(DEV), None, check_errors=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_out_of_range_loss_target_raises_index_error(lib):
    """The LAST caption column never passes through the embedding (targets[:, :-1] feeds the model) but it is a class index
    of the loss (utils.py:22 against target[:, 1:]): nn.CrossEntropyLoss raises for an id outside the vocabulary, here
    the loss kernel flags it and the error surfaces at the next synchronisation point."""
    import utils
    from s2vt_video_caption_amd import capi
    d = synth.CONFIGS["tiny"]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=3)
    feats, caps, mask = (t.to(DEV) for t in synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=3))
    m = _model(d, sd)
    crit = utils.MaskCriterion()
    logits = m(feats, targets=caps[:, :-1], mode="train")
    bad = caps.clone()
    bad[0, -1] = d["V"] + 7
    crit(logits, bad, mask)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        capi.check_async_error()
    loss = crit(logits, caps, mask)          # a clean call works again
    torch.cuda.synchronize()
    capi.check_async_error()
    assert torch.isfinite(loss)


@pytest.mark.parametrize("B,gemm_mode", [(5, 0), (5, 3), (64, 3), (64, 1)])
def test_out_dropout_train_mode_matches_oracle_with_the_same_mask(lib, B, gemm_mode):
    prev_pad = lib.s2vt_set_option(b"pad_min_batch", 1)       # (B = 5 in a plane mode: the padded path and its staged mask)
    try:
        _dropout_body(lib, B, gemm_mode)
    finally:
        lib.s2vt_set_option(b"pad_min_batch", prev_pad)


def _dropout_body(lib, B, gemm_mode):
    """out_dropout > 0 (S2VTModel.py:25,79): the decode-step hidden states are masked between word_rnn and out_linear.  With
    the SAME keep mask the oracle must give the same logits and the same 13 gradients (fp32 paths: gemm mode 0 = the fp32-MFMA
    driver, mode 3 the split-precision plane driver - B=5 padded to 64 inside the workspace, its mask staged time-major at the
    padded stride; gemm mode 1 = bf16 operands at bf16 bounds); at model level the mask is drawn like the reference draws it,
    eval mode ignores it."""
    import S2VTModel, utils
    from s2vt_video_caption_amd import functional as F
    L, Fd, H, E, V = 6, 48, 64, 40, 90
    sd = synth.make_state_dict(V, Fd, H, E, seed=8)
    feats, caps, mask = synth.make_batch(B, L, Fd, V, seed=8, min_words=1, max_words=3)
    p = 0.35
    keep = torch.nn.functional.dropout(torch.ones(B, L - 1, H), p, training=True)       # [B, L-1, H], 0 or 1/(1-p)
    om = orc.OracleModel(sd)
    ologits = om(feats, caps[:, :-1], out_mask=keep)
    oloss = orc.mask_criterion(ologits, caps, mask)
    oloss.backward()
    m = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E, out_dropout=p)
    m.load_state_dict(sd)
    m.to(DEV).train()
    prev = lib.s2vt_set_gemm_mode(gemm_mode)
    try:
        keep_tm = keep.transpose(0, 1).reshape((L - 1) * B, H).contiguous().to(DEV)
        logits = F.train_forward(feats.to(DEV), caps[:, :-1].to(DEV), m._hip_params(), out_mask=keep_tm)
        loss = utils.MaskCriterion()(logits, caps.to(DEV), mask.to(DEV))
        loss.backward()
        bf = gemm_mode == 1
        scale = ologits.detach().abs().max().item()
        assert (logits.detach().cpu() - ologits.detach()).abs().max().item() < (2e-2 * scale if bf else 2e-5)
        for (n, prm), (k, q) in zip(m.named_parameters(), om.as_dict().items()):
            if bf:
                assert (prm.grad.cpu() - q.grad).norm().item() <= 3e-2 * q.grad.norm().item() + 1e-9, n
            else:
                assert (prm.grad.cpu() - q.grad).abs().max().item() <= 1e-6 + 1e-4 * q.grad.abs().max().item(), n
        # model level: a mask is drawn (train mode differs from the no-dropout logits, two calls differ), eval ignores it
        with torch.no_grad():
            a = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")
            b = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")
            m.eval()
            c = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")
            d = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")
        assert not torch.equal(a, b) and torch.equal(c, d)
        if not bf:
            assert (c.cpu() - orc.forward_train(sd, feats, caps[:, :-1])).abs().max().item() < 2e-5
        # the mask is the one nn.Dropout draws for a [B, L-1, H] tensor from the same generator state
        m.train()
        torch.manual_seed(123)
        e = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train").detach()
        torch.manual_seed(123)
        keep2 = torch.nn.functional.dropout(torch.ones(B, L - 1, H, device=DEV), p, training=True)
        f = F.train_forward(feats.to(DEV), caps[:, :-1].to(DEV), m._hip_params(),
                            out_mask=keep2.transpose(0, 1).reshape((L - 1) * B, H).contiguous()).detach()
        assert torch.equal(e, f)
    finally:
        lib.s2vt_set_gemm_mode(prev)


def test_backward_refuses_a_workspace_of_another_mode(lib):
    """Changing the arithmetic mode between a forward and its backward would carve the workspace differently: refused."""
    import utils
    from s2vt_video_caption_amd import capi
    B, L, Fd, H, E, V = 64, 4, 64, 64, 64, 100
    sd = synth.make_state_dict(V, Fd, H, E, seed=9)
    feats, caps, mask = synth.make_batch(B, L, Fd, V, seed=9, min_words=1, max_words=2)
    import S2VTModel
    m = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
    m.load_state_dict(sd)
    m.to(DEV)
    logits = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")
    loss = utils.MaskCriterion()(logits, caps.to(DEV), mask.to(DEV))
    prev = lib.s2vt_set_gemm_mode(1 if lib.s2vt_set_gemm_mode(-1) != 1 else 3)
    try:
        with pytest.raises(capi.S2VTHipError):
            loss.backward()
    finally:
        lib.s2vt_set_gemm_mode(prev)


def test_bf16_mode_config3_arithmetic(lib):
    """BASELINE config 3 arithmetic (s2vt_set_gemm_mode(1)): bf16 operands for the batched GEMMs and the timestep
    kernels, fp32 accumulation / cell state / gradients.  Token ids are not compared in bf16; the loss and
    the gradients are compared with the fp32 oracle at bf16-level tolerances (8 mantissa bits: ~4e-3 relative per
    operand, averaged down over K)."""
    B, L, Fd, H, E, V, seed = 64, 6, 192, 128, 64, 500, 4
    sd = synth.make_state_dict(V, Fd, H, E, seed=seed)
    feats, caps, mask = synth.make_batch(B, L, Fd, V, seed=seed, min_words=1, max_words=3)
    import S2VTModel, utils
    om = orc.OracleModel(sd)
    ologits = om(feats, caps[:, :-1])
    oloss = orc.mask_criterion(ologits, caps, mask)
    oloss.backward()
    prev = lib.s2vt_set_gemm_mode(1)
    try:
        m = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
        m.load_state_dict(sd)
        m.to(DEV)
        logits = m(feats.to(DEV), targets=caps[:, :-1].to(DEV), mode="train")
        loss = utils.MaskCriterion()(logits, caps.to(DEV), mask.to(DEV))
        loss.backward()
        with torch.no_grad():
            ids_bf16_mode = m.eval()(feats.to(DEV), mode="test").cpu()
    finally:
        lib.s2vt_set_gemm_mode(prev)
    scale = ologits.detach().abs().max().item()
    assert (logits.detach().cpu() - ologits.detach()).abs().max().item() < 2e-2 * scale
    assert abs(float(loss) - float(oloss)) < 5e-3
    for (n, p), (k, q) in zip(m.named_parameters(), om.as_dict().items()):
        rel = (p.grad.cpu() - q.grad).norm().item() / (q.grad.norm().item() + 1e-12)
        assert rel < 3e-2, (n, rel)
    # greedy decode keeps fp32-equivalent arithmetic in every mode
    oids, marg = orc.greedy_decode(sd, feats, return_margins=True)
    if marg.min().item() > 1e-4:
        assert torch.equal(ids_bf16_mode, oids)


@pytest.mark.parametrize("cfg,B,gemm_mode", [("tiny", 3, None), ("c2", 64, None), ("c2", 64, "persist_x3_bwd"), ("c3", 256, 1)])
def test_dp_overlapped_allreduce_single_rank_rccl(lib, cfg, B, gemm_mode):
    """dp.FlatGradAllReducer.attach(): the backward writes its gradients straight into the flat buffer and the
    all-reduce of each gradient group is issued on a side stream behind s2vt_backward_wait_grads.  With one rank
    (RCCL world_size 1: sum == identity) two steps must leave exactly the parameters of the plain single-GPU loop;
    the c2 case (B=64) goes through the split-precision / two-lane driver, tiny through the fp32-MFMA one, and the c3 case
    (B=256, s2vt_set_gemm_mode(1)) through the PERSISTENT bf16 recurrence with the reducer attached - the combination in
    which "gradient group 0 is final" is re-recorded behind the last persistent BPTT launch, so that no communication
    kernel starts beside a launch that needs all of its workgroups resident (csrc/api.hip)."""
    import socket
    import torch.distributed as dist
    import utils
    from s2vt_video_caption_amd import capi, dp
    x3b = gemm_mode == "persist_x3_bwd"          # the split-precision configuration with its persistent BPTT (option persist_x3_bwd)
    if x3b:
        gemm_mode = None
        prev_x3b = lib.s2vt_set_option(b"persist_x3_bwd", 1)
    if gemm_mode is not None:
        prev_mode = lib.s2vt_set_gemm_mode(gemm_mode)
    try:
        _dp_overlapped_body(lib, cfg, B)
        capi.check_async_error()
    finally:
        if gemm_mode is not None:
            lib.s2vt_set_gemm_mode(prev_mode)
        if x3b:
            lib.s2vt_set_option(b"persist_x3_bwd", prev_x3b)


def capi_plan(B, H):
    from s2vt_video_caption_amd import capi
    return capi.recurrence_plan(B, H) if B % 64 == 0 else (0, 0)


def _dp_overlapped_body(lib, cfg, B):
    import socket
    import torch.distributed as dist
    import utils
    from s2vt_video_caption_amd import dp
    d = dict(synth.CONFIGS[cfg]); d["B"] = B
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=3)
    feats, caps, mask = (t.to(DEV) for t in synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=77))
    crit = utils.MaskCriterion()

    def run(use_reducer):
        m = _model(d, sd)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        red = dp.FlatGradAllReducer(m.parameters()).attach(m) if use_reducer else None
        losses = [float(dp.train_step(m, crit, opt, feats, caps, mask, red)) for _ in range(2)]
        torch.cuda.synchronize()
        return losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}

    ref_losses, ref_sd = run(False)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        losses, got = run(True)
        # no collective may be released beside a persistent BPTT launch: gradient group 0's event was recorded, for the last time,
        # behind every such launch of the backward (bf16 configuration: 7 launches; launch-per-timestep BPTT: none)
        import ctypes
        n_p, after = ctypes.c_int32(-1), ctypes.c_int32(-1)
        assert lib.s2vt_backward_order(ctypes.byref(n_p), ctypes.byref(after)) == 0
        assert after.value == n_p.value and (n_p.value > 0) == (capi_plan(d["B"], d["H"])[1] != 0), (n_p.value, after.value)
    finally:
        dist.destroy_process_group()
    # every reduction on the path has a fixed order (no atomics): the two loops must agree bitwise
    assert losses == ref_losses
    for k in ref_sd:
        assert torch.equal(got[k], ref_sd[k]), k


@pytest.mark.parametrize("cfg,gemm_mode", [("mid64", 3), ("mid64", 1), ("c2", 3)])
def test_fused_criterion_backward_is_bitwise_the_unfused_one(lib, cfg, gemm_mode):
    """MaskCriterion's backward fused into the model's (s2vt_mean_ce_backward_fused: the mean-CE gradient is evaluated from the
    logits inside the plane-split pass of the train workspace, no fp32 dlogits tensor; utils.py:22 under train.py:124) against the
    two-kernel route (functional.FUSE_CE = False): loss and all 13 gradients of two Adam steps bit for bit in the split-precision
    configuration; in the bf16 configuration equal up to the one systematic difference described below.  And the guard: a second consumer of the logits is refused, not silently dropped."""
    import utils
    from s2vt_video_caption_amd import capi, functional
    d = synth.CONFIGS[cfg]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=21)
    feats, caps, mask = (t.to(DEV) for t in synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=22))
    crit = utils.MaskCriterion()
    prev = lib.s2vt_set_gemm_mode(gemm_mode)
    keep = functional.FUSE_CE
    try:
        def run(fuse):
            functional.FUSE_CE = fuse
            m = _model(d, sd)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3)
            out = []
            for _ in range(2):
                opt.zero_grad()
                logits = m(feats, targets=caps[:, :-1], mode="train")
                loss = crit(logits, caps, mask)
                loss.backward()
                out.append((float(loss), {n: p.grad.clone() for n, p in m.named_parameters()}))
                opt.step()
            capi.check_async_error()
            return out
        ref, got = run(False), run(True)
        for (l0, g0), (l1, g1) in zip(ref, got):
            if gemm_mode != 1:
                assert l0 == l1
                for n in g0:
                    assert torch.equal(g0[n], g1[n]), n
            else:
                # bf16 operands: the fused route keeps the mantissa of gout / rows OUT of the bf16 planes (power-of-two scale in
                # dlogits, the mantissa as an fp32 factor on dh2 / dW_o: CeGradArgs::alpha_out), the two-kernel route rounds
                # (p - y) * gout / rows as a whole - for the one-hot entries that is one constant whose bf16 rounding scales the whole
                # gradient (up to 2^-9).  Same gradients up to that scale and bf16 rounding noise; out_linear.bias (fp32 sums) exact.
                assert abs(l0 - l1) <= 2e-3 * abs(l0)
                for n in g0:
                    a, b = g0[n].double().reshape(-1), g1[n].double().reshape(-1)
                    if n == "out_linear.bias":
                        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), n
                        continue
                    cos = float((a @ b) / (a.norm() * b.norm()))
                    assert cos >= 0.9999, (n, cos)
                    assert abs(float(b.norm() / a.norm()) - 1.0) <= 5e-3, (n, float(b.norm() / a.norm()))
        functional.FUSE_CE = True
        m = _model(d, sd)
        logits = m(feats, targets=caps[:, :-1], mode="train")
        loss = crit(logits, caps, mask) + 1e-3 * logits.sum()          # a second consumer of the logits
        with pytest.raises(capi.S2VTHipError):
            loss.backward()
    finally:
        functional.FUSE_CE = keep
        lib.s2vt_set_gemm_mode(prev)


@pytest.mark.parametrize("cfg,gemm_mode", [("mid64", 3), ("c2", 3), ("mid64", 1)])
def test_graph_replay_is_bitwise_the_eager_sequence(lib, cfg, gemm_mode):
    """s2vt_set_graph_mode(1): the launch sequences of s2vt_train_forward / s2vt_train_backward are captured into hipGraphs
    (second sighting of an argument set) and replayed; six Adam steps of train.py:116-127 must leave exactly the losses and
    parameters of the eager loop, and the later steps must really have been replays (the caching allocator hands the workspace,
    logits and gradient tensors back at the same addresses once the loop has settled)."""
    import ctypes
    import utils
    from s2vt_video_caption_amd import capi, dp
    d = synth.CONFIGS[cfg]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=31)
    feats, caps, mask = (t.to(DEV) for t in synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=32))
    crit = utils.MaskCriterion()
    prev = lib.s2vt_set_gemm_mode(gemm_mode)
    try:
        def run(graphs):
            lib.s2vt_set_graph_mode(1 if graphs else 0)
            m = _model(d, sd)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3)
            losses = [float(dp.train_step(m, crit, opt, feats, caps, mask, None)) for _ in range(6)]
            torch.cuda.synchronize()
            capi.check_async_error()
            return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}
        c0, r0 = ctypes.c_int64(0), ctypes.c_int64(0)
        lib.s2vt_graph_stats(ctypes.byref(c0), ctypes.byref(r0))
        ref_losses, ref_sd = run(False)
        losses, got = run(True)
        c1, r1 = ctypes.c_int64(0), ctypes.c_int64(0)
        lib.s2vt_graph_stats(ctypes.byref(c1), ctypes.byref(r1))
        assert losses == ref_losses
        for k in ref_sd:
            assert torch.equal(got[k], ref_sd[k]), k
        assert c1.value - c0.value >= 2 and r1.value - r0.value >= 4, (c1.value - c0.value, r1.value - r0.value)
    finally:
        lib.s2vt_set_graph_mode(0)
        lib.s2vt_set_gemm_mode(prev)


def test_decode_cache_follows_the_weights(lib, golden):
    """The weight-image cache of mode='test' (s2vt_greedy_decode_cached; functional._DECODE_CACHES): a second call on unchanged
    weights reuses the images and must give the ids of the first (= the reference's, c2 fixture); after the weights change - an
    in-place optimizer-style update, then load_state_dict - the key (data_ptr, _version) changes and the decode must be the one of
    the new weights, equal to what an uncached call computes."""
    from s2vt_video_caption_amd import functional
    g = golden("c2")
    d, sd, feats, caps, mask = _setup(g, "c2")
    f = feats.to(DEV)
    m = _model(d, sd).eval()
    keep = functional.DECODE_CACHE
    try:
        functional.DECODE_CACHE = True
        functional.clear_decode_cache()
        with torch.no_grad():
            a = m(f, mode="test").cpu().numpy()
            assert m in functional._DECODE_CACHES
            b = m(f, mode="test").cpu().numpy()                      # cached images
            np.testing.assert_array_equal(a, g["greedy_ids"])
            np.testing.assert_array_equal(b, g["greedy_ids"])
            m.out_linear.bias.add_(torch.linspace(-1.0, 1.0, d["V"], device=DEV))     # in-place update: version bump
            m.embedding.weight.mul_(0.5)
            c = m(f, mode="test").cpu()
            functional.DECODE_CACHE = False
            c_ref = m(f, mode="test").cpu()
            assert torch.equal(c, c_ref) and not np.array_equal(c.numpy(), a)
            functional.DECODE_CACHE = True
            m.load_state_dict(sd)                                    # back to the fixture's weights (in-place copy)
            e = m(f, mode="test").cpu().numpy()
            np.testing.assert_array_equal(e, g["greedy_ids"])
    finally:
        functional.DECODE_CACHE = keep
        functional.clear_decode_cache()



def test_decode_cache_filled_under_one_mode_serves_every_other(lib, golden):
    """The cache is keyed on the WEIGHTS; which images a call reads depends on its batch size, the recurrence mode and the
    pipeline block (the persistent encode phase reads the W_hh plane images, the launch-per-timestep schedule does not).  A call
    that fills the cache must therefore write every image: fill under a mode / batch that does not use the persistent kernels
    (recurrence mode 0; pipeline block 0; B = 192 at H = 1000, which the persistent forward does not take), then decode the
    fixture's batch under the default modes from that cache - the ids must be the reference's (round-3 advisor finding: the
    second call read torch.empty bytes as W_hh)."""
    from s2vt_video_caption_amd import functional
    g = golden("c2")
    d, sd, feats, caps, mask = _setup(g, "c2")
    f = feats.to(DEV)
    f192 = torch.cat([f, f, f], 0)
    m = _model(d, sd).eval()
    keep = functional.DECODE_CACHE
    mode0 = lib.s2vt_set_recurrence_mode(-1)
    blk0 = lib.s2vt_set_pipeline_block(0)                           # (returns the previous block length)
    lib.s2vt_set_pipeline_block(blk0)
    try:
        functional.DECODE_CACHE = True
        for how in ("recurrence_mode_0", "pipeline_block_0", "B192"):
            functional.clear_decode_cache()
            with torch.no_grad():
                if how == "recurrence_mode_0":
                    lib.s2vt_set_recurrence_mode(0)
                    first = m(f, mode="test").cpu().numpy()
                    lib.s2vt_set_recurrence_mode(mode0)
                elif how == "pipeline_block_0":
                    lib.s2vt_set_pipeline_block(0)
                    first = m(f, mode="test").cpu().numpy()
                    lib.s2vt_set_pipeline_block(blk0)
                else:
                    first = m(f192, mode="test").cpu().numpy()[:d["B"]]
                np.testing.assert_array_equal(first, g["greedy_ids"], err_msg=how)
                key_before = functional._DECODE_CACHES[m][0]
                second = m(f, mode="test").cpu().numpy()            # default modes, images from the cache filled above
                assert functional._DECODE_CACHES[m][0] == key_before, "the second call must have used the cache"
                np.testing.assert_array_equal(second, g["greedy_ids"], err_msg=how + " -> default")
    finally:
        lib.s2vt_set_recurrence_mode(mode0)
        lib.s2vt_set_pipeline_block(blk0)
        functional.DECODE_CACHE = keep
        functional.clear_decode_cache()


def _c2_step_summary(lib):
    from s2vt_video_caption_amd import capi
    d = synth.CONFIGS["c2"]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=21)
    feats, caps, mask = (t.to(DEV) for t in synth.make_batch(d["B"], d["L"], d["F"], d["V"], seed=22))
    m = _model(d, sd).train()
    loss = utils_mod().MaskCriterion()(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
    loss.backward()
    torch.cuda.synchronize()
    capi.check_async_error()
    return {"plan": list(capi.recurrence_plan(d["B"], d["H"])), "loss": float(loss.detach()),
            "norms": {k: float(p.grad.double().norm()) for k, p in m.named_parameters()},
            "heads": {k: p.grad.reshape(-1)[:8].cpu().tolist() for k, p in m.named_parameters()}}


def utils_mod():
    import utils
    return utils


def test_persistent_bptt_is_the_launch_per_timestep_backward_within_fp32_rounding(lib):
    """Option persist_x3_bwd (the split-precision reduce-scatter BPTT) against the launch-per-timestep BPTT through the whole
    config-2 train step: same loss bits (the forward is unchanged), every gradient within fp32 rounding."""
    outs = []
    for flag in (0, 1):
        with _options(lib, **{**_PLAN_DEFAULTS, "persist_x3_bwd": flag}):
            outs.append(_c2_step_summary(lib))
    base, opt = outs
    assert base["plan"] == [3, 0] and opt["plan"] == [3, 3]
    assert base["loss"] == opt["loss"]
    for k, n in base["norms"].items():
        assert abs(opt["norms"][k] - n) <= 2e-5 * n + 1e-9, k
        h0, h1 = np.array(base["heads"][k]), np.array(opt["heads"][k])
        assert np.abs(h0 - h1).max() <= 1e-6 + 2e-4 * np.abs(h0).max(), k
