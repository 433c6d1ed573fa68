"""GPU parity of the drop-in `attention_baseline.Att_Baseline` (the reference's second network, attention_baseline.py:9-105,
SURVEY.md §8 row f4) against outputs of the reference itself (tests/golden/att_*.npz, made by oracle/make_golden.py gen_att):
train-mode logits, MaskCriterion loss, the gradient of every parameter - exactly zero for the three attention layers, whose
softmax runs over a dimension of size one upstream - and the greedy ids of mode='test'."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup(name):
    import attention_baseline
    from s2vt_video_caption_amd import synth
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, L, F, H, E, V = (int(x) for x in g["dims"])
    seed = int(g["seed"])
    torch.manual_seed(seed)                    # the drop-in creates its containers in the reference's order: same default init
    m = attention_baseline.Att_Baseline(V, F, L, dim_hid=H, dim_embed=E)
    if "param/feat_linear.weight" in g.files:
        m.load_state_dict({k: torch.from_numpy(g["param/" + k]) for k in g["keys"]})
    feats, caps, mask = synth.make_batch(B, L, F, V, seed=900 + seed)
    return g, m.to(DEV), feats.to(DEV), caps.to(DEV), mask.to(DEV)


@pytest.mark.parametrize("name", ["att_tiny", "att_mid", "att_full"])
def test_att_baseline_train_step_against_reference(lib, name):
    import utils
    g, m, feats, caps, mask = _setup(name)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    m.train()
    logits = m(feats, targets=caps[:, :-1], mode="train")
    loss = utils.MaskCriterion()(logits, caps, mask)
    loss.backward()
    lg = logits.detach().cpu()
    if "logits" in g.files:
        assert (lg.numpy() - g["logits"]).__abs__().max() < 2e-5
    else:
        assert np.abs(lg[:, ::3, :64].numpy() - g["logits_rows"]).max() < 2e-5
        assert abs(lg.double().abs().sum().item() - float(g["logits_abs_sum"])) < 1e-5 * float(g["logits_abs_sum"])
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        gn = float(g["gradnorm/" + k])
        got = p.grad.detach().cpu()
        if k.startswith("att_"):                 # zeros upstream as well (not None): a softmax over one element has no gradient
            assert gn == 0.0 and float(got.abs().max()) == 0.0, k
            continue
        assert abs(float(got.double().norm()) - gn) <= 2e-4 * gn + 1e-7, k
        ref = g["gradhead/" + k]
        assert np.abs(got.reshape(-1)[:32].numpy() - ref).max() <= 2e-6 + 5e-4 * np.abs(ref).max(), k
        if "grad/" + k in g.files:
            full = g["grad/" + k]
            assert np.abs(got.numpy() - full).max() <= 2e-6 + 2e-4 * np.abs(full).max(), k


@pytest.mark.parametrize("name", ["att_tiny", "att_mid", "att_full"])
def test_att_baseline_greedy_ids_against_reference(lib, name):
    """mode='test': L greedy steps from <sos>.  A row is compared up to (not including) its first decision whose top-2 logit
    margin in the reference is below 2e-5 - the bound the train-mode logits are held to above; a flip below it is legitimate and
    changes everything after it - ;
    at least 90 % of all decisions must be compared."""
    g, m, feats, _, _ = _setup(name)
    m.eval()
    ids = m(feats, mode="test").cpu().numpy()
    ref, marg = g["ids"], g["margins"]
    assert ids.shape == ref.shape and ids.dtype == np.int64
    compared = 0
    for b in range(ref.shape[0]):
        weak = np.nonzero(marg[b] < 2e-5)[0]
        stop = int(weak[0]) if len(weak) else ref.shape[1]
        assert (ids[b, :stop] == ref[b, :stop]).all(), b
        compared += stop
    assert compared >= 0.9 * ref.size, compared


def test_att_baseline_second_backward_and_wrong_shapes_raise(lib):
    import utils
    g, m, feats, caps, mask = _setup("att_tiny")
    loss = utils.MaskCriterion()(m(feats, targets=caps[:, :-1], mode="train"), caps, mask)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError):
        loss.backward()
    with pytest.raises(ValueError):
        m(feats[:, :-1], targets=caps[:, :-1], mode="train")
    with pytest.raises(ValueError):
        m(feats, mode="train")
    from s2vt_video_caption_amd import capi
    with pytest.raises(capi.S2VTHipError):
        m(feats.cpu(), targets=caps[:, :-1].cpu(), mode="train")
