"""Rows f3 / f4 of SURVEY.md §8: the training harness (train.py) and caption generation (eval.py) of the product against
the oracle's restatement of the reference's epoch loop (oracle/train_oracle.py: train.py:104-175, utils.py:29-80) and of
its decode + id->word rules (eval.py:41-58, :81-96), on a toy dataset small enough for the CPU oracle.

Both sides read the SAME files through `dataloader.VideoDataset` in file order (--no-shuffle) with the same numpy seed,
so they see identical batches (the caption-sampling RNG stream is pinned in tests/test_dataloader.py) and start from the
same seeded weights."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import s2vt_oracle as orc
from oracle import train_oracle
from s2vt_video_caption_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, F, H, E, BS = 8, 24, 32, 24, 4
EPOCHS, LR, LR_PAT, ES_PAT, SAVE_FREQ, SEED = 14, 5e-3, 1, 3, 4, 7


def make_toy(root, n_train=12, n_valid=6, n_test=4, V=30, seed=0):
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "feats"), exist_ok=True)
    caps, ids = {}, []
    for i in range(n_train + n_valid + n_test):
        vid = "vid%02d" % i
        ids.append(vid)
        np.save(os.path.join(root, "feats", vid + ".npy"), rng.randn(L, F).astype(np.float32))
        caps[vid] = [[3] + [int(x) for x in rng.randint(5, V, size=rng.randint(2, 6))] + [4] for _ in range(rng.randint(2, 4))]
    w2i = {"<pad>": 0, "<unk>": 1, "<sos>": 3, "<eos>": 4}
    for i in range(V):
        if i not in (0, 1, 3, 4):
            w2i["w%d" % i] = i
    data = {"word2ix": w2i, "ix2word": {str(v): k for k, v in w2i.items()}, "captions": caps,
            "splits": {"train": ids[:n_train], "valid": ids[n_train:n_train + n_valid], "test": ids[n_train + n_valid:]}}
    with open(os.path.join(root, "captions.json"), "w") as f:
        json.dump(data, f)
    return data


def oracle_history(root, sd):
    import dataloader
    cf, fp = os.path.join(root, "captions.json"), os.path.join(root, "feats")
    tr = dataloader.VideoDataset(cf, fp, max_len=L)
    va = dataloader.VideoDataset(cf, fp, max_len=L, mode="valid")
    np.random.seed(SEED)

    def batches(ds):
        def per_epoch(epoch):
            for feats, targets, ids, masks in torch.utils.data.DataLoader(ds, batch_size=BS, shuffle=False):
                yield feats, targets, masks
        return per_epoch
    return train_oracle.run_epochs(sd, batches(tr), batches(va), EPOCHS, lr=LR, lr_patience=LR_PAT, es_patience=ES_PAT,
                                   save_freq=SAVE_FREQ)


def test_oracle_epoch_loop_takes_every_branch(tmp_path):
    """CPU: the toy run is a meaningful fixture - it improves, plateaus, cuts the learning rate, writes periodic and
    best-loss checkpoints and stops early - and no decision in it is a near-tie."""
    data = make_toy(str(tmp_path))
    sd = synth.make_state_dict(len(data["word2ix"]), F, H, E, seed=1)
    h = oracle_history(str(tmp_path), sd)
    assert h["stopped_at"] is not None and h["stopped_at"] < EPOCHS - 1
    assert min(h["lr"]) < LR                                   # ReduceLROnPlateau fired
    assert h["checkpoints"].count("stop.pth") >= 3 and "0.pth" in h["checkpoints"] and h["checkpoints"][-1] == "final.pth"
    v = h["valid_loss"]
    best = np.minimum.accumulate(v)
    assert min(abs(v[i] - best[i - 1]) for i in range(1, len(v))) > 1e-2     # every better/worse decision is clear-cut


@pytest.mark.gpu
def test_train_harness_and_generate_match_oracle(tmp_path):
    sys.path.insert(0, ROOT)
    import eval as s2vt_eval
    import train
    data = make_toy(str(tmp_path))
    V = len(data["word2ix"])
    sd = synth.make_state_dict(V, F, H, E, seed=1)
    torch.save(sd, tmp_path / "init.pt")
    ck = tmp_path / "ck"
    opt = train.parse(["--caption-file", str(tmp_path / "captions.json"), "--feats-path", str(tmp_path / "feats"),
                       "--train-length", str(L), "--dim-hidden", str(H), "--dim-embed", str(E), "--feat-dim", str(F),
                       "--batch-size", str(BS), "--epochs", str(EPOCHS), "--lr", str(LR), "--learning-rate-patience",
                       str(LR_PAT), "--early-stopping-patience", str(ES_PAT), "--save-freq", str(SAVE_FREQ),
                       "--save-path", str(ck), "--no-shuffle", "--seed", str(SEED), "--init-state", str(tmp_path / "init.pt")])
    got = train.run(opt)
    ref = oracle_history(str(tmp_path), sd)
    # ---- f3: the epoch loop
    assert got["stopped_at"] == ref["stopped_at"]
    assert got["lr"] == ref["lr"]                                        # same LR in force in every epoch
    assert got["checkpoints"] == ref["checkpoints"]                      # same files, same order
    assert len(got["train_loss"]) == len(ref["train_loss"])
    # up to 33 Adam steps at lr 5e-3 lie between the first and the last number: fp32 summation-order differences of the
    # two implementations (~1e-7 per step) are amplified by Adam's normalised update; 1e-4 is north_star's bound
    assert np.abs(np.array(got["train_loss"]) - np.array(ref["train_loss"])).max() < 1e-4, (got["train_loss"], ref["train_loss"])
    assert np.abs(np.array(got["valid_loss"]) - np.array(ref["valid_loss"])).max() < 1e-4, (got["valid_loss"], ref["valid_loss"])
    files = sorted(os.listdir(ck))
    st = got["start_time"]
    assert sorted([st + n for n in set(got["checkpoints"])] + [st + "opt.txt"]) == files      # + the run's configuration (save_opt)
    final = torch.load(ck / (st + "final.pth"), weights_only=False)
    for k, v in final.state_dict().items():
        assert (v.cpu() - ref["final_state"][k]).abs().max().item() < 2e-4, k
    # ---- f4: eval.generate (greedy, then beam search) from the best-loss checkpoint against the oracle's decode of the
    # same weights and the reference's id -> word rules (eval.py:54-58, :90-96)
    stop = ck / (st + "stop.pth")
    best = torch.load(stop, weights_only=False)
    bsd = {k: v.cpu() for k, v in best.state_dict().items()}
    ix2word = data["ix2word"]
    import dataloader
    test_ds = dataloader.VideoDataset(str(tmp_path / "captions.json"), str(tmp_path / "feats"), max_len=L, mode="test")
    feats = torch.stack([test_ds[i][0] for i in range(len(test_ds))])
    vids = [test_ds[i][2] for i in range(len(test_ds))]

    def words(ids, drop_sos):
        w = [ix2word[str(int(i))] for i in ids]
        if "<eos>" in w:
            w = w[:w.index("<eos>")]
        if drop_sos and "<sos>" in w:
            w.remove("<sos>")
        return " ".join(w)
    oids, marg = orc.greedy_decode(bsd, feats, return_margins=True)
    assert marg.min().item() > 1e-3                                       # the trained toy model decides clearly
    want = {v: words(oids[i].tolist(), False) for i, v in enumerate(vids)}
    got_g = s2vt_eval.generate(str(stop), str(tmp_path / "captions.json"), str(tmp_path / "feats"), batch_size=3, mode="test")
    assert got_g == want
    obeam, gap = orc.beam_search(bsd, feats, beam_width=3, max_depth=30, return_gap=True)
    want_b = {v: words(obeam[i], True) for i, v in enumerate(vids)}
    got_b = s2vt_eval.generate(str(stop), str(tmp_path / "captions.json"), str(tmp_path / "feats"), batch_size=3,
                               mode="beam_search", beam_width=3)
    if gap > 1e-4:
        assert got_b == want_b
    else:       # a tie in the reference's own ranking: compare everything the tie cannot touch
        assert set(got_b) == set(want_b)


@pytest.mark.gpu
def test_train_and_generate_with_the_attention_baseline(tmp_path):
    """train.py --model att_baseline (the network the reference's committed train.py:86 builds) through the same harness: the
    loss falls, the checkpoints are full-module pickles of attention_baseline.Att_Baseline, and eval.generate decodes from
    one of them exactly what a direct mode='test' call gives (the model's arithmetic is pinned to the reference in
    tests/test_gpu_att_baseline.py; the epoch loop above)."""
    sys.path.insert(0, ROOT)
    import eval as s2vt_eval
    import train
    data = make_toy(str(tmp_path))
    ck = tmp_path / "ck"
    opt = train.parse(["--caption-file", str(tmp_path / "captions.json"), "--feats-path", str(tmp_path / "feats"),
                       "--train-length", str(L), "--dim-hidden", str(H), "--dim-embed", str(E), "--feat-dim", str(F),
                       "--batch-size", str(BS), "--epochs", "6", "--lr", str(LR), "--save-freq", "3", "--save-path", str(ck),
                       "--no-shuffle", "--seed", str(SEED), "--model", "att_baseline"])
    got = train.run(opt)
    tl = got["train_loss"]
    assert len(tl) == 6 and all(np.isfinite(tl)) and tl[-1] < tl[0] - 0.1, tl
    st = got["start_time"]
    final = ck / (st + "final.pth")
    m = torch.load(final, weights_only=False)
    import attention_baseline
    assert isinstance(m, attention_baseline.Att_Baseline)
    import dataloader
    test_ds = dataloader.VideoDataset(str(tmp_path / "captions.json"), str(tmp_path / "feats"), max_len=L, mode="test")
    feats = torch.stack([test_ds[i][0] for i in range(len(test_ds))]).to("cuda:0")
    ids = m.to("cuda:0").eval()(feats, mode="test").cpu()
    ix2word = data["ix2word"]

    def words(row):
        w = [ix2word[str(int(i))] for i in row]
        return " ".join(w[:w.index("<eos>")] if "<eos>" in w else w)
    want = {test_ds[i][2]: words(ids[i].tolist()) for i in range(len(test_ds))}
    got_g = s2vt_eval.generate(str(final), str(tmp_path / "captions.json"), str(tmp_path / "feats"), batch_size=3, mode="test")
    assert got_g == want
