"""CPU: the drop-in `dataloader.VideoDataset` keeps the reference's item contract
(dataloader.py:28-50) and its caption-sampling RNG stream.  The reference's own `__getitem__` cannot run on
NumPy >= 1.24 (`np.random.choice` on a ragged list raises), so its rule is restated here on an object array."""
import json
import os

import numpy as np
import torch


def _make(tmp_path, n=6, L=8, F=16):
    rng = np.random.RandomState(0)
    caps, ids = {}, []
    os.makedirs(tmp_path / "feats")
    for i in range(n):
        vid = "vid%d" % i
        ids.append(vid)
        np.save(tmp_path / "feats" / (vid + ".npy"), rng.randn(L, F).astype(np.float32))
        caps[vid] = [[3] + list(rng.randint(5, 20, size=rng.randint(1, 12))) + [4] for _ in range(rng.randint(2, 5))]
        caps[vid] = [[int(x) for x in c] for c in caps[vid]]
    data = {"word2ix": {str(i): i for i in range(20)}, "ix2word": {str(i): str(i) for i in range(20)},
            "captions": caps, "splits": {"train": ids[:4], "valid": ids[4:5], "test": ids[5:]}}
    with open(tmp_path / "captions.json", "w") as f:
        json.dump(data, f)
    return data


def test_item_contract_and_sampling_stream(tmp_path):
    import dataloader
    data = _make(tmp_path)
    ds = dataloader.VideoDataset(str(tmp_path / "captions.json"), str(tmp_path / "feats"), max_len=8, mode="train")
    assert len(ds) == 4
    np.random.seed(123)
    items = [ds[i] for i in range(len(ds))]
    # the reference's rule (dataloader.py:41) on an object array, same seed
    np.random.seed(123)
    for (feat, pad_label, vid, mask), path in zip(items, ds.feat_paths):
        labels = data["captions"][vid]
        obj = np.empty(len(labels), dtype=object)
        for i, c in enumerate(labels):
            obj[i] = c
        label = np.random.choice(obj, 1)[0][:8]
        assert vid == path.stem
        assert feat.dtype == torch.float32 and tuple(feat.shape) == (8, 16)
        assert torch.equal(feat, torch.from_numpy(np.load(str(path))))
        assert pad_label.dtype == torch.int64 and pad_label.tolist() == list(label) + [0] * (8 - len(label))
        assert mask.dtype == torch.float32 and mask.tolist() == [1.0] * len(label) + [0.0] * (8 - len(label))
    # truncation of captions longer than max_len (dataloader.py:43-44)
    assert all(int(m.sum()) <= 8 for _, _, _, m in items)


def test_default_collate_and_feed_on_cpu(tmp_path):
    import dataloader
    _make(tmp_path)
    ds = dataloader.VideoDataset(str(tmp_path / "captions.json"), str(tmp_path / "feats"), max_len=8, mode="train")
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)
    batches = list(dataloader.feed_batches(loader, dev=torch.device("cpu")))
    assert len(batches) == 2
    feats, targets, ids, masks = batches[0]
    assert tuple(feats.shape) == (2, 8, 16) and tuple(targets.shape) == (2, 8) and tuple(masks.shape) == (2, 8)
    assert len(ids) == 2


import pytest


@pytest.mark.gpu
def test_feed_batches_on_gpu_matches_host(tmp_path):
    import dataloader
    _make(tmp_path, n=23, L=8, F=16)
    dev = torch.device("cuda", 0)
    ds = dataloader.VideoDataset(str(tmp_path / "captions.json"), str(tmp_path / "feats"), max_len=8, mode="test")   # 18 items
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)
    np.random.seed(7)
    host = list(loader)
    np.random.seed(7)
    n = 0
    # many more batches than ring slots (depth + 2), a ragged last batch; a yielded batch is a view of the ring that
    # stays valid until the next one is requested, so each is checked (after some stream work) before moving on
    for (f, t, ids, m), (hf, ht, hids, hm) in zip(dataloader.feed_batches(loader, dev=dev, depth=2), host):
        assert f.is_cuda and t.is_cuda and m.is_cuda
        g = (f * 2.0).sum()                              # work on the consumer stream that reads the batch
        assert torch.equal(f.cpu(), hf) and torch.equal(t.cpu(), ht) and torch.equal(m.cpu(), hm) and ids == hids
        assert abs(float(g) - float((hf * 2.0).sum())) < 1e-2
        n += 1
    assert n == len(host) and n > 4


@pytest.mark.gpu
def test_train_harness_end_to_end(tmp_path):
    """train.py on a toy dataset: two epochs, best-loss and final full-module checkpoints that load back through the
    drop-in class and decode (the reference's eval.py:41 protocol)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = _make(tmp_path, n=8, L=8, F=16)
    data["word2ix"].update({"<pad>": 0, "<unk>": 1, "<sos>": 3, "<eos>": 4})
    data["splits"] = {"train": ["vid%d" % i for i in range(6)], "valid": ["vid6"], "test": ["vid7"]}
    json.dump(data, open(tmp_path / "captions.json", "w"))
    ck = tmp_path / "ck"
    r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--caption-file", str(tmp_path / "captions.json"),
                        "--feats-path", str(tmp_path / "feats"), "--train-length", "8", "--dim-hidden", "32",
                        "--dim-embed", "24", "--feat-dim", "16", "--batch-size", "3", "--epochs", "2", "--save-path",
                        str(ck)], capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(os.listdir(ck))
    assert any(f.endswith("stop.pth") for f in files) and any(f.endswith("final.pth") for f in files)
    sys.path.insert(0, root)
    import S2VTModel  # noqa: F401
    m = torch.load(os.path.join(ck, [f for f in files if f.endswith("final.pth")][0]), weights_only=False).to("cuda:0")
    ids = m.eval()(torch.randn(2, 8, 16, device="cuda:0"), mode="test")
    assert tuple(ids.shape) == (2, 7)
