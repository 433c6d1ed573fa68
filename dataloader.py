"""Drop-in `dataloader` module: `VideoDataset` with the reference's constructor, item layout
and caption-sampling rule (dataloader.py:11-53), plus a batch feed that keeps the GPU busy.

Same contract as the reference per item: `(feat [max_len... as stored, 4096] f32, pad_label [max_len] i64, ID str,
mask [max_len] f32)`, a caption drawn with numpy's GLOBAL RNG exactly as `np.random.choice(labels, 1)[0]` draws it
(dataloader.py:41; that call no longer accepts ragged caption lists on NumPy >= 1.24, so the equivalent
`randint`-based draw is used: it consumes the RNG stream identically — tests/test_dataloader.py).

Differences, both invisible to `train.py`: items are produced as pinned HOST tensors unless `device_items=True`
(the reference builds four device tensors per item, dataloader.py:38,45-47, one small H2D copy each), and features
are memory-mapped.  `feed_batches` then moves whole batches with one async copy per tensor on a side stream while
the previous batch trains.
"""
import json
import pathlib as plb

import numpy as np
import os

import torch
from torch.utils.data import Dataset

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')


class VideoDataset(Dataset):
    def __init__(self, captions_file, feat_path, max_len=80, mode='train', device_items=False):
        with open(captions_file, encoding='utf-8') as f:
            data = json.load(f)
            self.word2ix = data['word2ix']
            self.ix2word = data['ix2word']
            self.captions = data['captions']  # {video id: [caption token lists]}
            self.splits = data['splits']
        keep = set(self.splits[mode])
        self.feat_paths = [p for p in plb.Path(feat_path).glob('*.npy') if p.stem in keep]
        self.max_len = max_len
        self.device_items = device_items
        print("prepare {} dataset. vocab_size: {}, dataset_size: {}".format(mode, len(self.word2ix), len(self.feat_paths)))

    def __getitem__(self, index):
        ID = self.feat_paths[index].stem
        feat = torch.from_numpy(np.load(str(self.feat_paths[index]), mmap_mode='r').astype(np.float32, copy=True))
        labels = self.captions[ID]
        label = labels[int(np.random.randint(0, len(labels), size=1)[0])]      # == np.random.choice(labels, 1)[0]
        if len(label) > self.max_len:
            label = label[:self.max_len]
        pad_label = torch.zeros([self.max_len], dtype=torch.long)
        pad_label[:len(label)] = torch.tensor(label, dtype=torch.long)
        mask = torch.zeros([self.max_len], dtype=torch.float)
        mask[:len(label)] = 1
        if self.device_items:       # the reference's per-item device tensors (feat is a leaf requiring grad there)
            feat = feat.to(device).requires_grad_(True)
            pad_label, mask = pad_label.to(device), mask.to(device)
        return feat, pad_label, ID, mask

    def __len__(self):
        return len(self.feat_paths)


COPY_THREADS = int(os.environ.get("S2VT_FEED_THREADS", "4"))      # host threads staging a batch into pinned memory


def feed_batches(loader, dev=None, depth=2):
    """Iterate a `torch.utils.data.DataLoader` over a host-item `VideoDataset`, yielding `(feats, targets, IDs, masks)`
    with the tensors already resident on `dev`.

    A background thread pulls batches from the loader, stages them in a ring of PERSISTENT pinned buffers and copies
    them on a side stream into a ring of PERSISTENT device buffers, up to `depth` batches ahead; the training thread
    only waits on the copy's event.  Nothing is allocated per batch (pinning 84 MB per step costs tens of ms, and device
    tensors allocated on the copy stream every step make the caching allocator stall the training thread), so the host
    memcpy, the PCIe transfer and the previous training step overlap (measured: tools/bench_feed.py).  The yielded
    tensors are views of the ring: they stay valid until the NEXT batch is requested from the generator - the contract
    of a training loop that consumes one batch per step.  At the target rates (>= 3e5 frames/s = 5 GB/s of fp32 features)
    the reference's per-item synchronous copies would bound training."""
    dev = dev or device
    if dev.type != 'cuda':
        for batch in loader:
            yield batch
        return
    import queue
    import threading
    copy_stream = torch.cuda.Stream(device=dev)
    nslot = depth + 2                           # `depth` in flight + the one being consumed + the one being filled
    slots = [dict(pinned=None, device=None, copied=None, consumed=None) for _ in range(nslot)]
    ready = queue.Queue(maxsize=depth)          # bounds the read-ahead
    stop = threading.Event()

    def producer():
        try:
            torch.cuda.set_device(dev)
            # the staging memcpy runs on torch's intra-op threads; this thread's team is kept small (the default is one
            # thread per visible core, which under a container CPU quota starves the training thread)
            torch.set_num_threads(COPY_THREADS)
            for i, (feats, targets, ids, masks) in enumerate(loader):
                if stop.is_set():
                    return
                slot = slots[i % nslot]
                host = (feats, targets, masks)
                if slot["pinned"] is None or any(p.shape[1:] != h.shape[1:] or p.dtype != h.dtype or p.shape[0] < h.shape[0]
                                                 for p, h in zip(slot["pinned"], host)):
                    if slot["copied"] is not None:
                        slot["copied"].synchronize()
                    slot["pinned"] = [torch.empty(h.shape, dtype=h.dtype).pin_memory() for h in host]
                    with torch.cuda.stream(copy_stream):
                        slot["device"] = [torch.empty(h.shape, dtype=h.dtype, device=dev) for h in host]
                if slot["copied"] is not None:
                    slot["copied"].synchronize()            # the previous H2D copy has read this slot's pinned buffers
                pviews = [p[:h.shape[0]] for p, h in zip(slot["pinned"], host)]
                for v, h in zip(pviews, host):
                    v.copy_(h)                              # pageable -> pinned (skipped work if the loader pins: same cost)
                dviews = [d[:h.shape[0]] for d, h in zip(slot["device"], host)]
                with torch.cuda.stream(copy_stream):
                    if slot["consumed"] is not None:
                        copy_stream.wait_event(slot["consumed"])    # the step that used this slot's device buffers is done
                    for d, v in zip(dviews, pviews):
                        d.copy_(v, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                slot["copied"] = ev
                ready.put((i % nslot, dviews, ids, ev))
            ready.put(None)
        except BaseException as e:                         # surface loader / copy errors in the training thread
            ready.put(e)

    th = threading.Thread(target=producer, name="s2vt-feed", daemon=True)
    th.start()
    try:
        while True:
            item = ready.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            slot_ix, (f, t, m), ids, ev = item
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            yield f, t, ids, m
            # The caller is back for the next batch: everything that reads this batch is enqueued.  Publish "consumed" NOW,
            # before the ready.get() above frees a queue place: that get() is what lets the producer move on to the slot
            # this batch lives in, and it must find the event there (an event published after the get() could be missed).
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            slots[slot_ix]["consumed"] = done
    finally:
        stop.set()
        while th.is_alive():                               # unblock a producer waiting on the bounded queue
            try:
                ready.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=0.05)
