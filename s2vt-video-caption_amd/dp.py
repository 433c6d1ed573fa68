"""Data-parallel training of the S2VT step: one process per GPU, gradients all-reduced with RCCL.

The reference has no multi-device mode; BASELINE.json asks for a data-parallel step
over the 8 GPUs of a node.  The path shards over the batch only (samples never interact except through
the mean in utils.py:22), so rank r takes rows [r*B/W, (r+1)*B/W) of the global batch, every rank holds
a full replica, and the only collective is one all-reduce (sum, then 1/W) of the fp32 gradients.

xGMI is point-to-point (7 links/GPU): the 13 gradient tensors are kept in ONE flat fp32 buffer
(48.1 M floats = 192.5 MB at H=E=1000, V=12000) so the all-reduce runs as a few large buckets instead of
13 small calls, and each bucket is issued asynchronously in reverse order of the backward
(out_linear/embedding first).  `torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests.
"""
import torch
import torch.distributed as dist


def shard_rows(n_rows, rank, world):
    """Row range of the global batch owned by `rank` (equal shards; n_rows must divide)."""
    if n_rows % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_rows, world))
    per = n_rows // world
    return rank * per, (rank + 1) * per


def shard_batch(tensors, rank, world):
    lo, hi = shard_rows(tensors[0].shape[0], rank, world)
    return tuple(t[lo:hi] for t in tensors)


class FlatGradAllReducer:
    """Flat gradient buffer + bucketed all-reduce for a replica.

    `param.grad` of every parameter becomes a view into one flat buffer (autograd accumulates into an
    existing `.grad` in place), so after `backward()` the buffer holds all gradients with no copies.
    Use `zero_grad()` of this object (or `optimizer.zero_grad(set_to_none=False)`) to keep the views.
    """

    def __init__(self, params, process_group=None, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        self.slices = []
        for p in self.params:
            k = p.numel()
            p.grad = self.flat[off:off + k].view_as(p)
            self.slices.append((off, off + k))
            off += k
        # buckets: contiguous ranges of the flat buffer, cut at parameter boundaries, issued last-first
        self.buckets = []
        per = max(1, bucket_bytes // self.flat.element_size())
        start = 0
        for (lo, hi) in self.slices:
            if hi - start >= per:
                self.buckets.append((start, hi))
                start = hi
        if start < n:
            self.buckets.append((start, n))

    def zero_grad(self):
        self.flat.zero_()

    def all_reduce(self):
        """Average gradients over ranks (sum all-reduce, then * 1/W). No-op for a single process."""
        if self.world == 1 and not dist.is_initialized():
            return
        works = [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for (lo, hi) in reversed(self.buckets)]
        for w in works:
            w.wait()
        self.flat.mul_(1.0 / self.world)


def train_step(model, criterion, optimizer, feats, caps, mask, reducer=None):
    """One optimisation step of train.py:116-127 on this rank's shard; returns the (local) loss tensor.
    With `reducer`, gradients are averaged over ranks before the optimiser step."""
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad()
    model.train()
    probs = model(feats, targets=caps[:, :-1], mode='train')
    loss = criterion(probs, caps, mask)
    loss.backward()
    if reducer is not None:
        reducer.all_reduce()
    optimizer.step()
    return loss.detach()
