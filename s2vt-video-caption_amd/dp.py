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


def plan_for_collectives(world, channels=16):
    """Call BEFORE the RCCL process group is created when world > 1.  The persistent GEMMs of the backward plan one workgroup per
    compute unit with a static share of the tiles; RCCL's all-reduce kernels (one workgroup per channel) run beside the
    weight-gradient GEMMs and would make the workgroups that find no unit wait for a whole share.  So the pair is made
    deterministic: RCCL is held to `channels` channels (NCCL_MAX_NCHANNELS, unless the user set it) and the GEMMs plan their grids
    for that many compute units fewer (library option cu_reserve, unless the user set it: S2VT_CU_RESERVE or s2vt_set_option).
    Returns the number of compute units reserved (0 at world 1)."""
    import os
    from . import capi
    if world <= 1:
        return 0
    n = int(os.environ.setdefault("NCCL_MAX_NCHANNELS", str(int(channels))))
    lib = capi.load()
    if "S2VT_CU_RESERVE" not in os.environ and lib.s2vt_set_option(b"cu_reserve", -1) == 0:
        lib.s2vt_set_option(b"cu_reserve", max(0, min(n, 128)))
    return int(lib.s2vt_set_option(b"cu_reserve", -1))


def fixed_global_shard(global_batch, rank, world):
    """Strong-scaling companion of the weak-scaling headline (keep per-GPU B fixed AND report fixed global B):
    rank r of `world` trains rows [r * G / W, (r + 1) * G / W) of a global batch of G; returns (per_gpu_batch, lo, hi)."""
    lo, hi = shard_rows(global_batch, rank, world)
    return hi - lo, lo, hi


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

        self.model = None
        self.groups = None
        self.comm_stream = None
        self.timing = False          # True: every all_reduce() brackets its collectives with events (read_timing())
        self._timed = []

    def attach(self, model):
        """Overlap mode for the HIP S2VT replica `model` (whose 13 parameters are exactly self.params):
        * the backward WRITES its gradients straight into the flat buffer (no autograd accumulation pass, no zeroing);
        * the all-reduce is issued per gradient group as soon as the library says the group is final
          (s2vt_backward_wait_grads): out_linear ~1 ms into the backward, word_rnn + embedding before the
          vid_rnn / feat_linear weight-gradient GEMMs, the rest at the end - on a side stream, under the backward."""
        from . import functional
        hip = model._hip_params()
        by_id = {id(p): i for i, p in enumerate(self.params)}
        if len(hip) != len(self.params) or any(id(p) not in by_id for p in hip):
            raise ValueError("attach(model): the reducer must have been built from exactly this model's parameters")
        functional.set_grad_sink(model, [self.params[by_id[id(p)]].grad for p in hip])
        self.model = model
        names = {id(p): n for n, p in model.named_parameters()}
        group_of = lambda n: 0 if n.startswith("out_linear") else 1 if n.startswith(("word_rnn", "embedding")) else 2
        groups = {0: [], 1: [], 2: []}
        for p, (lo, hi) in zip(self.params, self.slices):
            g = groups[group_of(names[id(p)])]
            if g and g[-1][1] == lo:
                g[-1] = (g[-1][0], hi)          # merge neighbours into one contiguous range
            else:
                g.append((lo, hi))
        self.groups = groups
        if self.flat.is_cuda:
            self.comm_stream = torch.cuda.Stream(device=self.flat.device)
        return self

    def zero_grad(self):
        if self.model is None:                  # attached: the backward overwrites every gradient
            self.flat.zero_()

    def all_reduce(self):
        """Average gradients over ranks (sum all-reduce, then * 1/W). No-op for a single process."""
        if self.world == 1 and not dist.is_initialized():
            return
        if self.model is not None and self.flat.is_cuda:
            return self._all_reduce_overlapped()
        import time
        t0 = time.perf_counter()
        works = [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for (lo, hi) in reversed(self.buckets)]
        for w in works:
            w.wait()
        self.flat.mul_(1.0 / self.world)
        if self.timing:          # bucketed path (CPU / gloo, or a replica that is not attached): host wall time, nothing overlaps
            if self.flat.is_cuda:
                torch.cuda.synchronize(self.flat.device)
            ms = (time.perf_counter() - t0) * 1e3
            self._timed.append({"host_ms": ms, "bytes": [self.flat.numel() * self.flat.element_size()]})

    def _all_reduce_overlapped(self):
        """Called right after loss.backward() returned, i.e. with the whole backward ENQUEUED but mostly not executed."""
        from . import capi
        lib = capi.load()
        main = torch.cuda.current_stream(self.flat.device)
        cs = self.comm_stream

        def reduce(lo, hi):          # SUM, then 1/W on the communication stream (0.06 ms for the whole 192-MB buffer):
            t = self.flat[lo:hi]     # one fixed arithmetic on every backend, no probing of ReduceOp.AVG support
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.mul_(1.0 / self.world)

        ev = None
        if self.timing:              # diagnosis of a multi-GPU run: when did each group's all-reduce start and end, when did the
            mk = lambda: torch.cuda.Event(enable_timing=True)     # backward end (all on the device's clock)
            ev = {"bwd_end": mk(), "g": [(mk(), mk()) for _ in range(3)]}
            ev["bwd_end"].record(main)
        with torch.cuda.stream(cs):
            for g in (0, 1):
                capi.check(lib.s2vt_backward_wait_grads(g, capi.c_void_p(cs.cuda_stream)), "s2vt_backward_wait_grads")
                if ev:
                    ev["g"][g][0].record(cs)
                for lo, hi in self.groups[g]:
                    reduce(lo, hi)
                if ev:
                    ev["g"][g][1].record(cs)
            cs.wait_stream(main)                 # the remaining gradients are final with the backward's stream
            if ev:
                ev["g"][2][0].record(cs)
            for lo, hi in self.groups[2]:
                reduce(lo, hi)
            if ev:
                ev["g"][2][1].record(cs)
        main.wait_stream(cs)
        if ev:
            self._timed.append(ev)

    def read_timing(self):
        """Mean over the all_reduce() calls made with `timing` on (and forget them): per gradient group (0 out_linear, 1 word_rnn +
        embedding, 2 vid_rnn + feat_linear) the duration of its all-reduce + 1/W scaling on the communication stream, its start
        relative to the end of the backward (negative = issued under the backward), and `exposed_after_backward_ms` = from the
        backward's last kernel to the last collective's end - the part of the communication the step waits for."""
        recs, self._timed = self._timed, []
        if not recs:
            return None
        if "host_ms" in recs[0]:
            return {"path": "bucketed (not overlapped)", "calls": len(recs), "host_ms": sum(r["host_ms"] for r in recs) / len(recs),
                    "bytes": recs[0]["bytes"], "world": self.world}
        torch.cuda.synchronize(self.flat.device)
        n = len(recs)
        esz = self.flat.element_size()
        grp = [sum(r["g"][g][0].elapsed_time(r["g"][g][1]) for r in recs) / n for g in range(3)]
        start = [sum(r["bwd_end"].elapsed_time(r["g"][g][0]) for r in recs) / n for g in range(3)]
        exposed = sum(max(0.0, r["bwd_end"].elapsed_time(r["g"][2][1])) for r in recs) / n
        return {"path": "overlapped with the backward (s2vt_backward_wait_grads)", "calls": n, "world": self.world,
                "group_ms": [round(x, 4) for x in grp], "group_start_after_backward_end_ms": [round(x, 4) for x in start],
                "exposed_after_backward_ms": round(exposed, 4),
                "bytes": [sum(hi - lo for lo, hi in self.groups[g]) * esz for g in range(3)]}


def global_mean(total, count, device="cpu"):
    """sum(total over ranks) / sum(count over ranks): the SAME number on every rank.  train.py feeds it to
    ReduceLROnPlateau / EarlyStopping: replicas that stepped their schedulers on rank-local validation losses would cut
    the learning rate at different epochs and drift apart while their gradients are still being averaged."""
    t = torch.tensor([float(total), float(count)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0] / torch.clamp(t[1], min=1.0))


def train_step(model, criterion, optimizer, feats, caps, mask, reducer=None, check_errors=False):
    """One optimisation step of train.py:116-127 on this rank's shard; returns the (local) loss tensor.
    With `reducer`, gradients are averaged over ranks before the optimiser step.
    `check_errors`: synchronise and raise device-side errors of this step (a caption id outside the vocabulary -> IndexError, as
    nn.Embedding raises in the reference's forward; a timed-out hand-off) BEFORE optimizer.step(), so that a bad batch never
    reaches the weights - what the reference's ordering gives for free.  Costs one synchronisation per step; a loop that reads
    loss.item() every step (train.py:127) pays that anyway."""
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
    if check_errors:
        from . import capi
        torch.cuda.synchronize(loss.device)
        err = None
        try:
            capi.check_async_error()
        except Exception as e:                  # noqa: BLE001 - re-raised below, on every rank
            err = e
        if reducer is not None and reducer.world > 1:
            # every rank learns whether ANY rank failed before one of them raises: a rank that raised alone would leave the
            # others blocked in the next step's all-reduce until the process-group time-out
            flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=loss.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=reducer.group)
            if err is None and int(flag[0]) != 0:
                err = capi.S2VTHipError("another rank reported a device-side error in this step (bad caption id or a timed-out "
                                        "hand-off): stopping with it")
        if err is not None:
            raise err
    optimizer.step()
    return loss.detach()
