"""CPU, world_size 2 over gloo: the data-parallel step (batch sharding + flat-bucket gradient all-reduce of
s2vt_video_caption_amd.dp) reproduces the single-process global-batch step.  The model used here is the
oracle's (tests may use it as the stand-in replica: the HIP model needs a GPU); what is under test is the
sharding / reduction logic that bench.py runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import s2vt_oracle as orc
    from s2vt_video_caption_amd import dp, synth
    d = synth.CONFIGS["tiny"]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=5)
    feats, caps, mask = synth.make_batch(4, d["L"], d["F"], d["V"], seed=5)     # global batch of 4
    f, c, m = dp.shard_batch((feats, caps, mask), rank, world)
    model = orc.OracleModel(sd)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    red = dp.FlatGradAllReducer(model.parameters(), bucket_bytes=4096)          # several buckets
    assert len(red.buckets) > 1

    class Wrap(torch.nn.Module):        # same call signature as S2VT.forward(feats, targets=..., mode=...)
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, feats, targets=None, mode="train"):
            return self.inner(feats, targets)

    wrapped = Wrap(model)
    losses = []
    for _ in range(2):
        losses.append(float(dp.train_step(wrapped, orc.mask_criterion, opt, f, c, m, red)))
    torch.save({"losses": losses, "params": {k: v.detach() for k, v in model.as_dict().items()},
                "flat_norm": float(red.flat.norm())}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_dp_equals_global_batch(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import s2vt_oracle as orc
    from s2vt_video_caption_amd import synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    d = synth.CONFIGS["tiny"]
    sd = synth.make_state_dict(d["V"], d["F"], d["H"], d["E"], seed=5)
    feats, caps, mask = synth.make_batch(4, d["L"], d["F"], d["V"], seed=5)
    losses, _, final = orc.train_steps(sd, feats, caps, mask, 2)
    # replicas stay identical, and equal the single-process global-batch run (mean of shard means = global mean)
    for k in final:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
        assert (r0["params"][k] - final[k]).abs().max().item() < 2e-6, k
    assert abs(r0["flat_norm"] - r1["flat_norm"]) < 1e-6
    for s in range(2):
        assert abs(0.5 * (r0["losses"][s] + r1["losses"][s]) - losses[s]) < 2e-6


def _plateau_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from s2vt_video_caption_amd import dp
    # dp.global_mean: rank-local sums (different on every rank) reduced to ONE number that can drive ReduceLROnPlateau on
    # every rank (train.py itself now validates unsharded and broadcasts rank 0's loss: the single-process value)
    w = torch.nn.Parameter(torch.ones(3))
    opt = torch.optim.Adam([w], lr=1e-2)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=1)
    g = torch.Generator().manual_seed(100 + rank)
    lrs, vals = [], []
    for epoch in range(12):
        local = 1.0 + 0.3 * float(torch.rand(1, generator=g)) + (0.5 if epoch < 2 else 0.0)   # plateaus after epoch 2
        count = 3 + rank                                                                    # ragged shards
        v = dp.global_mean(local * count, count)
        sched.step(v)
        lrs.append(opt.param_groups[0]["lr"])
        vals.append(v)
    torch.save({"lrs": lrs, "vals": vals}, os.path.join(out_dir, "plateau%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_validation_loss_is_identical_on_all_ranks_so_lr_schedules_stay_in_step(tmp_path):
    world = 2
    mp.spawn(_plateau_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "plateau0.pt"), torch.load(tmp_path / "plateau1.pt")
    assert r0["vals"] == r1["vals"]                  # bitwise the same validation loss on both ranks
    assert r0["lrs"] == r1["lrs"]                    # so the plateau scheduler cuts the LR at the same epochs
    assert r0["lrs"][-1] < r0["lrs"][0]              # and it did cut it


def test_shard_rows_and_buckets():
    sys.path.insert(0, ROOT)
    from s2vt_video_caption_amd import dp
    assert dp.shard_rows(1024, 3, 8) == (384, 512)
    with pytest.raises(ValueError):
        dp.shard_rows(10, 0, 4)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 1000, 7, 3000)]
    red = dp.FlatGradAllReducer(ps, bucket_bytes=4000)
    assert red.flat.numel() == 4017
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == 4017
    assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:]))
    for p in ps:                                   # grads are views into the flat buffer
        assert p.grad.data_ptr() >= red.flat.data_ptr()
    ps[1].grad.fill_(2.0)
    assert float(red.flat.sum()) == 2000.0
    red.all_reduce()                               # world 1: no-op
    red.zero_grad()
    assert float(ps[1].grad.sum()) == 0.0
