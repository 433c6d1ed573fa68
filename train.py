"""Training harness for the MI355X S2VT path: the reference's loop (train.py:56-175) —
Adam(lr) + ReduceLROnPlateau(patience) + EarlyStopping(patience) + full-module checkpoints every `save_freq` epochs, at
the best validation loss and at the end — on top of the drop-in `S2VTModel.S2VT`, `utils.MaskCriterion`,
`dataloader.VideoDataset` and, with more than one process, data parallelism over RCCL (`s2vt_video_caption_amd.dp`).

  python train.py --caption-file data/captions.json --feats-path data/feats/vgg16_bn
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...

Differences from the reference script: it trains `S2VT` (the reference's committed script instantiates the unrelated
`Att_Baseline`, train.py:86), configuration comes from argparse with the reference's `Opt` defaults (train.py:20-48),
TensorBoard logging is optional (tensorboardX is not a dependency), and `ReduceLROnPlateau` is built without the
`verbose` argument that torch >= 2.6 removed.
"""
import argparse
import os
import time

import torch
import torch.distributed as dist
from torch import optim


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--caption-file", default="./data/captions_server.json")
    ap.add_argument("--feats-path", default="./data/feats/vgg16_bn")
    ap.add_argument("--train-length", type=int, default=80)
    ap.add_argument("--dim-hidden", type=int, default=512)
    ap.add_argument("--dim-embed", type=int, default=512)
    ap.add_argument("--feat-dim", type=int, default=4096)
    ap.add_argument("--feat-dropout", type=float, default=0.0)
    ap.add_argument("--out-dropout", type=float, default=0.0)
    ap.add_argument("--rnn-dropout", type=float, default=0.0, help="accepted as upstream; no effect with one LSTM layer")
    ap.add_argument("--batch-size", type=int, default=16, help="per process")
    ap.add_argument("--workers", type=int, default=0,
                    help="DataLoader worker processes (0 = the reference's behaviour: items are loaded by the feed thread; "
                         "at B=64 a batch is 64 .npy files = 84 MB per 13-ms step, so real runs want a few workers)")
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--save-freq", type=int, default=100)
    ap.add_argument("--save-path", default="./checkpoint")
    ap.add_argument("--early-stopping-patience", type=int, default=30)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--learning-rate-patience", type=int, default=20)
    # reproducibility switches (not in the reference, whose runs are unseeded): used by the harness parity test
    ap.add_argument("--no-shuffle", action="store_true", help="iterate the training set in file order")
    ap.add_argument("--seed", type=int, default=None, help="seed numpy's global RNG (caption sampling, dataloader.py:41)")
    ap.add_argument("--model", choices=("s2vt", "att_baseline"), default="s2vt",
                    help="s2vt: S2VTModel.S2VT (the hot path); att_baseline: attention_baseline.Att_Baseline, the network the "
                         "reference's committed train.py:86 instantiates")
    ap.add_argument("--init-state", default=None, help="state_dict file to start from instead of the seeded default init")
    return ap.parse_args(argv)


def run(opt):
    """The reference's training loop (train.py:108-168).  Returns the history it went through:
    {"train_loss": [...], "valid_loss": [...], "lr": [lr in force DURING each epoch], "stopped_at": epoch or None,
     "checkpoints": [file names in the order they were written]}."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import s2vt_video_caption_amd  # noqa: F401
        from s2vt_video_caption_amd import dp as _dp
        _dp.plan_for_collectives(world)      # RCCL's channel count and the GEMMs' compute-unit reserve, before the communicator exists
        dist.init_process_group(backend="nccl", device_id=dev)

    import numpy as np
    import dataloader
    from S2VTModel import S2VT
    from utils import EarlyStopping, MaskCriterion
    from s2vt_video_caption_amd import capi, dp

    if opt.seed is not None:
        np.random.seed(opt.seed + rank)
    start_time = time.strftime('%y_%m_%d_%H_%M_%S-', time.localtime())
    os.makedirs(opt.save_path, exist_ok=True)
    if rank == 0:                                           # the run's configuration beside its checkpoints (save_opt, train.py:51-53)
        with open(os.path.join(opt.save_path, start_time + 'opt.txt'), 'w+', encoding='utf-8') as f:
            f.write(str(vars(opt)))
    trainset = dataloader.VideoDataset(opt.caption_file, opt.feats_path, max_len=opt.train_length)
    validset = dataloader.VideoDataset(opt.caption_file, opt.feats_path, max_len=opt.train_length, mode='valid')
    shuffle = not opt.no_shuffle
    sampler = torch.utils.data.distributed.DistributedSampler(trainset, shuffle=shuffle, drop_last=True) if world > 1 else None
    train_loader = torch.utils.data.DataLoader(trainset, batch_size=opt.batch_size, shuffle=shuffle and sampler is None,
                                               sampler=sampler, drop_last=world > 1, num_workers=opt.workers,
                                               persistent_workers=opt.workers > 0)
    # Validation is NOT sharded: every rank runs the whole split in the reference's batch composition (train.py:136-146), so
    # the number that drives ReduceLROnPlateau / EarlyStopping is the single-process one whatever the world size.  (A
    # DistributedSampler pads the split with duplicated samples and a mean of per-rank batch means weights the ragged last
    # batches differently: LR cuts and the stop epoch could then differ from a 1-GPU run.)  The split is small (MSVD: 100
    # videos); rank 0's value is broadcast so that every replica steps its schedulers on the same bits.
    valid_loader = torch.utils.data.DataLoader(validset, batch_size=opt.batch_size, shuffle=False,
                                               num_workers=opt.workers, persistent_workers=opt.workers > 0)
    word2ix = trainset.word2ix

    torch.manual_seed(0)        # identical replicas
    if opt.model == "att_baseline":
        from attention_baseline import Att_Baseline
        model = Att_Baseline(len(word2ix), opt.feat_dim, length=opt.train_length, dim_hid=opt.dim_hidden, dim_embed=opt.dim_embed,
                             feat_dropout=opt.feat_dropout, out_dropout=opt.out_dropout,
                             sos_ix=word2ix['<sos>'], eos_ix=word2ix['<eos>'])                      # train.py:86-87 upstream
    else:
        model = S2VT(len(word2ix), opt.feat_dim, length=opt.train_length, dim_hid=opt.dim_hidden, dim_embed=opt.dim_embed,
                     feat_dropout=opt.feat_dropout, rnn_dropout=opt.rnn_dropout, out_dropout=opt.out_dropout,
                     sos_ix=word2ix['<sos>'], eos_ix=word2ix['<eos>'])
    if opt.init_state:
        model.load_state_dict(torch.load(opt.init_state))
    model.to(dev)
    reducer = None
    if world > 1:       # S2VT: gradients written straight into the flat buffer, all-reduce overlapped with the backward;
        reducer = dp.FlatGradAllReducer(model.parameters())            # Att_Baseline: plain bucketed all-reduce after it
        if opt.model == "s2vt":
            reducer.attach(model)
    if opt.model == "s2vt":     # train.py:89-93's Adam, same arithmetic, as one launch over flat parameter / gradient / moment buffers
        from s2vt_video_caption_amd.optim import FlatAdam
        optimizer = FlatAdam(model, lr=opt.lr, reducer=reducer)
    else:
        optimizer = optim.Adam(model.parameters(), lr=opt.lr)                                       # train.py:89-93
    lr_scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=opt.learning_rate_patience)   # :95-97
    early_stopping = EarlyStopping(patience=opt.early_stopping_patience, verbose=rank == 0,
                                   path=os.path.join(opt.save_path, start_time + 'stop.pth'))        # :98-100
    criterion = MaskCriterion()
    hist = {"train_loss": [], "valid_loss": [], "lr": [], "stopped_at": None, "checkpoints": []}

    def save(name):
        if rank == 0:
            torch.save(model, os.path.join(opt.save_path, start_time + name))
            hist["checkpoints"].append(name)

    for epoch in range(opt.epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        hist["lr"].append(optimizer.param_groups[0]['lr'])
        running, count = 0.0, 0
        for feats, targets, ids, masks in dataloader.feed_batches(train_loader, dev):
            # train.py:116-127; check_errors: a device-side error of this step (IndexError for a caption id outside the vocabulary)
            # is raised BEFORE optimizer.step(), as in the reference, whose nn.Embedding raises in the forward
            loss = dp.train_step(model, criterion, optimizer, feats, targets, masks, reducer, check_errors=True)
            running += float(loss)
            count += 1
        train_loss = running / max(count, 1)
        running, count = 0.0, 0
        model.eval()
        with torch.no_grad():
            for feats, targets, ids, masks in dataloader.feed_batches(valid_loader, dev):
                probs = model(feats, targets=targets[:, :-1], mode='train')                    # train.py:141-143
                running += float(criterion(probs, targets, masks))
                count += 1
        valid_loss = running / max(count, 1)                                                   # train.py:147
        if world > 1:
            vl = torch.tensor([valid_loss], dtype=torch.float64, device=dev)
            dist.broadcast(vl, 0)
            valid_loss = float(vl[0])
        hist["train_loss"].append(train_loss)
        hist["valid_loss"].append(valid_loss)
        if rank == 0:
            print("epoch {} train loss:{} valid loss: {} lr: {}".format(epoch, train_loss, valid_loss,
                                                                        optimizer.param_groups[0]['lr']))
        lr_scheduler.step(valid_loss)                                                          # train.py:155
        n_before = early_stopping.val_loss_min
        if rank == 0:
            early_stopping(valid_loss, model)                                                  # train.py:158
            if early_stopping.val_loss_min != n_before:
                hist["checkpoints"].append('stop.pth')
        stop = torch.tensor([1 if (rank == 0 and early_stopping.early_stop) else 0], device=dev)
        if world > 1:
            dist.broadcast(stop, 0)
        if int(stop):                                                                          # train.py:159-161
            if rank == 0:
                print("Early stopping")
            hist["stopped_at"] = epoch
            break
        if epoch % opt.save_freq == 0:                                                         # train.py:164-167
            save(str(epoch) + '.pth')
    save('final.pth')                                                                          # train.py:175
    hist["start_time"] = start_time
    if world > 1:
        dist.destroy_process_group()
    return hist


def main():
    run(parse())


if __name__ == '__main__':
    main()
