"""CPU restatement of the reference's epoch loop.  TEST INFRASTRUCTURE ONLY (same rules as s2vt_oracle.py).

`run_epochs` follows train.py:104-175 of the reference step by step - Adam(lr) (train.py:89-93), per epoch a pass over
the training batches (:111-128) and one over the validation batches (:135-147), `ReduceLROnPlateau(patience)` stepped on
the mean validation loss (:95-97, :155), `EarlyStopping(patience)` (utils.py:29-80) which "saves" the full module at
every improvement and stops after `patience` epochs without one (:158-161), a periodic checkpoint every `save_freq`
epochs that is skipped in the epoch that stops (:164-167), and the final checkpoint (:175) - on the oracle's explicit-cell
model.  No file is written: checkpoint events are returned by name, in order.

Parity pinning: `EarlyStoppingOracle` is checked against the reference's own class (utils.py, importable once
`np.Inf` is aliased) by oracle/make_golden.py --check-early-stopping in the build container; ReduceLROnPlateau is
torch's own (third party, as in the reference).
"""
import torch

from . import s2vt_oracle as orc


class EarlyStoppingOracle:
    """utils.py:29-80: best_score = -val_loss of the best epoch; an epoch with score < best_score + delta increments the
    counter (stop at `patience`), any other epoch is an improvement: checkpoint + counter reset."""

    def __init__(self, patience=7, delta=0.0):
        self.patience, self.delta = patience, delta
        self.counter, self.best_score, self.early_stop = 0, None, False
        self.saves = 0

    def __call__(self, val_loss):
        score = -val_loss                                   # utils.py:60
        if self.best_score is None:                         # :62-64
            self.best_score = score
            self.saves += 1
            return True
        if score < self.best_score + self.delta:            # :65-69
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
            return False
        self.best_score = score                             # :70-73
        self.saves += 1
        self.counter = 0
        return True


def run_epochs(state_dict, train_batches, valid_batches, epochs, lr=1e-4, lr_patience=20, es_patience=30, save_freq=100):
    """train_batches / valid_batches: callables returning, per epoch, an iterable of (feats, caps, mask) CPU tensors (the
    reference draws a fresh random caption per item and epoch, dataloader.py:41).  Returns the same history dict as the
    product's train.run()."""
    model = orc.OracleModel(state_dict)
    opt = torch.optim.Adam(model.parameters(), lr=lr)                                   # train.py:89-93
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=lr_patience)       # train.py:95-97
    es = EarlyStoppingOracle(patience=es_patience)                                      # train.py:98-100
    hist = {"train_loss": [], "valid_loss": [], "lr": [], "stopped_at": None, "checkpoints": []}
    for epoch in range(epochs):                                                         # train.py:108
        hist["lr"].append(opt.param_groups[0]["lr"])
        run, n = 0.0, 0
        for feats, caps, mask in train_batches(epoch):                                  # :114-128
            opt.zero_grad()
            loss = orc.mask_criterion(model(feats, caps[:, :-1]), caps, mask)
            loss.backward()
            opt.step()
            run += float(loss.detach())
            n += 1
        hist["train_loss"].append(run / n)
        run, n = 0.0, 0
        with torch.no_grad():
            for feats, caps, mask in valid_batches(epoch):                              # :137-147
                run += float(orc.mask_criterion(model(feats, caps[:, :-1]), caps, mask))
                n += 1
        valid = run / n
        hist["valid_loss"].append(valid)
        sched.step(valid)                                                               # :155
        if es(valid):                                                                   # :158
            hist["checkpoints"].append("stop.pth")
        if es.early_stop:                                                               # :159-161
            hist["stopped_at"] = epoch
            break
        if epoch % save_freq == 0:                                                      # :164-167
            hist["checkpoints"].append(str(epoch) + ".pth")
    hist["checkpoints"].append("final.pth")                                             # :175
    hist["final_state"] = {k: v.detach().clone() for k, v in model.as_dict().items()}
    return hist
