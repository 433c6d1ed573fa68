"""Drop-in `utils` module for the MI355X S2VT path.

`MaskCriterion` keeps the reference's semantics (utils.py:6-26) with the cross-entropy itself computed by the HIP
kernels behind `s2vt_mean_ce_forward/backward`; `EarlyStopping` keeps the reference's constructor, attributes and
call protocol (utils.py:29-80)."""
import torch
import torch.nn as nn

import s2vt_video_caption_amd  # noqa: F401
from s2vt_video_caption_amd import functional as _F


class MaskCriterion(nn.Module):
    """Masked caption loss.  As upstream, the inner loss is ALREADY the mean cross-entropy over all N*(seq_len-1)
    positions (utils.py:22), so weighting it by the mask and dividing by the mask's sum returns that mean again
    (and NaN for an all-zero mask); the arithmetic is reproduced step by step."""

    def forward(self, logits, target, mask):
        # logits [N, seq_len-1, V]; target, mask [N, seq_len]
        if getattr(mask, "requires_grad", False):               # (never in the reference: the mask comes from the data loader)
            mean_ce = _F.mean_cross_entropy(logits, target)
            weights = mask[:, 1:].reshape(-1)                   # utils.py:23-24
            return (mean_ce * weights).sum() / weights.sum()    # utils.py:24-25
        # HIP kernels: CE against target[:, 1:], its mean, (mean * w).sum() / w.sum() with w = mask[:, 1:] - two launches
        return _F.mask_criterion(logits, target, mask)


class EarlyStopping:
    """Patience counter on the validation loss; every improvement writes the FULL module with `torch.save(model, path)`
    (the reference's checkpoint format).  Same public surface as upstream: `patience, verbose, delta, path, trace_func`,
    state in `counter, best_score, early_stop, val_loss_min`.  (`np.Inf` upstream no longer exists in NumPy 2.)"""

    def __init__(self, patience=7, verbose=False, delta=0, path='checkpoint.pt', trace_func=print):
        self.patience, self.verbose, self.delta = patience, verbose, delta
        self.path, self.trace_func = path, trace_func
        self.counter, self.best_score, self.early_stop = 0, None, False
        self.val_loss_min = float('inf')

    def __call__(self, val_loss, model):
        score = -val_loss
        if self.best_score is not None and score < self.best_score + self.delta:
            # no improvement beyond delta
            self.counter += 1
            self.trace_func('EarlyStopping counter: {} out of {}'.format(self.counter, self.patience))
            self.early_stop = self.early_stop or self.counter >= self.patience
            return
        self.best_score, self.counter = score, 0
        self.save_checkpoint(val_loss, model)

    def save_checkpoint(self, val_loss, model):
        if self.verbose:
            self.trace_func('Validation loss decreased ({:.6f} --> {:.6f}).  Saving model ...'.format(
                self.val_loss_min, val_loss))
        torch.save(model, self.path)
        self.val_loss_min = val_loss
