"""Drop-in `utils` module: `MaskCriterion` with the reference's exact semantics (utils.py:6-26), the
cross-entropy itself computed by the HIP kernels behind `s2vt_mean_ce_forward/backward`."""
import torch
import torch.nn as nn

import s2vt_video_caption_amd  # noqa: F401
from s2vt_video_caption_amd import functional as _F


class MaskCriterion(nn.Module):
    """calculate the CrossEntropyLoss in mask=1 area (as in the reference the inner loss is already the
    mean over all B*(L-1) positions, so the mask cancels; NaN if the mask is all zero)"""

    def __init__(self):
        super(MaskCriterion, self).__init__()

    def forward(self, logits, target, mask):
        """
        logits: (N, seq_len - 1, vocab_size); target: (N, seq_len); mask: (N, seq_len)
        """
        loss = _F.mean_cross_entropy(logits, target)              # utils.py:22 (mean over N*(seq_len-1))
        mask = mask[:, 1:]
        mask_loss = loss * mask.contiguous().view(-1)             # utils.py:24
        return torch.sum(mask_loss) / torch.sum(mask)             # utils.py:25


class EarlyStopping:
    """Stop when the validation loss has not improved for `patience` calls (same constructor, attributes and
    call protocol as the reference's `utils.EarlyStopping`, utils.py:29-80; `np.Inf` there breaks on NumPy 2).
    Every improvement saves the FULL module with `torch.save(model, path)` — the reference's checkpoint format."""

    def __init__(self, patience=7, verbose=False, delta=0, path='checkpoint.pt', trace_func=print):
        self.patience, self.verbose, self.delta, self.path, self.trace_func = patience, verbose, delta, path, trace_func
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = float('inf')

    def __call__(self, val_loss, model):
        score = -val_loss
        improved = self.best_score is None or not (score < self.best_score + self.delta)
        if improved:
            self.best_score = score
            self.save_checkpoint(val_loss, model)
            self.counter = 0
            return
        self.counter += 1
        self.trace_func('EarlyStopping counter: {} out of {}'.format(self.counter, self.patience))
        if self.counter >= self.patience:
            self.early_stop = True

    def save_checkpoint(self, val_loss, model):
        if self.verbose:
            self.trace_func('Validation loss decreased ({:.6f} --> {:.6f}).  Saving model ...'.format(
                self.val_loss_min, val_loss))
        torch.save(model, self.path)
        self.val_loss_min = val_loss
