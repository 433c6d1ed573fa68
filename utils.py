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
