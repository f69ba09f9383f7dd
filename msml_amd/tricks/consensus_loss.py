"""Segmentation loss on the HIP path (reference: tricks/consensus_loss.py:28-179).

blobs == target as the reference trains (train.py:228-229,258); both reductions of each term ('idx': per blob /
per non-zero entry, the defaults train.py uses; 'all': H * W / N * H * W, consensus_loss.py:127-133,159-162)."""
import os

import torch
import torch.nn as nn

from .. import functional as Fh


class StructureConsensuLossFunction(nn.Module):
    def __init__(self, consensus_loss_alpha=10.0, consensus_loss_beta=5.0, reduce_pixel="idx",
                 reduce_pixel_kl="idx"):
        super().__init__()
        self.reduce_pixel, self.reduce_pixel_kl = reduce_pixel, reduce_pixel_kl
        self.consensus_loss_alpha = consensus_loss_alpha
        self.consensus_loss_beta = consensus_loss_beta

        self._checked = False

    def forward(self, logit, blobs, target):
        # train.py:256-258 passes blobs = msk.clone(): equal by construction but never the same object.  The FIRST call
        # compares them on the host and raises (a full pass + a host synchronisation, which would stall the host that
        # the eager multi-stream issue relies on running ahead if it were paid per step; MSML_DEBUG_SEG_BLOBS=1: every
        # call).  EVERY later call still compares them, on the device and without a synchronisation: a mismatch turns
        # the returned loss into NaN (and with it every gradient of the step), so no caller can train on a silently
        # wrong `blobs` after call 1; the comparison is skipped when the same tensor is passed twice.
        poison = None
        if blobs is not target:
            if not self._checked or os.environ.get("MSML_DEBUG_SEG_BLOBS"):
                self._checked = True
                if blobs.shape != target.shape or not bool((blobs == target).all()):
                    raise NotImplementedError("msml_amd: blobs must equal target (as in train.py:258)")
            elif blobs.shape != target.shape:
                raise NotImplementedError("msml_amd: blobs must equal target (as in train.py:258)")
            else:
                bad = (blobs.to(target.device) != target).any()
                poison = torch.where(bad, torch.full((), float("nan"), device=logit.device),
                                     torch.ones((), device=logit.device))
        loss = Fh.seg_consensus_loss(logit, target, float(self.consensus_loss_alpha),
                                     float(self.consensus_loss_beta), self.reduce_pixel, self.reduce_pixel_kl)
        return loss if poison is None else loss * poison        # (x NaN: the loss AND its gradient)
