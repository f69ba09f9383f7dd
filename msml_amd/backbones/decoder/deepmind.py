"""Recover decoder of the reference (backbones/decoder/deepmind.py:20-118): 512 x 7 x 7 -> 3 x 112 x 112.

Same modules / parameter names (state dicts interchange).  In the reference its loss is silently
dropped and its output discarded (`_rec, l4 = self.decoder(x, ori) if ... else None, 0.` parses as a
tuple, backbones/frb/iresnet.py:228), so no parameter of it ever receives a gradient: the FRB
here does not run it (identical results, the dead compute is skipped).  `forward` is provided for direct
use; it runs on the HIP conv kernels without an autograd graph, matching that no gradient exists.
"""
import torch
from torch import nn

from ... import functional as Fh
from ..._lib import F32
from .._nn import conv

__all__ = ["dm_decoder", "DeepMindDecoder", "ResBlock"]


class ResBlock(nn.Module):
    def __init__(self, input_channels, channel):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(input_channels, channel, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(channel, input_channels, 1))

    def forward(self, x):
        out, _ = conv(self.conv[0], x)
        out, _ = conv(self.conv[2], Fh.relu_res(out))
        return Fh.relu_res(out, x)


class DeepMindDecoder(nn.Module):
    def __init__(self, n_init=32, n_hid=64, output_channels=3):
        super().__init__()
        def stage(cin):
            return [nn.Conv2d(cin, 2 * n_hid, 3, padding=1), nn.ReLU(), ResBlock(2 * n_hid, 2 * n_hid // 4),
                    ResBlock(2 * n_hid, 2 * n_hid // 4), nn.ConvTranspose2d(2 * n_hid, n_hid, 4, stride=2, padding=1),
                    nn.ReLU(inplace=True)]
        self.net = nn.Sequential(*(stage(n_init) + stage(n_hid) + stage(n_hid) +
                                   [nn.ConvTranspose2d(n_hid, output_channels, 4, stride=2, padding=1)]))
        self.output_channels = output_channels

    @torch.no_grad()
    def forward(self, x, ori=None):
        """x: NHWC storage tensor (B, 7, 7, 512) or NCHW f32; returns (recover NCHW f32, MSE vs ori or 0)."""
        from ... import ops
        if x.dim() == 4 and x.shape[1] == self.net[0].in_channels and x.shape[-1] != self.net[0].in_channels:
            x = Fh.to_nhwc(x.float(), F32)
        for m in self.net:
            if isinstance(m, nn.ReLU):
                x = Fh.relu_res(x)
            elif isinstance(m, ResBlock):
                x = m(x)
            else:
                x, _ = conv(m, x)
        rec = ops.to_nchw(x, self.output_channels)
        loss = torch.nn.functional.mse_loss(rec, ori) if ori is not None else 0.
        return rec, loss


def dm_decoder(pretrained=False, **kwargs):
    if pretrained:
        raise NotImplementedError("Pretrained model not support!")
    return DeepMindDecoder(n_hid=64, **kwargs)
