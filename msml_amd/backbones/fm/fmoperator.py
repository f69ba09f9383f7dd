"""Feature-Masking operators on the HIP path (reference: backbones/fm/fmoperator.py).

FMCnn.forward (:277-311): x = same_conv(cat(yf, yo)) -> res_block -> M = act(x);
z = arith(yf, M) + yf.  The concat is never materialised (two-segment implicit GEMM) and the
activation + arithmetic + skip are one fused kernel (msml_fm_fuse_fwd/bwd)."""
import torch
import torch.nn as nn

from ... import blocks, ops
from ... import functional as Fh
from .._nn import conv, conv_bn

__all__ = ["FMCnn", "FMNone"]


class resblock_bottle(nn.Module):
    """1x1 -> bn -> prelu -> 3x3 -> bn -> prelu -> 1x1 -> bn -> (+x) -> prelu  (:35-68)."""

    def __init__(self, in_channels, out_channels, bottle_channels=128):
        super().__init__()
        if in_channels <= 128:
            bottle_channels = in_channels // 2
        self.conv1 = nn.Conv2d(in_channels, bottle_channels, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(bottle_channels, eps=1e-05)
        self.prelu1 = nn.PReLU(bottle_channels)
        self.conv2 = nn.Conv2d(bottle_channels, bottle_channels, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(bottle_channels, eps=1e-05)
        self.prelu2 = nn.PReLU(bottle_channels)
        self.conv3 = nn.Conv2d(bottle_channels, out_channels, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out_channels, eps=1e-05)
        self.prelu3 = nn.PReLU(out_channels)

    def forward(self, x):
        if self.training and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and ops.BLOCK_FUNCTION \
                and ops.BOTTLE_FUNCTION:
            return blocks.bottleneck(self, x)      # one autograd node, fused backward (blocks.py)
        out = conv_bn(self.conv1, self.bn1, x, prelu=self.prelu1)
        out = conv_bn(self.conv2, self.bn2, out, prelu=self.prelu2)
        # prelu3(bn3(conv3(out)) + identity)
        return conv_bn(self.conv3, self.bn3, out, prelu=self.prelu3, residual=x, res_first=True)


class FMCnn(nn.Module):
    def __init__(self, height, width, channel_f, kernel_size=3, resblocks=2, activation="tanh",
                 arith_strategy="add", peer_params: dict = None):
        super().__init__()
        peer_params = peer_params or {}
        if peer_params.get("use_ori"):
            raise NotImplementedError("msml_amd: peer-guided FM branch (use_ori) is not built yet")
        self.height, self.width, self.channel_f = height, width, channel_f
        if kernel_size == 1:
            self.same_conv = nn.Conv2d(18 + channel_f, channel_f, 1, bias=False)
        else:
            self.same_conv = nn.Conv2d(18 + channel_f, channel_f, 3, 1, 1, bias=False)
        self.res_block = nn.Sequential(*[resblock_bottle(channel_f, channel_f)
                                         for _ in range(resblocks)])
        if activation not in Fh.ACTS or arith_strategy not in Fh.ARITHS:
            raise KeyError((activation, arith_strategy))
        self.activation = activation
        self.arith_strategy = arith_strategy
        self.use_ori = False
        self.conv1 = nn.Sequential()
        self.conv2 = nn.Sequential()
        self.conv_m = nn.Sequential()
        self.en_save = False

    def forward(self, yf, yo, yt=None):
        x, _ = conv(self.same_conv, yf, yo, c1=18)
        x = self.res_block(x)
        return Fh.fm_fuse(x, yf, self.activation, self.arith_strategy), None


class FMNone(nn.Module):
    def forward(self, yf, yo, yt=None):
        return yf, None
