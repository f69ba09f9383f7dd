"""Occlusion Segmentation Branch on the HIP path (reference: backbones/osb/unet.py:94-240).

IResNet-18 encoder (stem stride 2) + Global-Convolution modules (7x1/1x7, biased) + transposed
convolutions on cat(seg, gcm) (two-segment implicit GEMM, concat never materialised) + DAP.
Returns [seg0, seg1, seg2, seg3 (detached NHWC 18-channel maps), seg5 (NCHW f32 (B,2,H,W))]."""
import torch
import torch.nn as nn

from ... import functional as Fh
from ... import ops
from .._nn import conv, conv_bn
from ..frb.iresnet import IBasicBlock, make_layer

__all__ = ["unet", "Unet"]


class _GlobalConvModule(nn.Module):
    def __init__(self, in_dim, out_dim, kernel_size):
        super().__init__()
        pad0 = (kernel_size[0] - 1) // 2
        pad1 = (kernel_size[1] - 1) // 2
        self.conv_l1 = nn.Conv2d(in_dim, out_dim, kernel_size=(kernel_size[0], 1), padding=(pad0, 0))
        self.conv_l2 = nn.Conv2d(out_dim, out_dim, kernel_size=(1, kernel_size[1]), padding=(0, pad1))
        self.conv_r1 = nn.Conv2d(in_dim, out_dim, kernel_size=(1, kernel_size[1]), padding=(0, pad1))
        self.conv_r2 = nn.Conv2d(out_dim, out_dim, kernel_size=(kernel_size[0], 1), padding=(pad0, 0))

    def forward(self, x):
        if (ops.GCM_TEE and torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad
                and x.dtype == torch.bfloat16 and (x.shape[1] in (56, 28) or ops.GCM_TEE == "all")):
            # (the levels whose line convs run on k_conv_line; below, the sum is a few-MB add and the general kernel's
            # residual epilogue costs more than it)
            # x feeds both branches: their two input gradients meet in conv_l1's backward-data epilogue
            xl, _, x = conv(self.conv_l1, x, tee=True)
        else:
            xa, x = Fh.fanout2(x)
            xl, _ = conv(self.conv_l1, xa)
        xl, _ = conv(self.conv_l2, xl)
        xr, _ = conv(self.conv_r1, x)
        xr, _ = conv(self.conv_r2, xr)
        return Fh.add(xl, xr)


class Unet(nn.Module):
    def __init__(self, block, layers, groups=1, num_classes=2, kernel_size=7, dap_k=3, gray=True,
                 input_size=128):
        super().__init__()
        if gray or input_size != 112 or dap_k != 3 or num_classes != 2:
            raise NotImplementedError("msml_amd: OSB is built for RGB 112x112, 2 classes, DAP k=3")
        self.conv1 = nn.Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=1e-05)
        self.prelu = nn.PReLU(64)
        self.layer1 = make_layer(block, 64, 64, layers[0], 2)
        self.layer2 = make_layer(block, 64, 128, layers[1], 2)
        self.layer3 = make_layer(block, 128, 256, layers[2], 2)
        self.layer4 = make_layer(block, 256, 512, layers[3], 2)
        self.bn2 = nn.BatchNorm2d(512 * block.expansion, eps=1e-05)
        s = num_classes * dap_k ** 2
        ks = (kernel_size, kernel_size)
        self.gcm1 = _GlobalConvModule(512, num_classes * 4, ks)
        self.gcm2 = _GlobalConvModule(256, s, ks)
        self.gcm3 = _GlobalConvModule(128, s, ks)
        self.gcm4 = _GlobalConvModule(64, s, ks)
        self.gcm5 = _GlobalConvModule(64, s, ks)
        self.deconv1 = nn.ConvTranspose2d(num_classes * 4, s, kernel_size=3, stride=2, padding=1,
                                          bias=False)
        self.deconv2 = nn.ConvTranspose2d(2 * s, s, kernel_size=4, stride=2, padding=1, bias=False)
        self.deconv3 = nn.ConvTranspose2d(2 * s, s, kernel_size=4, stride=2, padding=1, bias=False)
        self.deconv4 = nn.ConvTranspose2d(2 * s, s, kernel_size=4, stride=2, padding=1, bias=False)
        self.deconv5 = nn.ConvTranspose2d(2 * s, s, kernel_size=4, stride=2, padding=1, bias=False)
        self.DAP = nn.Sequential(nn.PixelShuffle(dap_k), nn.AvgPool2d((dap_k, dap_k)))
        self.s = s

    def forward(self, x):
        # (every encoder output has two consumers, the next stage and a GCM: Fh.fanout2 sums their gradients)
        x0, x0g = Fh.fanout2(conv_bn(self.conv1, self.bn1, x, prelu=self.prelu))
        x1, x1g = Fh.fanout2(self.layer1(x0))
        x2, x2g = Fh.fanout2(self.layer2(x1))
        x3, x3g = Fh.fanout2(self.layer3(x2))
        x4 = self.layer4(x3)
        xx = Fh.bn_act(x4, None, self.bn2)
        seg0, _ = conv(self.deconv1, self.gcm1(xx))
        seg1, _ = conv(self.deconv2, seg0, self.gcm2(x3g), c1=self.s)
        seg2, _ = conv(self.deconv3, seg1, self.gcm3(x2g), c1=self.s)
        seg3, _ = conv(self.deconv4, seg2, self.gcm4(x1g), c1=self.s)
        seg5_, _ = conv(self.deconv5, seg3, self.gcm5(x0g), c1=self.s)
        seg5 = Fh.dap(seg5_)
        return [seg0.detach(), seg1.detach(), seg2.detach(), seg3.detach(), seg5]


def unet(pre_trained=False, backbone="r18", gray=True, input_size=128, **kwargs):
    layers = {"r18": [2, 2, 2, 2], "r34": [3, 4, 6, 3], "r50": [3, 4, 14, 3],
              "r100": [3, 13, 30, 3]}
    for key, ls in layers.items():
        if key in backbone:
            return Unet(IBasicBlock, ls, num_classes=2, gray=gray, input_size=input_size, **kwargs)
    raise ValueError("Error backbone type in OSB.")
