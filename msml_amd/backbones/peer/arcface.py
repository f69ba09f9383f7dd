"""Frozen teacher ("peer") networks on the HIP path (reference: backbones/peer/arcface.py).

The vanilla IResNet of the reference's peer package (:72-194): stem -> 4 stages -> bn2 -> fc ->
BatchNorm1d, returning the embedding AND the four stage outputs that the FM operators consume as
peer knowledge (`inter`, :159-194).  Same constructor / module names / state-dict keys as the
reference, same factories (:215-237) with the same cwd-relative checkpoint paths (:10-16); weights
load with `load_state_dict` from a reference-produced `r{18,34,50}-backbone.pth`.

Quirk kept: the factories return the net in eval mode, but `backbone.train()`
(train.py:138) flips it back, so during training the teacher's BatchNorms use batch statistics
and update their running statistics exactly like the reference's.
"""
import os

import torch
from torch import nn

from ... import functional as Fh
from ..._lib import BF16, F32
from .._nn import conv_bn
from ..frb.iresnet import IBasicBlock, make_layer

__all__ = ["arcface18", "arcface34", "arcface50", "arcface100", "cosface50_casia", "IResNet"]

model_dir = {
    "arcface18": "./backbones/pretrained/r18-backbone.pth",
    "arcface34": "./backbones/pretrained/r34-backbone.pth",
    "arcface50": "./backbones/pretrained/r50-backbone.pth",
    "arcface100": "./backbones/pretrained/r100-backbone.pth",
    "cosface50_casia": "./backbones/pretrained/cos50_no_occ_2.pth",
}


class IResNet(nn.Module):
    fc_scale = 7 * 7

    def __init__(self, block, layers, dim_feature=512, dropout=0, zero_init_residual=False, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, fp16=False):
        super().__init__()
        if groups != 1 or width_per_group != 64 or (replace_stride_with_dilation and any(replace_stride_with_dilation)):
            raise ValueError("peer IResNet only supports groups=1, base_width=64, no dilation")
        self.fp16 = fp16
        self.conv1 = nn.Conv2d(3, 64, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=1e-05)
        self.prelu = nn.PReLU(64)
        self.layer1 = make_layer(block, 64, 64, layers[0], 2)
        self.layer2 = make_layer(block, 64, 128, layers[1], 2)
        self.layer3 = make_layer(block, 128, 256, layers[2], 2)
        self.layer4 = make_layer(block, 256, 512, layers[3], 2)
        self.bn2 = nn.BatchNorm2d(512 * block.expansion, eps=1e-05)
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        self.fc = nn.Linear(512 * block.expansion * self.fc_scale, dim_feature)
        self.features = nn.BatchNorm1d(dim_feature, eps=1e-05)
        nn.init.constant_(self.features.weight, 1.0)
        self.features.weight.requires_grad = False
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.1)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, IBasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def stages(self, x):
        """x: functional.RawImage or NHWC storage tensor.  Returns (last map, [4 NHWC stage outputs])."""
        x = conv_bn(self.conv1, self.bn1, x, prelu=self.prelu)
        inter = []
        for k in range(4):
            x = getattr(self, "layer%d" % (k + 1))(x)
            inter.append(x.detach())
        return x, inter

    def embed(self, x):
        """bn2 -> flatten -> fc -> BatchNorm1d on the last NHWC map."""
        x = Fh.bn_act(x, None, self.bn2)
        if self.dropout.p > 0 and self.training:
            x = Fh.dropout(x, self.dropout.p)
        n, h, w, c = x.shape
        y = Fh.flat_fc(x, self.fc.weight.view(self.fc.out_features, c, h, w), self.fc.bias, self.fc.weight)
        y = Fh.bn_act(y, None, self.features)
        return Fh.to_vec(y, self.fc.out_features)

    def forward_nhwc(self, x, want_feature=False):
        """Internal entry (FRB): intermediates stay NHWC storage tensors; the embedding (unused by the
        FM operators, arcface.py:193 `_`) is only computed on request."""
        with torch.no_grad():
            last, inter = self.stages(x)
            feat = self.embed(last) if want_feature else None
        return feat, inter

    def forward(self, x):
        """Reference signature (arcface.py:159-194): img (B, 3, 112, 112) NCHW ->
        (feature (B, dim), [ft0 (B,64,56,56), ft1 (B,128,28,28), ft2 (B,256,14,14), ft3 (B,512,7,7)])."""
        if not x.is_cuda:
            raise RuntimeError("msml_amd peer networks run on an MI355X only (no CPU path)")
        from ... import ops
        xh = Fh.RawImage(x.float()) if self.fp16 else Fh.to_nhwc(x, F32)
        last, inter = self.stages(xh)
        feat = self.embed(last)
        return feat, [ops.to_nchw(t, c) for t, c in zip(inter, (64, 128, 256, 512))]


def _iresnet_v(arch, block, layers, pretrained, progress, **kwargs):
    model = IResNet(block, layers, **kwargs)
    if pretrained:
        if os.path.isfile(model_dir[arch]):
            weight = torch.load(model_dir[arch], map_location=torch.device("cpu"))
            model.load_state_dict(weight)
        else:
            raise FileNotFoundError("Make sure the file {" + model_dir[arch] + "} exists!")
    return model.eval()


def arcface18(pretrained=True, progress=True, **kwargs):
    return _iresnet_v("arcface18", IBasicBlock, [2, 2, 2, 2], pretrained, progress, **kwargs)


def arcface34(pretrained=True, progress=True, **kwargs):
    return _iresnet_v("arcface34", IBasicBlock, [3, 4, 6, 3], pretrained, progress, **kwargs)


def arcface50(pretrained=True, progress=True, **kwargs):
    return _iresnet_v("arcface50", IBasicBlock, [3, 4, 14, 3], pretrained, progress, **kwargs)


def arcface100(pretrained=True, progress=True, **kwargs):
    return _iresnet_v("arcface100", IBasicBlock, [3, 13, 30, 3], pretrained, progress, **kwargs)


def cosface50_casia(pretrained=True, progress=True, **kwargs):
    return _iresnet_v("cosface50_casia", IBasicBlock, [3, 4, 14, 3], pretrained, progress, **kwargs)
