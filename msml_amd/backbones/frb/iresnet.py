"""Face Recognition Branch: IResNet with Feature-Masking hooks, on the HIP path.

Mirrors backbones/frb/iresnet.py of the reference (IBasicBlock :38-67, IResNet :70-236,
factories :444-481): same constructor arguments that matter for the iresnet path, same module
/ parameter names (state dicts interchange, `strict=True`), same initialisation; forward runs
on NHWC tensors through libmsml_hip.so.  Peer / decoder branches (use_ori, use_decoder) are
out of scope this round and raise if requested.
"""
import torch
from torch import nn

from ... import blocks, ops
from ... import functional as Fh
from .._nn import conv, conv_bn

__all__ = ["iresnet18", "iresnet34", "iresnet50", "iresnet100", "IResNet", "IBasicBlock"]


def conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


class IBasicBlock(nn.Module):
    """bn1 -> conv3x3 -> bn2 -> PReLU -> conv3x3(stride) -> bn3 (+ downsample) + identity."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1):
        super().__init__()
        if groups != 1 or base_width != 64 or dilation > 1:
            raise ValueError("IBasicBlock only supports groups=1, base_width=64, dilation=1")
        self.bn1 = nn.BatchNorm2d(inplanes, eps=1e-05)
        self.conv1 = conv3x3(inplanes, planes)
        self.bn2 = nn.BatchNorm2d(planes, eps=1e-05)
        self.prelu = nn.PReLU(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn3 = nn.BatchNorm2d(planes, eps=1e-05)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if self.training and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and ops.BLOCK_FUNCTION:
            return blocks.iblock(self, x)          # one autograd node, fused backward (blocks.py)
        out = Fh.bn_conv_bn_eval_x3(x, self.bn1, self.conv1, self.bn2, self.prelu) if isinstance(x, Fh.SplitT) else None
        if out is None:
            out = Fh.bn_act(x, None, self.bn1)
            out = conv_bn(self.conv1, self.bn2, out, prelu=self.prelu)
        identity = x
        if self.downsample is not None:
            identity = conv_bn(self.downsample[0], self.downsample[1], x)
        return conv_bn(self.conv2, self.bn3, out, residual=identity)


def make_layer(block, inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(conv1x1(inplanes, planes * block.expansion, stride),
                                   nn.BatchNorm2d(planes * block.expansion, eps=1e-05))
    layers = [block(inplanes, planes, stride, downsample)]
    layers += [block(planes * block.expansion, planes) for _ in range(1, blocks)]
    for blk in layers[:-1]:
        blk.emit_stats = True          # its output feeds another block's bn1 (blocks.py)
    return nn.Sequential(*layers)


class IResNet(nn.Module):
    fc_scale = 7 * 7

    def __init__(self, block, layers, fm_ops, dim_feature=512, dropout=0, zero_init_residual=False,
                 groups=1, width_per_group=64, replace_stride_with_dilation=None, fp16=False,
                 peer_params: dict = None):
        super().__init__()
        peer_params = peer_params or {}
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
        assert len(fm_ops) == 4
        self.fm_ops = nn.ModuleList(fm_ops)
        # Peer (frozen teacher; its type follows msml.header_type, iresnet.py:126-144)
        from ..peer import arcface18, arcface34, arcface50, cosface50_casia
        self.peer = None
        self.header_type = str(peer_params.get("header_type", "")).lower()
        if peer_params.get("use_ori"):
            layers = list(layers)
            if "arc" in self.header_type:
                ctor = {(2, 2, 2, 2): arcface18, (3, 4, 6, 3): arcface34, (3, 4, 14, 3): arcface50}.get(tuple(layers))
                if ctor is not None:
                    self.peer = ctor().requires_grad_(False)
            elif "cos" in self.header_type:
                if layers == [3, 4, 14, 3]:
                    self.peer = cosface50_casia().requires_grad_(False)
            else:
                raise ValueError("Error type of iresnet, cannot decide peer network.")
        # Recover decoder (iresnet.py:146-150): parameters only -- its output and loss are dead in the
        # reference, so forward() below does not run it
        self.decoder = None
        if peer_params.get("use_decoder"):
            from ..decoder import dm_decoder
            self.decoder = dm_decoder(n_init=dim_feature)
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

    def forward(self, x, segs, ori=None, wait_segs=None):
        """x: NHWC image; segs: [seg3, seg2, seg1, seg0] NHWC 18-channel maps (detached).
        wait_segs: event after which `segs` are valid (OSB running on another stream).
        Returns (feature (B, dim) f32, kd)."""
        ft = (None, None, None, None)
        if ori is not None:                       # peer knowledge (iresnet.py:203-206)
            if self.peer is None:
                raise TypeError("'NoneType' object is not callable: `ori` given but no peer network was built "
                                "(peer_params.use_ori False)")
            _, ft = self.peer.forward_nhwc(ori)
        x = conv_bn(self.conv1, self.bn1, x, prelu=self.prelu)
        kd = None
        for k in range(4):
            x = getattr(self, "layer%d" % (k + 1))(x)
            if k == 0 and wait_segs is not None:
                torch.cuda.current_stream().wait_event(wait_segs)
            x, l = self.fm_ops[k](x, segs[k], ft[k])
            if ori is not None:                   # l + l1 + l2 + l3 (iresnet.py:234)
                kd = l if kd is None else kd + l
        x = Fh.bn_act(x, None, self.bn2)
        # (recover decoder: `_rec, l4 = self.decoder(x, ori) if ori is not None else None, 0.` gives l4 = 0
        # and discards _rec in the reference, iresnet.py:228,235 -- nothing to compute)
        if self.dropout.p > 0 and self.training:
            x = Fh.dropout(x, self.dropout.p)
        # flatten(C,H,W) + Linear(25088, 512): skinny GEMM on the NHWC-ordered operand
        n, h, w, c = x.shape
        if isinstance(x, Fh.SplitT):
            y = Fh.flat_fc_x3(x, self.fc)          # f32 [N,1,1,E]; BatchNorm1d below runs in f32
        else:
            wview = self.fc.weight.view(self.fc.out_features, c, h, w)
            y = Fh.flat_fc(x, wview, self.fc.bias, self.fc.weight)
        y = Fh.bn_act(y, None, self.features)
        return Fh.to_vec(y, self.fc.out_features), (kd * 1.0 if kd is not None else 0.0)


def _iresnet(layers, fm_ops, pretrained, **kw):
    if pretrained:
        raise NotImplementedError("msml_amd: pretrained FRB weights are loaded via load_state_dict")
    return IResNet(IBasicBlock, layers, fm_ops, **kw)


def iresnet18(fm_ops, pretrained=False, dim_feature=512, dropout=0., peer_params=None):
    return _iresnet([2, 2, 2, 2], fm_ops, pretrained, dim_feature=dim_feature, dropout=dropout,
                    peer_params=peer_params)


def iresnet34(fm_ops, pretrained=False, dim_feature=512, dropout=0., peer_params=None):
    return _iresnet([3, 4, 6, 3], fm_ops, pretrained, dim_feature=dim_feature, dropout=dropout,
                    peer_params=peer_params)


def iresnet50(fm_ops, pretrained=False, dim_feature=512, dropout=0., peer_params=None):
    return _iresnet([3, 4, 14, 3], fm_ops, pretrained, dim_feature=dim_feature, dropout=dropout,
                    peer_params=peer_params)


def iresnet100(fm_ops, pretrained=False, dim_feature=512, dropout=0., peer_params=None):
    """[3, 13, 30, 3]: not constructible through the reference's MSML; provided for
    BASELINE config 4 with the same block / FM wiring."""
    return _iresnet([3, 13, 30, 3], fm_ops, pretrained, dim_feature=dim_feature, dropout=dropout,
                    peer_params=peer_params)
