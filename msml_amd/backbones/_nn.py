"""Glue between torch parameter containers (nn.Conv2d, nn.BatchNorm2d, nn.PReLU ... kept so
that names, shapes and initialisation equal the reference's) and the HIP autograd functions.
The containers' own forward() is never called: all arithmetic goes through the C ABI."""
import torch
import torch.nn as nn

from .. import functional as Fh


def conv(m, x0, x1=None, want_stats=False, c0=None, c1=0, tee=False):
    """Run nn.Conv2d / nn.ConvTranspose2d `m` on NHWC input(s).  x1: second concat segment.
    tee: x0 has a second consumer -- also return an alias of x0 for it (Fh.conv_tee: the two gradients of x0 are summed
    in the backward-data epilogue)."""
    deconv = isinstance(m, nn.ConvTranspose2d)
    cin = m.in_channels
    if c0 is None:
        c0 = cin - c1
    assert c0 + c1 == cin
    if isinstance(x0, Fh.SplitT):              # split-bf16 inference (functional.py, csrc/x3.hip)
        return Fh.conv_plain_x3(m, x0, x1, c1), None
    assert m.groups == 1 and m.dilation == (1, 1) and m.stride[0] == m.stride[1]
    cfg = {"deconv": deconv, "c0": c0, "c1": c1, "cout": m.out_channels, "stride": m.stride[0],
           "pad_h": m.padding[0], "pad_w": m.padding[1], "want_stats": want_stats}
    # packed operands are cached / refreshed by ops.PACKS (keyed on parameter version)
    if tee:
        return Fh.conv_tee(x0, x1, m.weight, m.bias, cfg)
    return Fh.conv(x0, x1, m.weight, m.bias, cfg)


def conv_bn(conv_m, bn_m, x0, x1=None, prelu=None, residual=None, c1=0, res_first=False):
    """conv -> BatchNorm (-> PReLU) (+ residual); BN statistics come from the conv epilogue.
    Inference (eval mode, bf16, no autograd): BatchNorm / PReLU / residual are folded into the
    conv epilogue (one kernel, no intermediate tensor)."""
    if isinstance(x0, Fh.RawImage):
        cw = conv_m.weight
        if (x1 is None and residual is None and conv_m.bias is None and not isinstance(conv_m, nn.ConvTranspose2d)
                and cw.shape[1] * cw.shape[2] * cw.shape[3] <= 32 and conv_m.out_channels % 32 == 0
                and conv_m.padding[0] == conv_m.padding[1] and cw.shape[2] == cw.shape[3]):
            return Fh.stem_conv_bn(x0, conv_m, bn_m, prelu)
        if x0.x3:
            raise RuntimeError("msml_amd: split-bf16 inference expects the im2col stem shape")
        x0 = x0.nhwc()
    if isinstance(x0, Fh.SplitT):
        return Fh.conv_bn_eval_x3(x0, x1, conv_m, bn_m, prelu, residual, c1, res_first)
    if (not bn_m.training and not torch.is_grad_enabled() and x0.dtype == torch.bfloat16
            and conv_m.bias is None and not isinstance(conv_m, nn.ConvTranspose2d)
            and conv_m.out_channels % 32 == 0):
        return Fh.conv_bn_eval(x0, x1, conv_m, bn_m, prelu, residual, c1, res_first)
    y, stats = conv(conv_m, x0, x1, want_stats=bn_m.training, c1=c1)
    return Fh.bn_act(y, stats, bn_m, prelu, residual, res_first)
