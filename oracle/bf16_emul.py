"""bf16 rounding model of the HIP training path, applied to the CPU oracle (test infrastructure only).

The bf16 training path stores every activation and every activation gradient as bf16 and feeds bf16 operands to
f32-accumulating MFMAs; parameters, parameter gradients, BatchNorm statistics, losses and the logits stay f32.
`emulate(model)` installs that rounding on an oracle model (oracle/model.py) with plain PyTorch hooks -- no HIP code
involved -- so that the error of the emulated step against the f32 reference golden is an INDEPENDENT measurement of
what bf16 costs on a given network / batch:

  * conv / transposed conv: input and weight rounded (the weight in the forward only: its gradient is an f32
    accumulation of bf16 products), output rounded, output gradient rounded;
  * BatchNorm2d / PReLU / residual adds / FM fusion / GCM sums: output and output gradient rounded (the element-wise
    kernels read bf16, compute in f32, write bf16);
  * flatten + Linear (iresnet.py:230-232): bf16 operands, f32 output, output gradient rounded before the two
    backward GEMMs; BatchNorm1d and the loss tail stay f32;
  * cosine head (margin_losses.py:356-418): normalised embedding and normalised weight rounded to bf16, cosine f32,
    its gradient rounded to bf16 before the backward GEMMs.

tests/ use the per-group error floors recorded by oracle/make_bf16_floor.py from this model to DERIVE the tolerances
of the bf16 parity tests instead of fitting them to the HIP path's own error.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import model as om


GRID_SHIFT = 0.0      # draw selector, see _r


def _r(t):
    """Round to bf16 (nearest-even).  GRID_SHIFT = u != 0 selects another DRAW of the same rounding noise: the
    quantiser grid is shifted by u ulp (q_u(x) = rne(x + u ulp(x)) - u ulp(x)), which keeps the error distribution
    (uniform in +-ulp/2) but decorrelates the individual errors from the u = 0 draw.  The floor a test tolerance is
    derived from is the maximum over a few draws: one draw of a heavy-tailed error (BatchNorm1d over 4 samples in
    front of the s = 64 head) says little about the next."""
    if GRID_SHIFT == 0.0:
        return t.to(torch.bfloat16).to(torch.float32)
    ulp = torch.exp2(torch.floor(torch.log2(t.abs().clamp_min(1e-30))) - 7.0) * GRID_SHIFT
    return (t + ulp).to(torch.bfloat16).to(torch.float32) - ulp


class _RoundBoth(torch.autograd.Function):
    """bf16 storage of an activation: value rounded in the forward, its gradient rounded in the backward."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundFwd(torch.autograd.Function):
    """bf16 operand copy of an f32 master (packed weights): the gradient is not rounded."""

    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """f32 value whose GRADIENT is stored as bf16 (logits / fc output)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


def both(x):
    return _RoundBoth.apply(x)


def _conv_forward(mod):
    def fwd(x):
        w = _RoundFwd.apply(mod.weight)
        if isinstance(mod, nn.ConvTranspose2d):
            y = F.conv_transpose2d(x, w, mod.bias, mod.stride, mod.padding, mod.output_padding, mod.groups, mod.dilation)
        else:
            y = F.conv2d(x, w, mod.bias, mod.stride, mod.padding, mod.dilation, mod.groups)
        return both(y)
    return fwd


def _linear_forward(mod):
    def fwd(x):
        return _RoundBwd.apply(F.linear(x, _RoundFwd.apply(mod.weight), mod.bias))
    return fwd


def _head_forward(head):
    def fwd(emb, label):
        cos = F.linear(both(F.normalize(emb)), both(F.normalize(head.weight)))
        cos = _RoundBwd.apply(cos)
        return om.margin_logits(cos, label, head.kind, head.s, head.m, head.a, head.k)
    return fwd


def _out_hook(mod, inp, out):
    if isinstance(out, tuple):
        return (both(out[0]),) + tuple(out[1:])
    return both(out)


def emulate(model):
    """Install the bf16 rounding model on an oracle MSML (in place; returns the model)."""
    for mod in model.modules():
        if isinstance(mod, (nn.Conv2d, nn.ConvTranspose2d)):
            mod.forward = _conv_forward(mod)
        elif isinstance(mod, nn.Linear):
            mod.forward = _linear_forward(mod)
        elif isinstance(mod, om._CosHead):
            mod.forward = _head_forward(mod)
        elif isinstance(mod, (nn.BatchNorm2d, nn.PReLU, om.IBasicBlock, om.ResBottle, om.GCM, om.FMCnn)):
            mod.register_forward_hook(_out_hook)
    return model


def param_group(name):
    """Tolerance group of a picked parameter gradient: the bf16 error grows with the depth of the backward chain
    behind it, so the tests bound it per group (2 x the group's emulated floor)."""
    if name.startswith("osb."):
        return "osb"
    if name in ("classification.weight", "frb.fc.weight", "frb.fc.bias"):
        return "head"
    if name.startswith(("frb.conv1", "frb.bn1", "frb.prelu", "frb.layer1.", "frb.layer2.", "frb.fm_ops.0.",
                        "frb.fm_ops.1.")):
        return "frb_early"
    return "frb_late"
