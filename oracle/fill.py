"""Name-keyed deterministic weight fill ('Travel rule').

Weights never travel: the reference (in the build container), this oracle and the HIP model
are all filled from their state-dict KEYS alone, so the three hold identical parameters
without shipping a checkpoint.  Fan-in scaling with gain 0.5 keeps activations O(1-10) through
100 layers in eval mode (running statistics do not normalise there), so the f32 goldens are
well conditioned: f32-vs-f64 feature error of the reference itself is ~1e-6 for ires18/50/100.
(Gain 2 grows activations to 1e7 at ires50 and the reference's own f32 result is then only
accurate to 5e-3; the reference's N(0, 0.1) conv init, frb/iresnet.py:152-154, is worse.)
"""
import zlib

import torch

GAIN = 0.5


def fill_state_dict(sd, gain=None):
    """Fill every tensor of ``sd`` in place, in state_dict order, from its key.  gain: variance gain of
    the conv / linear weights (default GAIN = 0.5; 2.0 = the survey's sqrt(2 / fan_in))."""
    gain = GAIN if gain is None else gain
    with torch.no_grad():
        for key, t in sd.items():
            g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
            leaf = key.rsplit(".", 1)[-1]
            if leaf == "num_batches_tracked":
                t.zero_()
            elif leaf == "running_mean":
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif leaf == "running_var":
                t.copy_(1.0 + 0.1 * torch.rand(t.shape, generator=g))
            elif t.dim() >= 2:                      # conv / deconv / linear / classifier weight
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (gain / fan_in) ** 0.5)
            elif leaf == "bias":
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif leaf == "weight":
                owner = key.rsplit(".", 2)[-2]
                if "prelu" in owner:
                    t.copy_(0.25 + 0.05 * torch.randn(t.shape, generator=g))
                else:                               # BatchNorm gamma
                    t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            else:
                raise KeyError(key)
    return sd


def calibrate_running_stats(module, run_forward):
    """Set every BatchNorm's running statistics to the statistics of ONE calibration batch: momentum 1
    for a single train-mode forward (`run_forward(module)`, no gradients), then everything is restored.
    With calibrated statistics the eval-mode activations are normalised whatever the weight gain."""
    import torch.nn as nn
    bns = [m for m in module.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]
    saved = [(m.momentum, m.training) for m in bns]
    was_training = module.training
    module.train()
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        run_forward(module)
    for m, (mom, _) in zip(bns, saved):
        m.momentum = mom
        m.num_batches_tracked.zero_()
    module.train(was_training)
    return module


def fill_module(module, gain=None):
    fill_state_dict(module.state_dict(), gain)
    # frb.features.weight is frozen at 1.0 in the reference (iresnet.py:118-120)
    feats = getattr(getattr(module, "frb", module), "features", None)
    if feats is not None:
        with torch.no_grad():
            feats.weight.fill_(1.0)
    return module
