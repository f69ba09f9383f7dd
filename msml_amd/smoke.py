"""smoke(): one small pass of the hot path on cuda:0 checked against the CPU oracle."""
import torch


def run():
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    from msml_amd import _lib
    _lib.load()
    # FM fusion against the oracle formula (fmoperator.py:288,304-310)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 56, 56, 64, generator=g)
    yf = torch.randn(2, 56, 56, 64, generator=g)
    z = torch.empty_like(x, device="cuda")
    _lib.call("msml_fm_fuse_fwd", x.cuda(), yf.cuda(), z, x.numel(), 1, 2, _lib.F32)
    ref = yf * torch.sigmoid(x) + yf
    torch.cuda.synchronize()
    err = (z.cpu() - ref).abs().max().item()
    assert err < 1e-5, err
    print("smoke ok: fm_fuse max err %.2e" % err)
