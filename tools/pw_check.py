"""Pointwise kernel vs the general kernel at the step's own sizes (batch 256): outputs bit for bit, sums to 1e-5."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from msml_amd import _lib, ops
def both(fn):
    os.environ["MSML_PW_CONV"] = "1"; a = fn(); torch.cuda.synchronize()
    os.environ["MSML_PW_CONV"] = "0"; b = fn(); torch.cuda.synchronize()
    os.environ.pop("MSML_PW_CONV"); return a, b
for cin, cout, n, h in [(32, 64, 256, 112), (32, 64, 256, 56), (64, 32, 256, 56), (64, 128, 256, 28), (128, 64, 256, 28), (64, 64, 256, 28)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, h, h, cin, device="cuda")).bfloat16()
    w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).bfloat16().float().cuda()
    wp = ops.pack_weight(w, False, cin, 0, _lib.BF16); wpt = ops.pack_weight(w, True, cout, 0, _lib.BF16)
    (o1, s1), (o2, s2) = both(lambda: ops.conv2d(x, None, wp, None, cout, 1, 1, 1, 0, 0, False, want_stats=True))
    r = [torch.equal(o1, o2), float(((s1.sum(0) - s2.sum(0)).abs() / s2.sum(0).abs().max()).max())]
    dy = torch.randn(n, h, h, cout, device="cuda").bfloat16()
    (d1, _), (d2, _) = both(lambda: ops.conv2d(dy, None, wpt, None, cin, 1, 1, 1, 0, 0, True, p=h, q=h))
    r.append(torch.equal(d1, d2))
    other = torch.randn(n, h, h, cin, device="cuda").bfloat16()
    ones, zeros = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    def plus():
        out = torch.empty(n, h, h, cin, dtype=torch.bfloat16, device="cuda")
        _lib.call("msml_conv2d_fused", dy, cout, None, 0, wpt, wpt.shape[0], ones, zeros, None, other, 0, out, cin, n, h, h, h, h, 1, 1, 1, 0, 0, 1)
        return out
    p1, p2 = both(plus); r.append(torch.equal(p1, p2))
    coef = torch.stack([torch.rand(cin) + 0.5, torch.randn(cin) * 0.3, torch.randn(cin) * 0.2, torch.rand(cin) + 0.5]).cuda()
    alpha = (torch.rand(cin) * 0.5).cuda()
    (b1, q1), (b2, q2) = both(lambda: ops.conv_dgrad_bnbwd(dy, wpt, cin, 1, 1, 1, 0, 0, h, h, other, coef, alpha))
    r += [torch.equal(b1, b2), float(((q1.sum(0) - q2.sum(0)).abs() / q2.sum(0).abs().max()).max())]
    print(cin, cout, n, h, "fwd eq %s stats %.1e | dgrad eq %s | +add eq %s | +bnb eq %s sums %.1e" % tuple(r))
