"""GPU parity: boundary layout kernels, weight packing, FM fusion (fwd + bwd) vs the oracle
formulas.  Calls go through the C ABI (msml_amd._lib.call)."""
import pytest
import torch

from msml_amd import _lib
from oracle import model as om

pytestmark = pytest.mark.gpu
DT = [(_lib.F32, 1e-6), (_lib.BF16, 1e-2)]


@pytest.mark.parametrize("dtype,tol", DT)
@pytest.mark.parametrize("C,Cp", [(3, 8), (18, 24), (64, 64)])
def test_layout_roundtrip(dtype, tol, C, Cp):
    x = torch.randn(3, C, 14, 10)
    d = torch.empty(3, 14, 10, Cp, dtype=_lib.TORCH_DTYPE[dtype], device="cuda")
    _lib.call("msml_nchw_to_nhwc", x.cuda(), d, 3, C, 14, 10, Cp, dtype)
    ref = torch.zeros(3, 14, 10, Cp)
    ref[..., :C] = x.permute(0, 2, 3, 1)
    assert (d.float().cpu() - ref).abs().max() <= tol * 4
    assert (d[..., C:] == 0).all()
    back = torch.empty(3, C, 14, 10, device="cuda")
    _lib.call("msml_nhwc_to_nchw", d, back, 3, C, 14, 10, Cp, dtype)
    assert (back.cpu() - x).abs().max() <= tol * 4


@pytest.mark.parametrize("transpose", [0, 1])
def test_pack_weight(transpose):
    A, B, R, S = 5, 26, 3, 3
    w = torch.randn(A, B, R, S)
    CI, KO = (A, B) if transpose else (B, A)
    C1 = CI - 2 if CI > 4 else CI
    C2 = CI - C1
    C1p = (C1 + 7) // 8 * 8
    C2p = (C2 + 7) // 8 * 8
    KOp = 32
    K0 = (R * S * C1p + 31) // 32 * 32
    K1 = (R * S * C2p + 31) // 32 * 32 if C2 else 0
    dst = torch.full((KOp, K0 + K1), 7.0, device="cuda")
    _lib.call("msml_pack_weight", w.cuda(), dst, A, B, R, S, transpose, C1, C1p, C2, C2p, KOp,
              _lib.F32)
    wt = w.permute(1, 0, 2, 3) if transpose else w          # [KO][CI][R][S]
    ref = torch.zeros(KOp, K0 + K1)
    seg0 = torch.zeros(KO, R, S, C1p)
    seg0[..., :C1] = wt[:, :C1].permute(0, 2, 3, 1)
    ref[:KO, :R * S * C1p] = seg0.reshape(KO, -1)
    if C2:
        seg1 = torch.zeros(KO, R, S, C2p)
        seg1[..., :C2] = wt[:, C1:].permute(0, 2, 3, 1)
        ref[:KO, K0:K0 + R * S * C2p] = seg1.reshape(KO, -1)
    assert torch.equal(dst.cpu(), ref)


ACTS = {"tanh": 0, "sigmoid": 1}
ARITHS = {"add": 0, "sub": 1, "mul": 2, "div": 3}


@pytest.mark.parametrize("dtype,tol", DT)
@pytest.mark.parametrize("act", list(ACTS))
@pytest.mark.parametrize("arith", list(ARITHS))
def test_fm_fuse(dtype, tol, act, arith):
    g = torch.Generator().manual_seed(3)
    tdt = _lib.TORCH_DTYPE[dtype]
    x = torch.randn(2, 28, 28, 128, generator=g).to(tdt).float()
    yf = torch.randn(2, 28, 28, 128, generator=g).to(tdt).float()
    if arith == "div":
        x = (x.abs() + 0.5).to(tdt).float()   # keep 1/M tame: the domain where div is usable
    dz = torch.randn(2, 28, 28, 128, generator=g).to(tdt).float()
    xr = x.clone().requires_grad_(True)
    yr = yf.clone().requires_grad_(True)
    m = om._ACT[act](xr)
    zref = om._ARITH[arith](yr, m) + yr
    zref.backward(dz)
    xd, yd, dzd = x.cuda().to(tdt), yf.cuda().to(tdt), dz.cuda().to(tdt)
    z = torch.empty_like(xd)
    _lib.call("msml_fm_fuse_fwd", xd, yd, z, x.numel(), ACTS[act], ARITHS[arith], dtype)
    dx, dyf = torch.empty_like(xd), torch.empty_like(xd)
    _lib.call("msml_fm_fuse_bwd", dzd, xd, yd, dx, dyf, x.numel(), ACTS[act], ARITHS[arith], dtype)

    def worst(a, b):
        return ((a.float().cpu() - b).abs() / (1 + b.abs())).max().item()
    assert worst(z, zref.detach()) <= tol
    assert worst(dx, xr.grad) <= tol
    assert worst(dyf, yr.grad) <= tol


def test_fm_full_size_linearity():
    """BASELINE size (256 x 64 x 56 x 56), size-independent property: for arith=add the op is
    affine in yf:  z(yf1 + yf2) - z(yf1) == 2 * yf2 exactly in exact arithmetic."""
    n = 256 * 56 * 56 * 64
    x = torch.randn(n, device="cuda")
    y1 = torch.randn(n, device="cuda")
    y2 = torch.randn(n, device="cuda")
    za, zb = torch.empty_like(x), torch.empty_like(x)
    _lib.call("msml_fm_fuse_fwd", x, y1, za, n, 0, 0, _lib.F32)
    _lib.call("msml_fm_fuse_fwd", x, y1 + y2, zb, n, 0, 0, _lib.F32)
    assert ((zb - za) - 2 * y2).abs().max().item() < 1e-5


def test_pack_refresh_tiled_matches_elementwise():
    """PackCache.refresh() (LDS-tiled batched kernel, real elements only) == the element-wise pack
    of the same sub-blocks: forward / transposed operands, concat segments, sub-block offsets,
    1x1 / 3x3 / 4x4 / 7x7 taps, channel counts that are not multiples of the tile."""
    from msml_amd import ops
    torch.manual_seed(3)
    specs = [  # (shape [A][B][R][S], transpose, a_off, a_n, b_off, b_n, c1, c2)
        ((64, 64, 3, 3), False, 0, 64, 0, 64, 64, 0),
        ((64, 82, 3, 3), False, 0, 64, 0, 82, 64, 18),       # FM same_conv on cat(yf, yo)
        ((128, 64, 3, 3), True, 0, 128, 0, 64, 128, 0),      # backward-data operand
        ((64, 82, 3, 3), True, 0, 64, 64, 18, 64, 0),        # backward-data of the 2nd segment
        ((36, 18, 4, 4), True, 0, 36, 0, 18, 36, 0),         # deconv forward operand
        ((36, 18, 4, 4), False, 18, 18, 0, 18, 18, 0),       # deconv backward, sub-block of rows
        ((18, 64, 7, 1), False, 0, 18, 0, 64, 64, 0),
        ((40, 40, 7, 7), False, 0, 40, 0, 40, 40, 0),
        ((256, 128, 1, 1), False, 0, 256, 0, 128, 128, 0),
    ]
    ws = [torch.randn(sp[0], device="cuda") for sp in specs]
    cache = ops.PackCache()
    for w, sp in zip(ws, specs):
        cache.get(w, *sp[1:], _lib.BF16)
    for w in ws:
        w.mul_(1.5).add_(0.25)
    cache.refresh()
    ref = ops.PackCache()
    for w, sp in zip(ws, specs):
        want = ref.get(w, *sp[1:], _lib.BF16)
        got = cache.get(w, *sp[1:], _lib.BF16)
        assert torch.equal(got, want), sp


@pytest.mark.parametrize("E", [512, 1024, 320])          # 512 / 1024: the 8-per-lane kernels; 320: the scalar ones
@pytest.mark.parametrize("dtype,tol", DT)
def test_rownorm_fwd_bwd(E, dtype, tol):
    """F.normalize of the head weight rows (headers/partial_fc.py:126, margin_losses.py:274) and its backward,
    against f64 torch; padded rows [R, Rp) come back zero."""
    torch.manual_seed(E)
    R, Rp = 301, 320
    w = (torch.randn(R, E, dtype=torch.float64) * 0.3).requires_grad_(True)
    wn = torch.nn.functional.normalize(w)
    dy = torch.randn(R, E, dtype=torch.float64)
    wn.backward(dy)
    wd = w.detach().float().cuda()
    dst = torch.full((Rp, E), 7.0, device="cuda").to(_lib.TORCH_DTYPE[dtype])
    inv = torch.empty(R, device="cuda")
    _lib.call("msml_rownorm_fwd", wd, R, Rp, E, dst, E, inv, dtype)
    assert (dst[:R].double().cpu() - wn.detach()).abs().max() < tol
    assert (dst[R:] == 0).all()
    assert ((inv.double().cpu() - 1.0 / w.detach().norm(dim=1)).abs() * w.detach().norm(dim=1)).max() < 1e-5
    for acc in (0, 1):
        dw = torch.full((R, E), 0.5, device="cuda")
        _lib.call("msml_rownorm_bwd", wd, inv, dy.float().cuda(), E, R, E, dw, acc)
        ref = w.grad + (0.5 if acc else 0.0)
        assert (dw.double().cpu() - ref).abs().max() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("C,ld", [(1000, 1024), (85742, 85760), (37, 37), (5, 8)])
@pytest.mark.parametrize("kind", ["arc", "cos"])
def test_pfc_rowstats(C, ld, kind):
    """Online (row max, sum exp) of the margin logits of one class shard (headers/partial_fc.py:132-141) against
    f64 torch, rows with and without a local target; ld % 4 == 0 takes the 16-B-load kernel, 37 the scalar one."""
    from msml_amd import functional as Fh
    torch.manual_seed(C)
    n, s, m = 9, 64.0, 0.4
    cos = (torch.rand(n, ld, dtype=torch.float64) * 1.6 - 0.8)
    label = torch.randint(0, C, (n,))
    label[::3] = -1
    logits = s * cos[:, :C].clone()
    for i in range(n):
        y = int(label[i])
        if y >= 0:
            th = torch.acos(cos[i, y])
            logits[i, y] = s * torch.cos(th + m) if kind == "arc" else s * (cos[i, y] - m)
    rm, rs = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    _lib.call("msml_pfc_rowstats", cos.float().cuda(), ld, n, C, label.cuda(), Fh.HEAD_KIND[kind], s, m, 0.0, 0.0, rm, rs)
    mx = logits.max(dim=1)[0]
    # the kernel's max may be any valid max of the row (it is the row max): compare log-sum-exp, then the max itself
    lse_ref = mx + torch.log(torch.exp(logits - mx[:, None]).sum(1))
    lse = rm.double().cpu() + torch.log(rs.double().cpu())
    assert (lse - lse_ref).abs().max() < 2e-4
    assert (rm.double().cpu() - mx).abs().max() < 1e-4
