"""GPU parity of the implicit-GEMM convolution (forward-type kernel) against torch CPU
F.conv2d / F.conv_transpose2d, for every shape class of SURVEY section 2.3 at small batch.
f32 mode must agree to f32 rounding; bf16 mode to bf16 operand rounding."""
import pytest
import torch
import torch.nn.functional as F

from msml_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _experiments():
    """The measured-slower kernel variants are compiled into experiment builds only
    (tools/build_variant.py --all MSML_EXPERIMENTS, selected with MSML_LIB; tools/experiments_run.sh runs their tests)."""
    return bool(_lib.value("msml_has_experiments"))


def needs_experiments():
    if not _experiments():
        pytest.skip("kernel variant of an experiment build (tools/experiments_run.sh; profiles/r06_experiments_variant.log)")

# (Cin1, Cin2, Cout, H, R, S, stride, pad_h, pad_w, bias)
SHAPES = [
    (3, 0, 64, 28, 3, 3, 1, 1, 1, False),      # FRB stem (C=3 padded to 8)
    (3, 0, 64, 28, 3, 3, 2, 1, 1, False),      # OSB stem (stride 2)
    (64, 0, 64, 14, 3, 3, 1, 1, 1, False),
    (64, 0, 64, 14, 3, 3, 2, 1, 1, False),
    (64, 0, 128, 14, 3, 3, 1, 1, 1, False),
    (128, 0, 128, 7, 3, 3, 2, 1, 1, False),     # odd size, stride 2
    (256, 0, 256, 14, 3, 3, 1, 1, 1, False),
    (512, 0, 512, 7, 3, 3, 1, 1, 1, False),
    (64, 18, 64, 14, 3, 3, 1, 1, 1, False),     # FM same_conv on cat(yf, yo)
    (128, 18, 128, 7, 3, 3, 1, 1, 1, False),
    (64, 0, 128, 14, 1, 1, 2, 0, 0, False),     # downsample 1x1 s2
    (128, 0, 64, 7, 1, 1, 1, 0, 0, False),      # bottleneck 1x1
    (64, 0, 18, 14, 7, 1, 1, 3, 0, True),       # GCM 7x1 with bias
    (18, 0, 18, 14, 1, 7, 1, 0, 3, True),       # GCM 1x7 with bias
    (512, 0, 8, 4, 1, 7, 1, 0, 3, True),
    (512, 0, 512, 7, 7, 7, 1, 0, 0, True),      # fc as a 7x7 valid window (Linear 25088->512)
]


def run_conv(x1, x2, w, bias, stride, ph, pw, dtype):
    xs = [ops.to_nhwc(x1.cuda(), dtype)]
    if x2 is not None:
        xs.append(ops.to_nhwc(x2.cuda(), dtype))
    cout, cin, r, s = w.shape
    wp = ops.pack_weight(w.cuda(), False, x1.shape[1], 0 if x2 is None else x2.shape[1], dtype)
    bp = None
    if bias is not None:
        bp = torch.zeros(ops.cpad(cout), device="cuda")
        bp[:cout] = bias.cuda()
    out, stats = ops.conv2d(xs[0], xs[1] if x2 is not None else None, wp, bp, ops.cpad(cout), r, s,
                            stride, ph, pw, False, want_stats=True)
    return out, stats


@pytest.mark.parametrize("dtype,tol", [(_lib.F32, 2e-6), (_lib.BF16, 1.5e-2)])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_fwd(shape, dtype, tol):
    c1, c2, cout, h, r, s, stride, ph, pw, has_bias = shape
    g = torch.Generator().manual_seed(hash(shape) & 0xffff)
    n = 3
    x1 = torch.randn(n, c1, h, h, generator=g)
    x2 = torch.randn(n, c2, h, h, generator=g) if c2 else None
    w = torch.randn(cout, c1 + c2, r, s, generator=g) * (2.0 / ((c1 + c2) * r * s)) ** 0.5
    bias = torch.randn(cout, generator=g) if has_bias else None
    if dtype == _lib.BF16:      # compare against the same bf16-rounded operands
        x1 = x1.bfloat16().float()
        x2 = x2.bfloat16().float() if c2 else None
        w = w.bfloat16().float()
    xin = x1 if x2 is None else torch.cat((x1, x2), 1)
    ref = F.conv2d(xin.double(), w.double(), None if bias is None else bias.double(), stride,
                   (ph, pw)).float()
    out, stats = run_conv(x1, x2, w, bias, stride, ph, pw, dtype)
    got = ops.to_nchw(out, cout).cpu()
    scale = ref.abs().max().item()
    if dtype == _lib.F32:       # f32 accumulation error grows ~ sqrt(K) * eps
        tol = 3e-7 * max(4.0, ((c1 + c2) * r * s) ** 0.5)
    assert (got - ref).abs().max().item() <= tol * scale, (got - ref).abs().max().item() / scale
    assert (out[..., cout:] == 0).all()                       # pad channels stay exact zeros
    # epilogue statistics: per-channel sum / sum of squares over all pixels
    ssum = stats.sum(0).float().cpu()      # partial rows (f32) or the f64 accumulator of ops.ACC_STATS
    assert torch.allclose(ssum[0, :cout], ref.sum((0, 2, 3)), rtol=0, atol=tol * scale * ref[:, 0].numel() ** 0.5 + 1e-3)
    assert torch.allclose(ssum[1, :cout], (ref * ref).sum((0, 2, 3)), rtol=max(tol * 4, 1e-4), atol=1e-3)


# ConvTranspose2d forward (transposed gather): unet.py:145-156
DECONVS = [
    (8, 0, 18, 4, 3, 2, 1),        # deconv1: 8 -> 18, k3 s2 p1, 4x4 -> 7x7
    (18, 18, 18, 7, 4, 2, 1),      # deconv2..5 on cat(seg, gcm): 36 -> 18, k4 s2 p1
    (18, 18, 18, 14, 4, 2, 1),     # from here: the dedicated kernel (conv_d4.hip) in bf16
    (18, 18, 18, 28, 4, 2, 1),
    (18, 18, 18, 56, 4, 2, 1),
]


@pytest.mark.parametrize("dtype,tol", [(_lib.F32, 2e-6), (_lib.BF16, 1.5e-2)])
@pytest.mark.parametrize("shape", DECONVS)
def test_deconv_fwd(shape, dtype, tol):
    c1, c2, cout, h, k, stride, pad = shape
    g = torch.Generator().manual_seed(7)
    n = 2
    x1 = torch.randn(n, c1, h, h, generator=g)
    x2 = torch.randn(n, c2, h, h, generator=g) if c2 else None
    wt = torch.randn(c1 + c2, cout, k, k, generator=g) * 0.2      # ConvTranspose layout (Cin, Cout, k, k)
    if dtype == _lib.BF16:
        x1 = x1.bfloat16().float()
        x2 = x2.bfloat16().float() if c2 else None
        wt = wt.bfloat16().float()
    xin = x1 if x2 is None else torch.cat((x1, x2), 1)
    ref = F.conv_transpose2d(xin.double(), wt.double(), None, stride, pad).float()
    xs = [ops.to_nhwc(x1.cuda(), dtype)] + ([ops.to_nhwc(x2.cuda(), dtype)] if c2 else [])
    wp = ops.pack_weight(wt.cuda(), True, c1, c2, dtype)
    out, _ = ops.conv2d(xs[0], xs[1] if c2 else None, wp, None, ops.cpad(cout), k, k, stride, pad,
                        pad, True)
    got = ops.to_nchw(out, cout).cpu()
    assert got.shape == ref.shape
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= tol * scale


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("shape", [(64, 128, 14), (128, 64, 9), (256, 256, 7), (32, 32, 12)])
def test_conv_dgrad_bf16(stride, shape):
    """bf16 backward-data (fast path, transposed gather) against f64 autograd on the same
    bf16-rounded operands."""
    cin, cout, h = shape
    g = torch.Generator().manual_seed(11)
    n = 3
    x = torch.randn(n, cin, h, h, generator=g, dtype=torch.double, requires_grad=True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).bfloat16().double()
    y = F.conv2d(x, w, None, stride, 1)
    dy = torch.randn(y.shape, generator=g).bfloat16().double()
    y.backward(dy)
    wp = ops.pack_weight(w.float().cuda(), True, cout, 0, _lib.BF16)
    dyd = ops.to_nhwc(dy.float().cuda(), _lib.BF16)
    dx, _ = ops.conv2d(dyd, None, wp, None, cin, 3, 3, stride, 1, 1, True, p=h, q=h)
    got = ops.to_nchw(dx, cin).cpu().double()
    assert (got - x.grad).abs().max().item() <= 1e-2 * x.grad.abs().max().item()


@pytest.mark.parametrize("stride", [1, 2])
def test_conv_dgrad_matches_autograd(stride):
    """conv backward-data == transposed gather with the (ko=cin, ci=cout) packing."""
    g = torch.Generator().manual_seed(9)
    n, cin, cout, h = 2, 64, 128, 14
    x = torch.randn(n, cin, h, h, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    y = F.conv2d(x, w, None, stride, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wp = ops.pack_weight(w.cuda(), True, cout, 0, _lib.F32)
    dyd = ops.to_nhwc(dy.cuda(), _lib.F32)
    dx, _ = ops.conv2d(dyd, None, wp, None, cin, 3, 3, stride, 1, 1, True, p=h, q=h)
    got = ops.to_nchw(dx, cin).cpu()
    assert (got - x.grad).abs().max().item() <= 2e-6 * x.grad.abs().max().item() * 4


WGRADS = [
    # (cin1, cin2, cout, h, r, s, stride, ph, pw)
    (64, 0, 64, 14, 3, 3, 1, 1, 1),
    (64, 0, 128, 14, 3, 3, 2, 1, 1),
    (128, 18, 128, 7, 3, 3, 1, 1, 1),     # two-segment input: one call per segment
    (3, 0, 64, 28, 3, 3, 1, 1, 1),
    (256, 0, 256, 14, 3, 3, 1, 1, 1),
    (64, 0, 128, 14, 1, 1, 2, 0, 0),
    (64, 0, 18, 14, 7, 1, 1, 3, 0),
    (512, 0, 512, 7, 7, 7, 1, 0, 0),      # fc
]


@pytest.mark.parametrize("dtype,tol", [(_lib.F32, 1e-5), (_lib.BF16, 1.5e-2)])
@pytest.mark.parametrize("shape", WGRADS)
def test_conv_wgrad(shape, dtype, tol):
    c1, c2, cout, h, r, s, stride, ph, pw = shape
    g = torch.Generator().manual_seed(21)
    n = 3
    x = torch.randn(n, c1 + c2, h, h, generator=g)
    p = (h + 2 * ph - r) // stride + 1
    q = (h + 2 * pw - s) // stride + 1
    dy = torch.randn(n, cout, p, q, generator=g)
    if dtype == _lib.BF16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    w = torch.zeros(cout, c1 + c2, r, s, dtype=torch.double, requires_grad=True)
    F.conv2d(x.double(), w, None, stride, (ph, pw)).backward(dy.double())
    ref = w.grad.float()
    dyd = ops.to_nhwc(dy.cuda(), dtype)
    dw = torch.full((cout, c1 + c2, r, s), 3.0, device="cuda")
    ops.conv_wgrad(dyd, ops.to_nhwc(x[:, :c1].cuda(), dtype), dw, cout, c1, c1 + c2, 0, r, s, stride,
                   ph, pw)
    if c2:
        ops.conv_wgrad(dyd, ops.to_nhwc(x[:, c1:].cuda(), dtype), dw, cout, c2, c1 + c2, c1, r, s,
                       stride, ph, pw)
    scale = ref.abs().max().item()
    assert (dw.cpu() - ref).abs().max().item() <= tol * scale


def test_deconv_wgrad():
    """ConvTranspose2d weight gradient: u = x (natural grid), v = dy (shifted), roles swapped."""
    g = torch.Generator().manual_seed(23)
    n, cin, cout, h, k = 2, 36, 18, 7, 4
    x = torch.randn(n, cin, h, h, generator=g)
    wt = torch.zeros(cin, cout, k, k, dtype=torch.double, requires_grad=True)
    y = F.conv_transpose2d(x.double(), wt, None, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    ref = wt.grad.float()
    dw = torch.zeros(cin, cout, k, k, device="cuda")
    xd = ops.to_nhwc(x.cuda(), _lib.F32)
    ops.conv_wgrad(xd, ops.to_nhwc(dy.cuda(), _lib.F32), dw, cin, cout, cout, 0, k, k, 2, 1, 1)
    assert (dw.cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("h", [14, 9])
def test_conv1x1_s2_dgrad_bf16(h):
    """Downsample 1x1 stride-2 backward-data: three of the four output parity classes receive
    no tap at all and must come out as exact zeros."""
    g = torch.Generator().manual_seed(13)
    n, cin, cout = 2, 64, 128
    x = torch.randn(n, cin, h, h, generator=g, dtype=torch.double, requires_grad=True)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).bfloat16().double()
    y = F.conv2d(x, w, None, 2, 0)
    dy = torch.randn(y.shape, generator=g).bfloat16().double()
    y.backward(dy)
    wp = ops.pack_weight(w.float().cuda(), True, cout, 0, _lib.BF16)
    dyd = ops.to_nhwc(dy.float().cuda(), _lib.BF16)
    dx, _ = ops.conv2d(dyd, None, wp, None, cin, 1, 1, 2, 0, 0, True, p=h, q=h)
    got = ops.to_nchw(dx, cin).cpu().double()
    assert (got - x.grad).abs().max().item() <= 1e-2 * x.grad.abs().max().item()
    assert (got[:, :, 1::2, :] == 0).all() and (got[:, :, :, 1::2] == 0).all()


@pytest.mark.parametrize("res_first", [False, True])
@pytest.mark.parametrize("shape", [(64, 0, 64, 14, 3, 1, 1), (64, 0, 128, 14, 3, 2, 1), (64, 18, 64, 9, 3, 1, 1),
                                   (128, 0, 64, 7, 1, 1, 0), (3, 0, 64, 28, 3, 1, 1)])
def test_conv_fused_eval_bn(shape, res_first):
    """msml_conv2d_fused (inference epilogue: eval BatchNorm + PReLU + residual in both orders)
    against torch CPU conv2d -> affine -> prelu / residual in f32."""
    c1, c2, cout, h, k, stride, pad = shape
    torch.manual_seed(5)
    bf = lambda t: t.to(torch.bfloat16).float()
    x1 = bf(torch.randn(3, c1, h, h))
    x2 = bf(torch.randn(3, c2, h, h)) if c2 else None
    w = bf(torch.randn(cout, c1 + c2, k, k) * (1.0 / ((c1 + c2) * k * k)) ** 0.5)
    scale = torch.rand(cout) + 0.5
    shift = torch.randn(cout) * 0.3
    alpha = torch.rand(cout) * 0.5
    ref = F.conv2d(torch.cat([x1, x2], 1) if c2 else x1, w, None, stride, pad)
    res = bf(torch.randn_like(ref))
    z = ref * scale[None, :, None, None] + shift[None, :, None, None]
    if res_first:
        want = F.prelu(z + res, alpha)
    else:
        want = F.prelu(z, alpha) + res
    xs = [ops.to_nhwc(x1.cuda(), _lib.BF16)] + ([ops.to_nhwc(x2.cuda(), _lib.BF16)] if c2 else [])
    wp = ops.pack_weight(w.cuda(), False, c1, c2, _lib.BF16)
    p = ref.shape[2]
    out = torch.empty(3, p, p, cout, dtype=torch.bfloat16, device="cuda")
    rn = ops.to_nhwc(res.cuda(), _lib.BF16)
    _lib.call("msml_conv2d_fused", xs[0], xs[0].shape[3], xs[1] if c2 else None, xs[1].shape[3] if c2 else 0,
              wp, wp.shape[0], scale.cuda(), shift.cuda(), alpha.cuda(), rn, int(res_first), out, cout,
              3, h, h, p, p, k, k, stride, pad, pad, 0)
    got = ops.to_nchw(out, cout).float().cpu()
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err < 1.5e-2, err
    # no residual / no activation: plain affine epilogue
    _lib.call("msml_conv2d_fused", xs[0], xs[0].shape[3], xs[1] if c2 else None, xs[1].shape[3] if c2 else 0,
              wp, wp.shape[0], scale.cuda(), shift.cuda(), None, None, 0, out, cout,
              3, h, h, p, p, k, k, stride, pad, pad, 0)
    got = ops.to_nchw(out, cout).float().cpu()
    err = (got - z).abs().max().item() / z.abs().max().item()
    assert err < 1e-2, err


# (N, Cin, Cout, H, W): shapes routed to the halo-tile 3x3 kernel (conv_halo.hip): W <= 15,
# Cin % 64 == 0, Cout % 256 == 0, >= 128 row strips
HALO = [   # (Cin = 64 stays on the im2col kernel: the shapes below with 64 input channels test that fallback too)
    (128, 64, 256, 14, 14),      # one 64-channel slab
    (128, 192, 256, 14, 14),     # three slabs: image double buffer wraps
    (130, 64, 256, 7, 7),        # too few real rows per strip: stays on the im2col kernel
    (128, 64, 512, 13, 13),      # two channel tiles, ragged width
    (70, 64, 256, 26, 14),       # two strips per image, the second one 12 rows
    (40, 64, 256, 28, 28),       # 2 x 2 tiles per image: interior halos are real pixels
    (40, 128, 128, 28, 28),      # 128 output channels: 4 channel groups x 2 pixel-row groups
    (24, 64, 128, 28, 40),       # ragged last tile column (12 px), 6 tiles per image
    (10, 64, 384, 56, 56),       # 128-channel variant with 3 channel tiles, 16 tiles per image
    (20, 64, 64, 56, 56),        # weights-stationary persistent kernel (conv_ws.hip): 320 tiles > 256 CUs
    (5, 64, 64, 28, 40),         # ... ragged tile column, fewer tiles than CUs
]


@pytest.mark.parametrize("shape", HALO)
def test_conv_halo_fwd_dgrad_stats(shape):
    n, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w_, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).bfloat16().float()
    ref = F.conv2d(x, w, None, 1, 1)
    out, stats = run_conv(x, None, w, None, 1, 1, 1, _lib.BF16)
    got = ops.to_nchw(out, cout).cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 1.5e-2 * scale
    ssum = stats.double().sum(0).cpu()
    assert torch.allclose(ssum[0].float(), ref.double().sum((0, 2, 3)).float(), rtol=0,
                          atol=1.5e-2 * scale * ref[:, 0].numel() ** 0.5)
    assert torch.allclose(ssum[1].float(), (ref.double() ** 2).sum((0, 2, 3)).float(), rtol=2e-2)
    # backward-data: dX = transposed gather of dY with the (ko = cin) packing; cin must be a
    # multiple of 256 to take the halo kernel, so use the roles swapped: dY has `cin` channels
    dy = x                                                   # [n, cin, h, w]
    wd = (torch.randn(cin, cout, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).bfloat16().float()
    dref = torch.nn.grad.conv2d_input((n, cout, h, w_), wd, dy, 1, 1)      # conv: cout -> cin, weight [cin][cout]
    wp = ops.pack_weight(wd.cuda(), True, cin, 0, _lib.BF16)
    dx, _ = ops.conv2d(ops.to_nhwc(dy.cuda(), _lib.BF16), None, wp, None, cout, 3, 3, 1, 1, 1, True, p=h, q=w_)
    gd = ops.to_nchw(dx, cout).cpu()
    assert (gd - dref).abs().max().item() <= 1.5e-2 * dref.abs().max().item()


def test_conv_halo_fused_epilogue():
    n, cin, cout, h = 128, 64, 256, 14
    torch.manual_seed(3)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(n, cin, h, h))
    w = bf(torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9)) ** 0.5)
    scale, shift, alpha = torch.rand(cout) + 0.5, torch.randn(cout) * 0.3, torch.rand(cout) * 0.5
    ref = F.conv2d(x, w, None, 1, 1)
    res = bf(torch.randn_like(ref))
    z = ref * scale[None, :, None, None] + shift[None, :, None, None]
    xn, rn = ops.to_nhwc(x.cuda(), _lib.BF16), ops.to_nhwc(res.cuda(), _lib.BF16)
    wp = ops.pack_weight(w.cuda(), False, cin, 0, _lib.BF16)
    out = torch.empty(n, h, h, cout, dtype=torch.bfloat16, device="cuda")
    for res_first, want in ((0, F.prelu(z, alpha) + res), (1, F.prelu(z + res, alpha))):
        _lib.call("msml_conv2d_fused", xn, cin, None, 0, wp, wp.shape[0], scale.cuda(), shift.cuda(),
                  alpha.cuda(), rn, res_first, out, cout, n, h, h, h, h, 3, 3, 1, 1, 1, 0)
        got = ops.to_nchw(out, cout).float().cpu()
        assert (got - want).abs().max().item() <= 1.5e-2 * want.abs().max().item()


def _bn_bwd_unfused(dy, x, coef, alpha, m, c):
    dx = torch.empty_like(x)
    pg = torch.zeros(3, c, device="cuda")
    rows = ops.bn_stats_rows(m, c)
    ws = torch.empty(rows * 3 * c + 2 * c, device="cuda")
    _lib.call("msml_bn_act_bwd", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], None, dx, None, pg[0], pg[1],
              pg[2] if alpha is not None else None, 0, m, c, ws, ws.numel(), _lib.BF16)
    return dx, pg


# (N, K = dy channels, C = dX / BatchNorm channels, H, stride): halo kernel, im2col kernel with
# full and ragged tiles, stride-2 parity classes (odd size: unequal classes)
BNBWD = [(128, 64, 256, 14, 1), (40, 128, 128, 28, 1), (3, 64, 64, 14, 1), (20, 64, 64, 56, 1), (5, 128, 64, 9, 1), (4, 64, 128, 14, 2), (3, 128, 64, 9, 2)]


@pytest.mark.parametrize("acc_mode", [True, False])
@pytest.mark.parametrize("with_alpha", [False, True])
@pytest.mark.parametrize("shape", BNBWD)
def test_conv_dgrad_fused_bn_backward_reduce(shape, with_alpha, acc_mode, monkeypatch):
    """msml_conv2d_bnbwd + msml_bn_act_bwd_apply == msml_conv2d (dgrad) + msml_bn_act_bwd, and the
    `add` operand of the apply step is summed into dx.  acc_mode: the same through the f64 accumulator protocol
    (msml_conv2d_bnbwd_acc + msml_bn_fin_bwd_apply: no finalize launch)."""
    monkeypatch.setattr(ops, "ACC_STATS", acc_mode)
    n, k, c, h, stride = shape
    g = torch.Generator().manual_seed(sum(shape))
    ho = (h + 2 - 3) // stride + 1
    dyc = torch.randn(n, k, ho, ho, generator=g).bfloat16().float()          # gradient of the conv output
    w = (torch.randn(k, c, 3, 3, generator=g) * 0.05).bfloat16().float()      # conv: c -> k
    xbn = ops.to_nhwc(torch.randn(n, c, h, h, generator=g).cuda(), _lib.BF16)   # saved BatchNorm input
    coef = torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3,
                        torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5]).cuda()
    alpha = (torch.rand(c, generator=g) * 0.5).cuda() if with_alpha else None
    wp = ops.pack_weight(w.cuda(), True, k, 0, _lib.BF16)
    dyd = ops.to_nhwc(dyc.cuda(), _lib.BF16)
    m = n * h * h
    dx_ref, _ = ops.conv2d(dyd, None, wp, None, c, 3, 3, stride, 1, 1, True, p=h, q=h)
    want_dx, want_pg = _bn_bwd_unfused(dx_ref, xbn, coef, alpha, m, c)
    got = ops.conv_dgrad_bnbwd(dyd, wp, c, 3, 3, stride, 1, 1, h, h, xbn, coef, alpha)
    assert got is not None
    dxc, partial = got
    assert torch.equal(dxc, dx_ref)
    add = ops.to_nhwc(torch.randn(n, c, h, h, generator=g).cuda(), _lib.BF16)
    for addt in (None, add):
        dx = torch.empty_like(xbn)
        pg = torch.zeros(3, c, device="cuda")
        cw = torch.empty(98 * c, device="cuda")
        if acc_mode:
            assert partial.dtype == torch.float64 and partial.shape == (8, 3, c)
            _lib.call("msml_bn_fin_bwd_apply", dxc, xbn, coef[0], coef[1], alpha, coef[2], coef[3], partial, None, addt, 0, 0,
                      dx, None, pg[0], pg[1], pg[2] if with_alpha else None, 0, m, c, None, None, None, None, _lib.BF16)
        else:
            _lib.call("msml_bn_act_bwd_apply", dxc, xbn, coef[0], coef[1], alpha, coef[2], coef[3], partial,
                      partial.shape[0], addt, dx, pg[0], pg[1], pg[2] if with_alpha else None, 0, m, c, cw, _lib.BF16)
        ref = want_dx.float() + (addt.float() if addt is not None else 0)
        scale = ref.abs().max().item()
        assert (dx.float() - ref).abs().max().item() <= 1e-2 * scale
        for i in range(3 if with_alpha else 2):
            assert torch.allclose(pg[i], want_pg[i], rtol=1e-3, atol=1e-3 * want_pg[i].abs().max().item())


# (N, Cin, Cout, H, W): weight gradients routed to the strip / halo kernel (wgrad_halo.hip):
# Cout % 128 == 0, Cin % 64 == 0, 3x3 s1 p1
# (7 x 7 maps: two images per strip, side by side on the 16-pixel pitch -- odd batches leave the last strip half empty)
WGRAD_HALO = [(6, 64, 64, 14, 14), (9, 128, 64, 28, 28), (6, 64, 128, 14, 14), (5, 128, 128, 21, 28), (4, 128, 256, 13, 27), (40, 64, 128, 28, 28),
              (7, 128, 128, 7, 7), (10, 64, 128, 7, 7), (33, 256, 64, 7, 7), (1, 64, 64, 7, 7)]


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("shape", WGRAD_HALO)
def test_conv_wgrad_halo(shape, accumulate):
    n, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w_, generator=g).bfloat16().float()
    dy = torch.randn(n, cout, h, w_, generator=g).bfloat16().float()
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(dy.double())
    ref = w.grad.float()
    dw = torch.full((cout, cin, 3, 3), 2.0, device="cuda")
    ops.conv_wgrad(ops.to_nhwc(dy.cuda(), _lib.BF16), ops.to_nhwc(x.cuda(), _lib.BF16), dw, cout, cin, cin, 0,
                   3, 3, 1, 1, 1, accumulate=accumulate)
    want = ref + (2.0 if accumulate else 0.0)
    assert (dw.cpu() - want).abs().max().item() <= 1.5e-2 * ref.abs().max().item()


@pytest.mark.parametrize("group", [2, 3, 4, 8])
@pytest.mark.parametrize("shape", [(16, 256, 256, 14, 14), (12, 128, 128, 28, 28), (6, 64, 64, 56, 56), (9, 128, 64, 21, 28),
                                   (256, 512, 512, 7, 7), (200, 256, 256, 7, 7), (256, 128, 128, 7, 7), (256, 512, 512, 4, 4)])
def test_conv_wgrad_group(shape, group):
    """msml_conv_wgrad_group: `group` same-shape layers in one launch pair (each layer gets 1 / group of the
    workgroups, XCD-aware order for the many-tile shapes; the small maps run on the im2col kernel, down to one split =
    tiles written straight into dW) -- every layer's gradient against f64 torch, accumulated onto existing values,
    through the raw entry point and through the host-side queue."""
    import ctypes
    n, cin, cout, h, w_ = shape
    gmax = _lib.value("msml_conv_wgrad_group_max", cout, cin, cout, cin, n, h, w_, h, w_, 3, 3, 1, 1, 1)
    if group > gmax:
        pytest.skip("shape groups at most %d layers" % gmax)
    g = torch.Generator().manual_seed(sum(shape) + group)
    xs = [torch.randn(n, cin, h, w_, generator=g).bfloat16().float() for _ in range(group)]
    dys = [torch.randn(n, cout, h, w_, generator=g).bfloat16().float() for _ in range(group)]
    refs = []
    for x, dy in zip(xs, dys):
        w = torch.zeros(cout, cin, 3, 3, dtype=torch.double, requires_grad=True)
        F.conv2d(x.double(), w, None, 1, 1).backward(dy.double())
        refs.append(w.grad.float())
    us = [ops.to_nhwc(dy.cuda(), _lib.BF16) for dy in dys]
    vs = [ops.to_nhwc(x.cuda(), _lib.BF16) for x in xs]
    dws = [torch.full((cout, cin, 3, 3), 1.0 + i, device="cuda") for i in range(group)]
    need = _lib.value("msml_conv_wgrad_workspace", cout, cin, n, h, w_, 3, 3)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    arr = ctypes.c_void_p * group
    _lib.call("msml_conv_wgrad_group", arr(*[t.data_ptr() for t in us]), arr(*[t.data_ptr() for t in vs]),
              arr(*[t.data_ptr() for t in dws]), group, cout, cin, cout, cin, cin, 0, n, h, w_, h, w_, 3, 3, 1, 1, 1, 1,
              ws, ws.numel(), _lib.BF16)
    for i in range(group):
        assert (dws[i].cpu() - (refs[i] + 1.0 + i)).abs().max().item() <= 1.5e-2 * refs[i].abs().max().item(), i
    # the host-side queue: `group` queued layers -> one grouped launch; a shorter tail is issued by wgrad_flush()
    old = ops.WGRAD_GROUP
    ops.WGRAD_GROUP, ops._GROUP_MAX = group, {}
    try:
        dq = [torch.zeros(cout, cin, 3, 3, device="cuda") for _ in range(group + 1)]
        # (outside a backward pass there is no engine callback to register: fill the queue directly)
        for i in range(group):
            ops._WQ.items.append((us[i], vs[i], dq[i], None))
        ops._WQ.key = (cout, cin, n, h, w_, cout, cin, cin, 0, _lib.raw_stream())
        ops._WQ.stream = None
        ops.wgrad_flush()
        assert not ops._WQ.items
        for i in range(group):
            assert (dq[i].cpu() - refs[i]).abs().max().item() <= 1.5e-2 * refs[i].abs().max().item(), i
            assert torch.equal(dq[i] + (1.0 + i), dws[i]) or (dq[i] + (1.0 + i) - dws[i]).abs().max().item() <= 1e-5 * refs[i].abs().max().item()
    finally:
        ops.WGRAD_GROUP, ops._GROUP_MAX = old, {}


# (N, Cin, Cout, H, W): BatchNorm(+PReLU) applied to the conv input inside the halo-tile kernels
# (msml_conv2d_bnin / msml_conv_wgrad_bnin): must equal msml_bn_act_fwd -> msml_conv2d / msml_conv_wgrad
# bit for bit (same rounding of the normalised activation, same MFMA order)
BNIN = [
    (22, 64, 64, 56, 56),        # weights-stationary kernel, more tiles than one pass (it needs
    (60, 64, 64, 28, 40),        # ... ragged tile column     >= as many statistics rows as workgroups)
    (9, 128, 128, 28, 28),       # halo kernel, 128 output channels, two slabs
    (6, 128, 256, 28, 28),       # halo kernel, 256 output channels
    (7, 256, 256, 14, 14),       # four slabs: the image double buffer wraps twice
    (4, 256, 128, 13, 27),       # ragged tiles (zero padding must stay zero)
]


@pytest.mark.parametrize("with_alpha", [False, True])
@pytest.mark.parametrize("shape", BNIN)
def test_conv_bn_in_lds_matches_unfused(shape, with_alpha):
    n, cin, cout, h, w_ = shape
    if cin == 64:                # (the weights-stationary kernel takes an input transform in experiment builds only)
        needs_experiments()
    g = torch.Generator().manual_seed(sum(shape) + int(with_alpha))
    x = ops.to_nhwc(torch.randn(n, cin, h, w_, generator=g).cuda(), _lib.BF16)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).cuda()
    coef = torch.stack([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5]).cuda()
    alpha = (torch.rand(cin, generator=g) * 0.3).cuda() if with_alpha else None
    assert _lib.value("msml_conv2d_bnin_applies", cin, cout, n, h, w_, h, w_, 3, 3, 1, 1, 1, 1) == 1
    act = torch.empty_like(x)
    _lib.call("msml_bn_act_fwd", x, coef[0], coef[1], alpha, None, 0, act, n * h * w_, cin, _lib.BF16)
    wp = ops.pack_weight(w, False, cin, 0, _lib.BF16)
    ref, rstats = ops.conv2d(act, None, wp, None, cout, 3, 3, 1, 1, 1, False, want_stats=True)
    got, gstats = ops.conv2d_bnin(x, coef, alpha, wp, cout)
    assert torch.equal(got, ref)
    # (partial rows from the bnin kernel, f64 accumulator from the plain conv under ops.ACC_STATS: the same totals up
    # to the order of the per-workgroup f32 sums -- the plain conv runs the 16x16x32 MFMA tiling since round 4, the
    # bnin instantiation the 32x32x16 one: identical outputs, another lane-to-pixel map in the statistics epilogue)
    assert torch.allclose(gstats.double().sum(0), rstats.double().sum(0), rtol=1e-5, atol=1e-4)
    # weight gradient
    dy = ops.to_nhwc(torch.randn(n, cout, h, w_, generator=g).cuda(), _lib.BF16)
    assert _lib.value("msml_conv_wgrad_bnin_applies", cout, cin, cout, cin, n, h, w_, h, w_, 3, 3, 1, 1, 1) == 1
    dref = torch.full((cout, cin, 3, 3), 0.5, device="cuda")
    dgot = dref.clone()
    ops.conv_wgrad(dy, act, dref, cout, cin, cin, 0, 3, 3, 1, 1, 1, accumulate=True)
    ops.conv_wgrad_bnin(dy, x, coef, alpha, dgot, cout, cin, cin, 0, accumulate=True)
    assert torch.equal(dgot, dref)


def test_conv_bn_in_lds_refuses_other_shapes():
    # stride 2, 1x1, 7x7 maps (too few real rows per tile), 32 channels: not on the halo-tile kernels
    for args in [(64, 64, 4, 56, 56, 28, 28, 3, 3, 2, 1, 1, 1), (64, 64, 4, 56, 56, 56, 56, 1, 1, 1, 0, 0, 1),
                 (512, 512, 4, 7, 7, 7, 7, 3, 3, 1, 1, 1, 1), (32, 32, 4, 56, 56, 56, 56, 3, 3, 1, 1, 1, 1)]:
        assert _lib.value("msml_conv2d_bnin_applies", *args) == 0
    x = torch.zeros(4, 7, 7, 512, dtype=torch.bfloat16, device="cuda")
    coef = torch.ones(2, 512, device="cuda")
    wp = torch.zeros(512, 9 * 512, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(4, 7, 7, 512, dtype=torch.bfloat16, device="cuda")
    rc = _lib.try_call("msml_conv2d_bnin", x, 512, coef[0], coef[1], None, wp, 512, out, 512, None, 4, 7, 7, 7, 7,
                       3, 3, 1, 1, 1)
    assert rc == _lib.UNSUPPORTED


# Narrow-operand weight-gradient kernel (wgrad_n32.hip): (N, P, Q, kind) with kind 'd4' = 4x4 / stride-2
# transposed conv on 18-channel maps (OSB deconv2..5: u = x on P x Q, v = dy on 2P x 2Q) and 'c3' = 3x3 /
# stride-1 conv with 32 -> 32 channels (FM bottleneck).  Odd sizes exercise the strip / plane edges.
WGRAD_N32 = [(3, 14, 14, "d4"), (2, 28, 28, "d4"), (2, 15, 17, "d4"), (5, 56, 56, "d4"),
             (3, 14, 14, "c3"), (2, 56, 56, "c3"), (2, 19, 33, "c3")]


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("shape", WGRAD_N32)
def test_conv_wgrad_n32(shape, accumulate):
    n, p, q, kind = shape
    g = torch.Generator().manual_seed(31 + p + q)
    if kind == "d4":
        cin, cout, k = 18, 18, 4
        x = torch.randn(n, cin, p, q, generator=g).bfloat16().float()
        wt = torch.zeros(cin, cout, k, k, dtype=torch.double, requires_grad=True)
        y = F.conv_transpose2d(x.double(), wt, None, 2, 1)
        dy = torch.randn(y.shape, generator=g).bfloat16().float()
        y.backward(dy.double())
        assert _lib.value("msml_conv_wgrad_kernel_is_n32", 32, 32, n, 2 * p, 2 * q, p, q, 4, 4, 2, 1, 1) == 1
        dw = torch.full((cin, cout, k, k), 2.0, device="cuda")
        ops.conv_wgrad(ops.to_nhwc(x.cuda(), _lib.BF16), ops.to_nhwc(dy.cuda(), _lib.BF16), dw, cin, cout, cout, 0,
                       k, k, 2, 1, 1, accumulate=accumulate)
        ref = wt.grad.float()
    else:
        cin, cout = 32, 32
        x = torch.randn(n, cin, p, q, generator=g).bfloat16().float()
        w = torch.zeros(cout, cin, 3, 3, dtype=torch.double, requires_grad=True)
        y = F.conv2d(x.double(), w, None, 1, 1)
        dy = torch.randn(y.shape, generator=g).bfloat16().float()
        y.backward(dy.double())
        assert _lib.value("msml_conv_wgrad_kernel_is_n32", 32, 32, n, p, q, p, q, 3, 3, 1, 1, 1) == 1
        dw = torch.full((cout, cin, 3, 3), 2.0, device="cuda")
        ops.conv_wgrad(ops.to_nhwc(dy.cuda(), _lib.BF16), ops.to_nhwc(x.cuda(), _lib.BF16), dw, cout, cin, cin, 0,
                       3, 3, 1, 1, 1, accumulate=accumulate)
        ref = w.grad.float()
    if accumulate:
        ref = ref + 2.0
    assert (dw.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()     # bf16 operands, f32 sums


# single-stage launches (K <= 64: one LDS stage is allocated, conv_fast.hip launch_fast) in every tile shape
@pytest.mark.parametrize("shape", [(32, 64, 20, 1), (64, 32, 20, 1), (64, 64, 30, 2), (64, 128, 14, 1), (32, 18, 12, 1),
                                   (64, 256, 9, 1)])
def test_conv1x1_single_stage(shape):
    cin, cout, h, stride = shape
    g = torch.Generator().manual_seed(cin + cout + h)
    n = 5
    x = torch.randn(n, cin, h, h, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), None, stride, 0).float()
    out, stats = run_conv(x, None, w, None, stride, 0, 0, _lib.BF16)
    got = ops.to_nchw(out, cout).cpu()
    assert (got - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    s = stats.sum(0).float().cpu()
    assert torch.allclose(s[0][:cout], ref.sum((0, 2, 3)), rtol=2e-2, atol=0.05 * ref.abs().max().item() * n)
    # backward-data of the same layer (transposed gather, K = cout)
    if stride == 1 and cout <= 64:
        dy = torch.randn(n, cout, h, h, generator=g).bfloat16().float()
        wpt = ops.pack_weight(w.cuda(), True, cout, 0, _lib.BF16)
        dx, _ = ops.conv2d(ops.to_nhwc(dy.cuda(), _lib.BF16), None, wpt, None, ops.cpad(cin), 1, 1, 1, 0, 0, True, p=h, q=h)
        rdx = F.conv_transpose2d(dy.double(), w.double(), None, 1, 0).float()
        assert (ops.to_nchw(dx, cin).cpu() - rdx).abs().max().item() <= 1.5e-2 * rdx.abs().max().item()


# Line-conv kernel (conv_line.hip): GCM 7x1 / 1x7 convs with bias at the 56 / 28 levels, forward and backward-data
@pytest.mark.parametrize("shape", [(64, 18, 56, 7, 1), (64, 18, 56, 1, 7), (18, 18, 56, 7, 1), (18, 18, 56, 1, 7),
                                   (64, 18, 28, 7, 1), (18, 18, 28, 1, 7)])
def test_conv_line(shape):
    cin, cout, h, r, s = shape
    ph, pw = (r - 1) // 2, (s - 1) // 2
    g = torch.Generator().manual_seed(cin + h + r)
    n = 5
    x = torch.randn(n, cin, h, h, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, r, s, generator=g) * (2.0 / (cin * 7)) ** 0.5).bfloat16().float()
    bias = torch.randn(cout, generator=g)
    assert _lib.value("msml_conv2d_kernel", ops.cpad(cin), 0, ops.cpad(cout), n, h, h, h, h, r, s, 1, ph, pw, 0, 1, 1, 0) \
        .decode().startswith("k_conv_line")
    ref = F.conv2d(x.double(), w.double(), bias.double(), 1, (ph, pw)).float()
    wp = ops.pack_weight(w.cuda(), False, cin, 0, _lib.BF16)
    bp = torch.zeros(ops.cpad(cout), device="cuda")
    bp[:cout] = bias.cuda()
    out, _ = ops.conv2d(ops.to_nhwc(x.cuda(), _lib.BF16), None, wp, bp, ops.cpad(cout), r, s, 1, ph, pw, False)
    assert (ops.to_nchw(out, cout).cpu() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    assert out[..., cout:].abs().max().item() == 0                       # pad channels stay exact zeros
    # backward-data: transposed gather with the (ko = cin, ci = cout) packing
    dy = torch.randn(n, cout, h, h, generator=g).bfloat16().float()
    rdx = F.conv_transpose2d(dy.double(), w.double(), None, 1, (ph, pw)).float()
    wpt = ops.pack_weight(w.cuda(), True, cout, 0, _lib.BF16)
    dx, _ = ops.conv2d(ops.to_nhwc(dy.cuda(), _lib.BF16), None, wpt, None, ops.cpad(cin), r, s, 1, ph, pw, True, p=h, q=h)
    assert (ops.to_nchw(dx, cin).cpu() - rdx).abs().max().item() <= 1.5e-2 * rdx.abs().max().item()


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("shape", [(64, 18, 56, 7, 1), (64, 18, 56, 1, 7), (18, 18, 56, 7, 1), (18, 18, 28, 1, 7), (64, 18, 28, 7, 1)])
def test_conv_wgrad_line(shape, accumulate):
    """Weight gradient of the GCM line convs on the line-major kernel (wgrad_n32.hip k_wgrad_line)."""
    cin, cout, h, r, s = shape
    ph, pw = (r - 1) // 2, (s - 1) // 2
    g = torch.Generator().manual_seed(cin + h + r + 5)
    n = 4
    x = torch.randn(n, cin, h, h, generator=g).bfloat16().float()
    w = torch.zeros(cout, cin, r, s, dtype=torch.double, requires_grad=True)
    y = F.conv2d(x.double(), w, None, 1, (ph, pw))
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy.double())
    ref = w.grad.float() + (2.0 if accumulate else 0.0)
    assert _lib.value("msml_conv_wgrad_kernel_is_n32", 32, ops.cpad(cin), n, h, h, h, h, r, s, 1, ph, pw) == 1
    dw = torch.full((cout, cin, r, s), 2.0, device="cuda")
    ops.conv_wgrad(ops.to_nhwc(dy.cuda(), _lib.BF16), ops.to_nhwc(x.cuda(), _lib.BF16), dw, cout, cin, cin, 0, r, s, 1,
                   ph, pw, accumulate=accumulate)
    assert (dw.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("h", [14, 28, 56])
def test_deconv4_bwd_data_both_segments(h):
    """Input gradients of ConvTranspose2d(36 -> 18, 4, 2, 1) on cat(seg, gcm): one kernel for both segments
    (msml_deconv4_bwd_data) against autograd in f64."""
    g = torch.Generator().manual_seed(40 + h)
    n = 3
    x = torch.randn(n, 36, h, h, generator=g, dtype=torch.double, requires_grad=True)
    wt = (torch.randn(36, 18, 4, 4, generator=g) * 0.2).bfloat16().double()
    y = F.conv_transpose2d(x, wt, None, 2, 1)
    dy = torch.randn(y.shape, generator=g).bfloat16().double()
    y.backward(dy)
    wc = wt.float().cuda()
    wp0 = ops.pack_weight(wc[:18].contiguous(), False, 18, 0, _lib.BF16)
    wp1 = ops.pack_weight(wc[18:].contiguous(), False, 18, 0, _lib.BF16)
    dyd = ops.to_nhwc(dy.float().cuda(), _lib.BF16)
    g0 = torch.empty(n, h, h, 32, dtype=torch.bfloat16, device="cuda")
    g1 = torch.empty_like(g0)
    _lib.call("msml_deconv4_bwd_data", dyd, wp0, wp1, g0, g1, n, h)
    ref = x.grad.float()
    scale = ref.abs().max().item()
    assert (ops.to_nchw(g0, 18).cpu() - ref[:, :18]).abs().max().item() <= 1.5e-2 * scale
    assert (ops.to_nchw(g1, 18).cpu() - ref[:, 18:]).abs().max().item() <= 1.5e-2 * scale
    assert g0[..., 18:].abs().max().item() == 0 and g1[..., 18:].abs().max().item() == 0


@pytest.mark.parametrize("n,c,e,k,boff,btot,acc", [(256, 512, 512, 7, 0, 512, 0), (40, 64, 96, 7, 0, 64, 1),
                                                   (16, 32, 32, 4, 32, 96, 0), (33, 64, 64, 3, 0, 64, 1)])
def test_fc_wgrad_window(n, c, e, k, boff, btot, acc):
    """Weight gradient of flatten + Linear on a k x k map (the 'window' conv: R = H, S = W, one output pixel) on the
    all-taps-per-workgroup kernel (fc_wgrad.hip): against f64 torch, with a batch that is not a multiple of the
    16-image k-step, a channel offset into a wider dW (two-segment inputs) and accumulation into a live gradient."""
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, k, k, generator=g).bfloat16().float()
    dy = torch.randn(n, e, 1, 1, generator=g).bfloat16().float()
    ref = torch.einsum("ne,nchw->echw", dy[:, :, 0, 0].double(), x.double()).float()
    dw = torch.full((e, btot, k, k), 0.25, device="cuda")
    ops.conv_wgrad(ops.to_nhwc(dy.cuda(), _lib.BF16), ops.to_nhwc(x.cuda(), _lib.BF16), dw, e, c, btot, boff, k, k, 1, 0, 0,
                   accumulate=bool(acc))
    got = dw[:, boff:boff + c].cpu()
    want = ref + (0.25 if acc else 0.0)
    assert (got - want).abs().max().item() <= 2e-3 * ref.abs().max().item()
    rest = torch.cat([dw[:, :boff], dw[:, boff + c:]], 1)
    assert (rest == 0.25).all()                     # columns of other segments are left alone


@pytest.mark.parametrize("n,k,c,h,stride,fold", [(64, 64, 64, 56, 1, False), (256, 256, 256, 14, 1, False),
                                                 (64, 64, 128, 56, 1, True), (64, 128, 64, 56, 2, True)])
def test_bn_backward_row_and_accumulator_protocols_agree(n, k, c, h, stride, fold, monkeypatch):
    """The two hand-offs of the fused BatchNorm backward sums -- one partial row per workgroup + finalize launch, or f64
    atomics into 8 accumulator rows folded by the apply kernel (ops.ACC_STATS) -- carry the same numbers: the f64
    totals are EQUAL (f64 sums of f32 partials are exact), and so are dx and the parameter gradients bit for bit while
    the row protocol sums its rows directly (<= 512 rows).  Beyond that the row protocol first folds to 32 rows rounded
    to f32, which costs it ~1e-5 in the parameter gradients and a handful of 1-ulp flips in dx."""
    g = torch.Generator().manual_seed(n + c)
    ho = (h + 2 - 3) // stride + 1
    dyd = ops.to_nhwc(torch.randn(n, k, ho, ho, generator=g).cuda(), _lib.BF16)
    w = (torch.randn(k, c, 3, 3, generator=g) * 0.05).cuda()
    xbn = ops.to_nhwc(torch.randn(n, c, h, h, generator=g).cuda(), _lib.BF16)
    coef = torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3,
                        torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5]).cuda()
    alpha = (torch.rand(c, generator=g) * 0.5).cuda()
    wp = ops.pack_weight(w, True, k, 0, _lib.BF16)
    m = n * h * h
    out = {}
    for acc in (False, True):
        monkeypatch.setattr(ops, "ACC_STATS", acc)
        dxc, partial = ops.conv_dgrad_bnbwd(dyd, wp, c, 3, 3, stride, 1, 1, h, h, xbn, coef, alpha)
        dx, pg = torch.empty_like(xbn), torch.zeros(3, c, device="cuda")
        if acc:
            assert partial.dtype == torch.float64 and tuple(partial.shape) == (8, 3, c)
            _lib.call("msml_bn_fin_bwd_apply", dxc, xbn, coef[0], coef[1], alpha, coef[2], coef[3], partial, None, None, 0, 0,
                      dx, None, pg[0], pg[1], pg[2], 0, m, c, None, None, None, None, _lib.BF16)
        else:
            assert (partial.shape[0] > 512) == fold
            cw = torch.empty(98 * c, device="cuda")
            _lib.call("msml_bn_act_bwd_apply", dxc, xbn, coef[0], coef[1], alpha, coef[2], coef[3], partial,
                      partial.shape[0], None, dx, pg[0], pg[1], pg[2], 0, m, c, cw, _lib.BF16)
        out[acc] = (dxc, dx, pg, partial.double().sum(0))
    a, b = out[True], out[False]
    assert torch.equal(a[0], b[0])
    # the persistent 128-channel tile (accumulator mode only) adds the tiles of a workgroup in f32 before its f64 atomics:
    # its totals equal the row protocol's to f32 rounding, not bit for bit
    name = _lib.value("msml_conv2d_kernel", k, 0, c, n, ho, ho, h, h, 3, 3, stride, 1, 1, 1, _lib.BF16, _lib.BF16, 0).decode()
    if "k_conv_halo_p" in name:
        assert torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-5 * b[3].abs().max().item())
        fold = True
    else:
        assert torch.equal(a[3], b[3])                 # the three sums, in f64: identical totals
    if not fold:
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    else:
        assert ((a[2] - b[2]).abs() <= 1e-4 * b[2].abs() + 1e-6).all()
        assert int((a[1] != b[1]).sum()) < 1e-5 * a[1].numel()


# (N, Cin, Cout, H, R, S, stride, pad_h, pad_w, bias): small maps whose default 128- / 256-row tile would leave the
# chip half empty -- at these batch sizes (N * P * Q >= 2048) msml_conv_fast_dispatch takes the 64-row tiles
SMALL_MAPS = [
    (160, 512, 512, 4, 3, 3, 1, 1, 1, False),      # OSB layer4 (512 @ 4x4)
    (64, 256, 256, 7, 3, 3, 1, 1, 1, False),       # OSB layer3
    (64, 128, 128, 7, 3, 3, 1, 1, 1, False),       # FM stage-3 bottleneck 3x3
    (160, 512, 8, 4, 7, 1, 1, 3, 0, True),         # GCM1 7x1 (64 x 32 tile)
    (64, 256, 18, 7, 1, 7, 1, 0, 3, True),         # GCM2 1x7
    (64, 512, 128, 7, 1, 1, 1, 0, 0, False),       # bottleneck 1x1 (64 x 128 tile, one K stage per 64 channels)
    (64, 256, 256, 14, 3, 3, 2, 1, 1, False),      # stride 2: 14x14 -> 7x7
]


@pytest.mark.parametrize("shape", SMALL_MAPS)
def test_conv_small_map_tiles(shape):
    """The 64-row tiles of k_conv_fast: forward (+ accumulator-mode statistics), backward-data, and backward-data with
    the fused BatchNorm backward sums (accumulator mode), against f64 torch on the same bf16-rounded operands."""
    n, cin, cout, h, r, s, stride, ph, pw, has_bias = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, h, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, r, s, generator=g) * (2.0 / (cin * r * s)) ** 0.5).bfloat16().float()
    bias = torch.randn(cout, generator=g) if has_bias else None
    xd = x.double().requires_grad_(True)
    ref = F.conv2d(xd, w.double(), None if bias is None else bias.double(), stride, (ph, pw))
    out, stats = run_conv(x, None, w, bias, stride, ph, pw, _lib.BF16)
    got = ops.to_nchw(out, cout).cpu()
    scale = ref.abs().max().item()
    assert (got - ref.float()).abs().max().item() <= 1.5e-2 * scale
    if stats is not None and stats.dtype == torch.float64:
        ssum = stats.sum(0).float().cpu()
        assert torch.allclose(ssum[0, :cout], ref.float().sum((0, 2, 3)), rtol=0, atol=1.5e-2 * scale * ref[:, 0].numel() ** 0.5)
    # backward-data
    dy = torch.randn(ref.shape, generator=g).bfloat16().float()
    ref.backward(dy.double())
    p_, q_ = ref.shape[2], ref.shape[3]
    wpt = ops.pack_weight(w.cuda(), True, cout, 0, _lib.BF16)
    dyd = ops.to_nhwc(dy.cuda(), _lib.BF16)
    dx, _ = ops.conv2d(dyd, None, wpt, None, ops.cpad(cin), r, s, stride, ph, pw, True, p=h, q=h)
    gx = ops.to_nchw(dx, cin).cpu().double()
    gscale = xd.grad.abs().max().item()
    assert (gx - xd.grad).abs().max().item() <= 1.5e-2 * gscale
    if r == 3 and stride == 1 and ops.acc_applies(ops.cpad(cin), _lib.BF16):
        # fused BatchNorm backward sums of the conv input's BatchNorm (+PReLU): sum g, sum g * xhat, sum dy * min(z, 0)
        bnx = torch.randn(n, h, h, ops.cpad(cin), generator=g).bfloat16().cuda()
        coef = (torch.rand(4, ops.cpad(cin), generator=g) + 0.5).cuda()
        alpha = torch.full((ops.cpad(cin),), 0.25, device="cuda")
        res = ops.conv_dgrad_bnbwd(dyd, wpt, ops.cpad(cin), r, s, stride, ph, pw, h, h, bnx, coef, alpha)
        assert res is not None
        dx2, acc = res
        assert torch.equal(dx2, dx)
        gq = dx2.float().reshape(-1, dx2.shape[-1])
        xf = bnx.float().reshape(-1, bnx.shape[-1])
        z = xf * coef[0] + coef[1]
        gg = torch.where(z > 0, gq, gq * alpha)
        xh = (xf - coef[2]) * coef[3]
        want = torch.stack((gg.sum(0), (gg * xh).sum(0), (gq * torch.clamp(z, max=0)).sum(0))).double()
        have = acc.sum(0)
        assert torch.allclose(have, want, rtol=2e-3, atol=2e-3 * want.abs().max().item())


@pytest.mark.parametrize("with_alpha", [False, True])
@pytest.mark.parametrize("shape", [(7, 256, 256, 14, 14), (4, 256, 128, 13, 27), (3, 512, 256, 14, 14), (5, 256, 512, 14, 14),
                                   (9, 128, 128, 28, 28), (6, 128, 256, 28, 28),
                                   # the persistent 128-channel tile (round 6): two and more rounds of tiles, ragged tile column
                                   # and row, 256 input channels (four slabs per tile)
                                   (128, 128, 128, 28, 28), (150, 128, 128, 27, 26), (131, 256, 128, 28, 28), (37, 128, 128, 56, 56),
                                   (33, 64, 128, 56, 56),          # ... and its one-slab shape (conv1 of a stage's first block)
                                   # the weights-stationary 64-channel kernel (round 5): more tiles than CUs, ragged tile column, 112 x 112
                                   (24, 64, 64, 56, 56), (60, 64, 64, 28, 40), (6, 64, 64, 112, 112)])
def test_conv_bn_from_accumulator_in_the_prologue(shape, with_alpha):
    """msml_conv2d_bnin_acc: training-mode BatchNorm (+ PReLU) -> 3x3 conv in ONE launch (coefficients derived from the
    producer's f64 accumulator in the kernel prologue, normalised tile applied in LDS and written through, running
    statistics updated by one workgroup) against msml_bn_fin_act_fwd + msml_conv2d_acc: activation, conv output, saved
    coefficients and running statistics bit for bit, output statistics to the order of the f64 adds."""
    n, cin, cout, h, w_ = shape
    if cin == 64 and cout == 64:  # (k_conv_ws's prologue transform: measured slower, experiment builds only)
        needs_experiments()
    g = torch.Generator().manual_seed(sum(shape) + int(with_alpha))
    x = ops.to_nhwc((torch.randn(n, cin, h, w_, generator=g) * 1.7 + 0.3).cuda(), _lib.BF16)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).cuda()
    gamma = (torch.rand(cin, generator=g) + 0.5).cuda()
    beta = (torch.randn(cin, generator=g) * 0.5).cuda()
    alpha = (torch.rand(cin, generator=g) * 0.3).cuda() if with_alpha else None
    wp = ops.pack_weight(w, False, cin, 0, _lib.BF16)
    m = n * h * w_
    tiles = n * ((h + 13) // 14) * ((w_ + 13) // 14)
    persistent = cout == 128 and tiles >= 512                         # (>= two rounds on 256 CUs: k_conv_halo_p)
    assert _lib.value("msml_conv2d_bnin_acc_applies", cin, cout, n, h, w_, h, w_, 3, 3, 1, 1, 1) == \
        (3 if persistent else (2 if cin == 64 else 1))

    def stats_of_x():
        acc = ops.stats_acc(cin, x.device)
        _lib.call("msml_bn_stats_acc", x, m, cin, acc, _lib.BF16)
        return acc
    # reference: two launches
    rm_a, rv_a = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
    coef_a = torch.empty(4, cin, device="cuda")
    act_a = torch.empty_like(x)
    _lib.call("msml_bn_fin_act_fwd", stats_of_x(), float(m), gamma, beta, rm_a, rv_a, 0.1, 1e-5, coef_a[0], coef_a[1],
              coef_a[2], coef_a[3], x, alpha, None, 0, act_a, m, cin, None, _lib.BF16)
    y_a, st_a = ops.conv2d(act_a, None, wp, None, cout, 3, 3, 1, 1, 1, False, want_stats=True)
    assert st_a.dtype == torch.float64
    # one launch
    rm_b, rv_b = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
    act_b, coef_b, y_b, st_b = ops.conv2d_bnin_acc(x, stats_of_x(), (gamma, beta, rm_b, rv_b, 0.1, 1e-5), alpha, wp, cout)
    torch.cuda.synchronize()
    assert torch.equal(act_b, act_a)
    assert torch.equal(y_b, y_a)
    assert torch.equal(coef_b, coef_a)
    assert torch.equal(rm_b, rm_a) and torch.equal(rv_b, rv_a)
    # (64 channels: the two-launch side's conv runs on k_conv_s2r, the fused one on k_conv_ws -- identical outputs, but the
    # f32 per-workgroup partial sums of the statistics are cut differently before they meet in f64)
    sa, sb = st_a.sum(0), st_b.sum(0)
    assert torch.allclose(sb, sa, rtol=1e-12, atol=0) if (cin != 64 or persistent) else \
        torch.allclose(sb, sa, rtol=1e-6, atol=1e-6 * sa.abs().max().item())


@pytest.mark.parametrize("with_alpha", [False, True])
@pytest.mark.parametrize("shape", [(7, 256, 256, 14, 14), (4, 256, 128, 13, 27), (3, 512, 256, 14, 14), (5, 256, 512, 14, 14),
                                   (9, 128, 128, 28, 28)])
def test_conv_dgrad_bn_backward_in_the_prologue(shape, with_alpha):
    """msml_conv2d_bnbwd_in_acc: BatchNorm backward (sums from the producer's accumulator) -> 3x3 backward-data conv ->
    sums of the next BatchNorm backward in ONE launch, against msml_bn_fin_bwd_apply + msml_conv2d_bnbwd_acc: the
    BatchNorm's input gradient (written through), the conv's input gradient and the parameter gradients bit for bit,
    the lower sums to the order of the f64 adds."""
    needs_experiments()                   # (measured slower than the two launches: the XB instantiations are not shipped)
    n, cdy, cdx, h, w_ = shape            # dy has cdy channels (the conv's output side), dx cdx
    g = torch.Generator().manual_seed(sum(shape) + int(with_alpha))
    rnd = lambda *s: torch.randn(*s, generator=g)       # noqa: E731
    dy = ops.to_nhwc(rnd(n, cdy, h, w_).cuda(), _lib.BF16)
    up_x = ops.to_nhwc((rnd(n, cdy, h, w_) * 1.3 + 0.2).cuda(), _lib.BF16)
    up_coef = torch.stack([torch.rand(cdy, generator=g) + 0.5, rnd(cdy) * 0.3, rnd(cdy) * 0.2,
                           torch.rand(cdy, generator=g) + 0.5]).cuda()
    up_alpha = (torch.rand(cdy, generator=g) * 0.3).cuda() if with_alpha else None
    m = n * h * w_
    up_acc = (torch.randn(8, 3, cdy, generator=g, dtype=torch.float64) * (m ** 0.5) / 8).cuda()
    w = (rnd(cdy, cdx, 3, 3) * (2.0 / (cdx * 9)) ** 0.5).cuda()
    wp = ops.pack_weight(w, True, cdy, 0, _lib.BF16)
    bn_x = ops.to_nhwc(rnd(n, cdx, h, w_).cuda(), _lib.BF16)
    coef = torch.stack([torch.rand(cdx, generator=g) + 0.5, rnd(cdx) * 0.3, rnd(cdx) * 0.2,
                        torch.rand(cdx, generator=g) + 0.5]).cuda()
    alpha = (torch.rand(cdx, generator=g) * 0.3).cuda()
    assert _lib.value("msml_conv2d_bnbwd_in_acc_applies", cdy, cdx, n, h, w_, h, w_, 3, 3, 1, 1, 1) == 1
    # two launches
    pg_a = [torch.full((cdy,), 0.25, device="cuda") for _ in range(3)]
    dc_a = torch.empty_like(dy)
    _lib.call("msml_bn_fin_bwd_apply", dy, up_x, up_coef[0], up_coef[1], up_alpha, up_coef[2], up_coef[3], up_acc, None,
              None, 0, 0, dc_a, None, pg_a[0], pg_a[1], pg_a[2] if with_alpha else None, 1, m, cdy, None, None, None, None,
              _lib.BF16)
    dx_a, acc_a = ops.conv_dgrad_bnbwd(dc_a, wp, cdx, 3, 3, 1, 1, 1, h, w_, bn_x, coef, alpha)
    # one launch
    pg_b = [torch.full((cdy,), 0.25, device="cuda") for _ in range(3)]
    dc_b, dx_b, acc_b = ops.conv_dgrad_bnbwd_in(dy, up_x, up_coef, up_alpha, up_acc,
                                                (pg_b[0], pg_b[1], pg_b[2] if with_alpha else None), True, wp, cdx, bn_x,
                                                coef, alpha)
    torch.cuda.synchronize()
    assert torch.equal(dc_b, dc_a)
    assert torch.equal(dx_b, dx_a)
    for a, b in zip(pg_a[:3 if with_alpha else 2], pg_b):
        assert torch.equal(a, b)
    assert acc_a.dtype == torch.float64 and torch.allclose(acc_b.sum(0), acc_a.sum(0), rtol=1e-11, atol=1e-9)


# Pointwise kernel (conv_pw.hip): 1x1 / stride-1 layers of the FM bottlenecks and the im2col'd stems -- every mode against
# the general kernel ON THE SAME OPERANDS (bit-identical outputs: same MFMA shape, k order and rounding points) and
# against an f64 reference; M = n h h is a multiple of 32 / of the 64- / 128-pixel work units or neither
@pytest.mark.parametrize("shape", [(32, 64, 9, 13), (64, 32, 8, 28), (64, 64, 3, 7), (64, 128, 5, 14), (128, 64, 4, 28),
                                   (32, 64, 1, 5), (128, 64, 1, 3)])
def test_pointwise_conv_kernel_matches_the_general_kernel(shape, monkeypatch):
    cin, cout, n, h = shape
    g = torch.Generator().manual_seed(cin * 7 + cout + n + h)
    x = ops.to_nhwc(torch.randn(n, cin, h, h, generator=g).cuda(), _lib.BF16)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).bfloat16().float().cuda()
    wp = ops.pack_weight(w, False, cin, 0, _lib.BF16)
    m = n * h * h
    monkeypatch.setattr(ops, "ACC_STATS", True)

    def both(fn):
        monkeypatch.setenv("MSML_PW_CONV", "all")      # every case the kernel supports (the default policy takes a few)
        a = fn()
        monkeypatch.setenv("MSML_PW_CONV", "0")
        b = fn()
        monkeypatch.delenv("MSML_PW_CONV", raising=False)
        return a, b

    # forward + statistics accumulator
    (o1, s1), (o2, s2) = both(lambda: ops.conv2d(x, None, wp, None, cout, 1, 1, 1, 0, 0, False, want_stats=True))
    assert s1.dtype == torch.float64 and s1.shape == (8, 2, cout)
    assert torch.equal(o1, o2)
    ref = F.conv2d(ops.to_nchw(x, cin).double(), w.double()).float()
    assert (ops.to_nchw(o1, cout) - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    f1, f2 = s1.sum(0), s2.sum(0)
    assert torch.allclose(f1, f2, rtol=1e-5, atol=1e-5 * f2.abs().max().item()), (f1 - f2).abs().max()
    # plain backward-data (transposed pack): cout -> cin
    dy = ops.to_nhwc(torch.randn(n, cout, h, h, generator=g).cuda(), _lib.BF16)
    wpt = ops.pack_weight(w, True, cout, 0, _lib.BF16)
    (d1, _), (d2, _) = both(lambda: ops.conv2d(dy, None, wpt, None, cin, 1, 1, 1, 0, 0, True, p=h, q=h))
    assert torch.equal(d1, d2)
    # backward-data + another gradient in the epilogue (msml_conv2d_fused, unit scale / zero shift)
    other = ops.to_nhwc(torch.randn(n, cin, h, h, generator=g).cuda(), _lib.BF16)
    ones, zeros = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")

    def plus():
        out = torch.empty(n, h, h, cin, dtype=torch.bfloat16, device="cuda")
        _lib.call("msml_conv2d_fused", dy, cout, None, 0, wpt, wpt.shape[0], ones, zeros, None, other, 0, out, cin,
                  n, h, h, h, h, 1, 1, 1, 0, 0, 1)
        return out
    p1, p2 = both(plus)
    assert torch.equal(p1, p2)
    assert (p1.float() - (d1.float() + other.float())).abs().max().item() <= 2e-2 * p1.float().abs().max().item()
    # ... with a real per-channel scale / shift
    sc, sh = torch.rand(cin, generator=g).cuda() + 0.5, torch.randn(cin, generator=g).cuda() * 0.1

    def affine():
        out = torch.empty(n, h, h, cin, dtype=torch.bfloat16, device="cuda")
        _lib.call("msml_conv2d_fused", dy, cout, None, 0, wpt, wpt.shape[0], sc, sh, None, other, 0, out, cin,
                  n, h, h, h, h, 1, 1, 1, 0, 0, 1)
        return out
    a1, a2 = both(affine)
    assert torch.equal(a1, a2)
    # backward-data + the BatchNorm(+PReLU) backward sums of the layer in front (accumulator protocol)
    xbn = ops.to_nhwc(torch.randn(n, cin, h, h, generator=g).cuda(), _lib.BF16)
    coef = torch.stack([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3,
                        torch.randn(cin, generator=g) * 0.2, torch.rand(cin, generator=g) + 0.5]).cuda()
    for alpha in (None, (torch.rand(cin, generator=g) * 0.5).cuda()):
        (b1, q1), (b2, q2) = both(lambda: ops.conv_dgrad_bnbwd(dy, wpt, cin, 1, 1, 1, 0, 0, h, h, xbn, coef, alpha))
        assert torch.equal(b1, b2) and torch.equal(b1, d1)
        assert q1.shape == (8, 3, cin) and q1.dtype == torch.float64
        # both against the sums recomputed in f64 from the stored gradient (bn.hip's k_bn_bwd_reduce arithmetic)
        d, xs = b1.double().reshape(-1, cin), xbn.double().reshape(-1, cin)
        z = xs * coef[0].double() + coef[1].double()
        neg = (z <= 0) if alpha is not None else torch.zeros_like(z, dtype=torch.bool)
        gg = torch.where(neg, d * alpha.double(), d) if alpha is not None else d
        xh = (xs - coef[2].double()) * coef[3].double()
        want = torch.stack([gg.sum(0), (gg * xh).sum(0), torch.where(neg, d * z, torch.zeros_like(z)).sum(0)])
        for got in (q1.sum(0), q2.sum(0)):
            assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-6, (got - want).abs().max()


# Row-streaming kernel for the 32 -> 32 channel 3x3 layers (conv_r32.hip: conv2 of the first FM stage's bottlenecks):
# forward + statistics, plain backward-data, backward-data + BatchNorm backward sums against the general kernel on the
# same operands (bit-identical outputs) and against f64; odd sizes, the widest row it takes, more strips than workers
@pytest.mark.parametrize("shape", [(3, 56, 56), (2, 28, 28), (5, 14, 14), (2, 9, 13), (1, 60, 62), (1, 2, 8), (300, 56, 56)])
def test_conv3x3_32_channel_row_kernel_matches_the_general_kernel(shape, monkeypatch):
    n, h, w_ = shape
    c = 32
    g = torch.Generator().manual_seed(n * 7 + h + w_)
    x = ops.to_nhwc(torch.randn(n, c, h, w_, generator=g).cuda(), _lib.BF16)
    w = (torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5).bfloat16().float().cuda()
    wp = ops.pack_weight(w, False, c, 0, _lib.BF16)
    wpt = ops.pack_weight(w, True, c, 0, _lib.BF16)
    monkeypatch.setattr(ops, "ACC_STATS", True)

    def both(fn):
        monkeypatch.delenv("MSML_NO_R32_CONV", raising=False)
        a = fn()
        monkeypatch.setenv("MSML_NO_R32_CONV", "1")
        b = fn()
        monkeypatch.delenv("MSML_NO_R32_CONV", raising=False)
        return a, b

    (o1, s1), (o2, s2) = both(lambda: ops.conv2d(x, None, wp, None, c, 3, 3, 1, 1, 1, False, want_stats=True))
    assert torch.equal(o1, o2)
    if n <= 8:
        ref = F.conv2d(ops.to_nchw(x, c).double(), w.double(), None, 1, 1).float()
        assert (ops.to_nchw(o1, c) - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    f1, f2 = s1.sum(0), s2.sum(0)
    assert torch.allclose(f1, f2, rtol=1e-5, atol=1e-5 * f2.abs().max().item()), (f1 - f2).abs().max()
    (p1, _), (p2, _) = both(lambda: ops.conv2d(x, None, wp, None, c, 3, 3, 1, 1, 1, False))
    assert torch.equal(p1, p2) and torch.equal(p1, o1)
    dy = ops.to_nhwc(torch.randn(n, c, h, w_, generator=g).cuda(), _lib.BF16)
    (d1, _), (d2, _) = both(lambda: ops.conv2d(dy, None, wpt, None, c, 3, 3, 1, 1, 1, True, p=h, q=w_))
    assert torch.equal(d1, d2)
    if n <= 8:
        rdx = F.conv_transpose2d(ops.to_nchw(dy, c).double(), w.double(), None, 1, 1).float()
        assert (ops.to_nchw(d1, c) - rdx).abs().max().item() <= 1.5e-2 * rdx.abs().max().item()
    xbn = ops.to_nhwc(torch.randn(n, c, h, w_, generator=g).cuda(), _lib.BF16)
    coef = torch.stack([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3,
                        torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5]).cuda()
    for alpha in (None, (torch.rand(c, generator=g) * 0.5).cuda()):
        (b1, q1), (b2, q2) = both(lambda: ops.conv_dgrad_bnbwd(dy, wpt, c, 3, 3, 1, 1, 1, h, w_, xbn, coef, alpha))
        assert torch.equal(b1, b2) and torch.equal(b1, d1)
        d, xs = b1.double().reshape(-1, c), xbn.double().reshape(-1, c)
        z = xs * coef[0].double() + coef[1].double()
        neg = (z <= 0) if alpha is not None else torch.zeros_like(z, dtype=torch.bool)
        gg = torch.where(neg, d * alpha.double(), d) if alpha is not None else d
        xh = (xs - coef[2].double()) * coef[3].double()
        want = torch.stack([gg.sum(0), (gg * xh).sum(0), torch.where(neg, d * z, torch.zeros_like(z)).sum(0)])
        for got in (q1.sum(0), q2.sum(0)):
            assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-6, (got - want).abs().max()


# conv_halo2.hip: (N, Cin, Cout, H, stride).  Stride 2: forward through the four parity planes, backward-data per output
# class; 7 x 7 GEMM grids (stride 1 at 7 x 7, stride 2 at 14 -> 7): 2 x 2 image mosaics (ragged batches: N % 4 != 0)
HALO2 = [
    (5, 128, 128, 28, 2),        # FRB layer2 entry at a quarter of the map: one tile per image, 128-channel tiling
    (3, 128, 128, 56, 2),        # 2 x 2 tiles per image
    (6, 256, 256, 28, 2),        # 256-channel tiling, four slabs x four planes
    (4, 128, 256, 26, 2),        # ragged tile (13 x 13 outputs)
    (161, 512, 512, 7, 1),       # mosaic, ragged batch (the last tile holds one image); >= 160 workgroups
    (325, 256, 256, 7, 1),
    (162, 512, 512, 14, 2),      # stride 2 onto a 7 x 7 grid: mosaic of parity planes
    (241, 512, 512, 4, 1),       # 4 x 4 maps: 3 x 2 images per tile, ragged batch (the last tile holds one image)
    (482, 256, 256, 4, 1),
    (81, 256, 256, 14, 2),       # backward-data only (four slices per tile): the forward has 42 workgroups
]


@pytest.mark.parametrize("shape", HALO2)
def test_conv_halo2_stride2_and_mosaic(shape):
    """k_conv_halo2 (stride-2 3x3 layers on the halo tile, 7x7 maps as four-image mosaics): forward + accumulator-mode
    statistics, backward-data, backward-data + residual and backward-data with the fused BatchNorm backward sums, against
    f64 torch on the same bf16-rounded operands (backbones/frb/iresnet.py:56-67, stride-2 conv2 and the 512-channel stage)."""
    n, cin, cout, h, stride = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, h, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).bfloat16().float()
    ho = (h + 2 - 3) // stride + 1
    name = _lib.value("msml_conv2d_kernel", cin, 0, cout, n, h, h, ho, ho, 3, 3, stride, 1, 1, 0, _lib.BF16, _lib.BF16, 1).decode()
    assert "k_conv_halo2" in name or shape == (81, 256, 256, 14, 2), name     # (that forward: < 160 workgroups, im2col)
    xd = x.double().requires_grad_(True)
    ref = F.conv2d(xd, w.double(), None, stride, 1)
    out, stats = run_conv(x, None, w, None, stride, 1, 1, _lib.BF16)
    got = ops.to_nchw(out, cout).cpu()
    scale = ref.abs().max().item()
    assert (got - ref.float()).abs().max().item() <= 1.5e-2 * scale
    assert stats.dtype == torch.float64
    ssum = stats.sum(0).float().cpu()
    assert torch.allclose(ssum[0, :cout], ref.float().sum((0, 2, 3)), rtol=0, atol=1.5e-2 * scale * ref[:, 0].numel() ** 0.5)
    assert torch.allclose(ssum[1, :cout], (ref.float() ** 2).sum((0, 2, 3)), rtol=2e-2)
    # backward-data
    dy = torch.randn(ref.shape, generator=g).bfloat16().float()
    ref.backward(dy.double())
    wpt = ops.pack_weight(w.cuda(), True, cout, 0, _lib.BF16)
    dyd = ops.to_nhwc(dy.cuda(), _lib.BF16)
    name = _lib.value("msml_conv2d_kernel", cout, 0, cin, n, ho, ho, h, h, 3, 3, stride, 1, 1, 1, _lib.BF16, _lib.BF16, 0).decode()
    assert "k_conv_halo2" in name, name
    dx, _ = ops.conv2d(dyd, None, wpt, None, cin, 3, 3, stride, 1, 1, True, p=h, q=h)
    gx = ops.to_nchw(dx, cin).cpu().double()
    gscale = xd.grad.abs().max().item()
    assert (gx - xd.grad).abs().max().item() <= 1.5e-2 * gscale
    # ... + residual (the FM `conv_tee` join, msml_conv2d_fused with unit scale / zero shift)
    res = torch.randn(n, h, h, cin, generator=g).bfloat16().cuda()
    dxr = torch.empty_like(dx)
    one, zero = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    _lib.call("msml_conv2d_fused", dyd, cout, None, 0, wpt, wpt.shape[0], one, zero, None, res, 0, dxr, cin, n, ho, ho, h, h,
              3, 3, stride, 1, 1, 1)
    want = (dx.float() + res.float()).bfloat16()
    assert (dxr.float() - want.float()).abs().max().item() <= 8e-3 * max(1.0, want.float().abs().max().item())
    # ... with the fused BatchNorm backward sums (accumulator mode): sum g, sum g * xhat, sum dy * min(z, 0)
    bnx = torch.randn(n, h, h, cin, generator=g).bfloat16().cuda()
    coef = (torch.rand(4, cin, generator=g) + 0.5).cuda()
    alpha = torch.full((cin,), 0.25, device="cuda")
    r2 = ops.conv_dgrad_bnbwd(dyd, wpt, cin, 3, 3, stride, 1, 1, h, h, bnx, coef, alpha)
    assert r2 is not None
    dx2, acc = r2
    assert torch.equal(dx2, dx)
    gq = dx2.float().reshape(-1, cin)
    xf = bnx.float().reshape(-1, cin)
    z = xf * coef[0] + coef[1]
    gg = torch.where(z > 0, gq, gq * alpha)
    xh = (xf - coef[2]) * coef[3]
    want = torch.stack((gg.sum(0), (gg * xh).sum(0), (gq * torch.clamp(z, max=0)).sum(0))).double()
    assert torch.allclose(acc.sum(0), want, rtol=2e-3, atol=2e-3 * want.abs().max().item())


# conv_s2r.hip: 64 -> 64 channel stride-2 3x3 layers with the weights in registers (N, H): one tile per workgroup, more
# tiles than workgroups (persistent loop, counted waits across tiles), ragged tiles (13 x 13 outputs), the OSB's 56 -> 28
@pytest.mark.parametrize("shape", [(6, 56), (20, 112), (5, 28), (4, 26), (70, 56)])
def test_conv_s2r_64_channel_stride2(shape):
    """k_conv_s2r: forward + accumulator-mode statistics, backward-data, backward-data with the fused BatchNorm backward
    sums, against f64 torch on the same bf16-rounded operands (backbones/frb/iresnet.py:56-67, layer1's stride-2 conv2)."""
    n, h = shape
    c = 64
    g = torch.Generator().manual_seed(n * 131 + h)
    x = torch.randn(n, c, h, h, generator=g).bfloat16().float()
    w = (torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5).bfloat16().float()
    ho = h // 2
    name = _lib.value("msml_conv2d_kernel", c, 0, c, n, h, h, ho, ho, 3, 3, 2, 1, 1, 0, _lib.BF16, _lib.BF16, 1).decode()
    assert "k_conv_s2r" in name, name
    xd = x.double().requires_grad_(True)
    ref = F.conv2d(xd, w.double(), None, 2, 1)
    out, stats = run_conv(x, None, w, None, 2, 1, 1, _lib.BF16)
    got = ops.to_nchw(out, c).cpu()
    scale = ref.abs().max().item()
    assert (got - ref.float()).abs().max().item() <= 1.5e-2 * scale
    assert stats.dtype == torch.float64
    ssum = stats.sum(0).float().cpu()
    assert torch.allclose(ssum[0], ref.float().sum((0, 2, 3)), rtol=0, atol=1.5e-2 * scale * ref[:, 0].numel() ** 0.5)
    assert torch.allclose(ssum[1], (ref.float() ** 2).sum((0, 2, 3)), rtol=2e-2)
    dy = torch.randn(ref.shape, generator=g).bfloat16().float()
    ref.backward(dy.double())
    wpt = ops.pack_weight(w.cuda(), True, c, 0, _lib.BF16)
    dyd = ops.to_nhwc(dy.cuda(), _lib.BF16)
    name = _lib.value("msml_conv2d_kernel", c, 0, c, n, ho, ho, h, h, 3, 3, 2, 1, 1, 1, _lib.BF16, _lib.BF16, 0).decode()
    assert "k_conv_s2r" in name, name
    dx, _ = ops.conv2d(dyd, None, wpt, None, c, 3, 3, 2, 1, 1, True, p=h, q=h)
    gx = ops.to_nchw(dx, c).cpu().double()
    assert (gx - xd.grad).abs().max().item() <= 1.5e-2 * xd.grad.abs().max().item()
    bnx = torch.randn(n, h, h, c, generator=g).bfloat16().cuda()
    coef = (torch.rand(4, c, generator=g) + 0.5).cuda()
    for alpha in (None, torch.full((c,), 0.25, device="cuda")):
        r2 = ops.conv_dgrad_bnbwd(dyd, wpt, c, 3, 3, 2, 1, 1, h, h, bnx, coef, alpha)
        assert r2 is not None
        dx2, acc = r2
        assert torch.equal(dx2, dx)
        gq = dx2.float().reshape(-1, c)
        xf = bnx.float().reshape(-1, c)
        z = xf * coef[0] + coef[1]
        neg = (z <= 0) if alpha is not None else torch.zeros_like(z, dtype=torch.bool)
        gg = torch.where(neg, gq * 0.25, gq)
        xh = (xf - coef[2]) * coef[3]
        want = torch.stack((gg.sum(0), (gg * xh).sum(0), torch.where(neg, gq * z, torch.zeros_like(z)).sum(0))).double()
        assert torch.allclose(acc.sum(0), want, rtol=2e-3, atol=2e-3 * want.abs().max().item())


# stride-2 weight gradients on the strip kernel (k_wgrad_halo<64, S2>: the X strip as four parity planes): (N, Cin, Cout, H)
# with H the conv INPUT size; more strips than splits, ragged strips (26 -> 13), the 256-channel layer (16 dW tiles)
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("shape", [(6, 64, 64, 56), (3, 64, 64, 112), (9, 128, 128, 28), (4, 64, 128, 26), (5, 256, 256, 28)])
def test_conv_wgrad_stride2_on_the_strip_kernel(shape, accumulate, monkeypatch):
    needs_experiments()
    monkeypatch.setenv("MSML_HALO_WGRAD_S2", "1")         # (opt-in: measured not faster than the im2col kernel)
    n, cin, cout, h = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, h, generator=g).bfloat16().float()
    ho = h // 2
    dy = torch.randn(n, cout, ho, ho, generator=g).bfloat16().float()
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv2d(x.double(), w, None, 2, 1).backward(dy.double())
    ref = w.grad.float()
    dw = torch.full((cout, cin, 3, 3), 2.0, device="cuda")
    ops.conv_wgrad(ops.to_nhwc(dy.cuda(), _lib.BF16), ops.to_nhwc(x.cuda(), _lib.BF16), dw, cout, cin, cin, 0,
                   3, 3, 2, 1, 1, accumulate=accumulate)
    want = ref + (2.0 if accumulate else 0.0)
    assert (dw.cpu() - want).abs().max().item() <= 1.5e-2 * ref.abs().max().item()


# conv_s2r.hip, k_conv_s2r_x3: split-bf16 (config 5) 64 -> 64 channel 3x3 / stride-1 layers with both weight planes in
# registers (N, H, W): one tile per workgroup, the persistent loop (more tiles than workgroups), ragged tiles, a non-square map
@pytest.mark.parametrize("epi", ["plain", "bn_prelu", "bn_res_prelu", "bn_prelu_res", "bias"])
@pytest.mark.parametrize("shape", [(3, 56, 56), (2, 112, 112), (5, 28, 28), (4, 27, 40), (40, 56, 56)])
def test_conv_x3_64_channel_register_kernel(shape, epi, monkeypatch):
    """Split-bf16 conv + affine / PReLU / residual epilogue (functional.conv_x3 -> msml_conv2d_x3) on the register-weights
    kernel: against f64 torch on the values the split tensors hold (f32-class: 3e-5 of the output scale, where one bf16
    product is 4e-3), and against the general kernel (k_conv_fast X3), which differs only in the order of its f32 sums."""
    import torch.nn as nn
    from msml_amd import functional as Fh
    n, h, w = shape
    c = 64
    g = torch.Generator().manual_seed(n * 977 + h * 31 + w + len(epi))
    x = torch.randn(n, h, w, c, generator=g).cuda()
    conv = nn.Conv2d(c, c, 3, 1, 1, bias=(epi == "bias")).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5)
        if conv.bias is not None:
            conv.bias.copy_(torch.randn(c, generator=g))
    scale = (torch.rand(c, generator=g) + 0.5).cuda() if epi.startswith("bn") else None
    shift = torch.randn(c, generator=g).cuda() if epi.startswith("bn") else (conv.bias.detach() if epi == "bias" else None)
    alpha = (torch.rand(c, generator=g) * 0.5).cuda() if "prelu" in epi else None
    res = torch.randn(n, h, w, c, generator=g).cuda() if "res" in epi else None
    res_first = epi == "bn_res_prelu"
    xs = Fh.x3_from_f32(x)
    rs = Fh.x3_from_f32(res) if res is not None else None

    def run():
        return Fh.x3_to_f32(Fh.conv_x3(xs, None, conv, scale, shift, alpha, rs, res_first))

    got = run()
    torch.cuda.synchronize()
    monkeypatch.setenv("MSML_NO_S2R_X3", "1")
    gen = run()
    torch.cuda.synchronize()
    monkeypatch.delenv("MSML_NO_S2R_X3")
    xv = Fh.x3_to_f32(xs).double().permute(0, 3, 1, 2)
    wh = conv.weight.detach().to(torch.bfloat16).float()
    wv = (wh + (conv.weight.detach() - wh).to(torch.bfloat16).float()).double()      # what [wh | wh | wl] holds
    y = F.conv2d(xv, wv, None, 1, 1)
    cs = (1, c, 1, 1)
    if scale is not None:
        y = y * scale.double().view(cs)
    if shift is not None:
        y = y + shift.double().view(cs)
    rv = Fh.x3_to_f32(rs).double().permute(0, 3, 1, 2) if rs is not None else None
    if rv is not None and res_first:
        y = y + rv
    if alpha is not None:
        y = torch.where(y > 0, y, y * alpha.double().view(cs))
    if rv is not None and not res_first:
        y = y + rv
    ref = y.permute(0, 2, 3, 1)
    tol = 3e-5 * ref.abs().max().item()
    assert (gen.double() - ref).abs().max().item() <= tol          # (the general kernel meets the same bar)
    assert (got.double() - ref).abs().max().item() <= tol
    assert (got - gen).abs().max().item() <= tol


# msml_conv2d_x3_border on its three kernels: (N, Cin, Cout, H, W) -- k_conv_s2r_x3 (64 -> 64), k_conv_halo X3 (one 14 x 14 tile
# per image and four per image, ragged 13 x 20), k_conv_fast X3 (channel-changing conv1 of a stage's first block, 7 x 7 maps)
@pytest.mark.parametrize("shape", [(3, 64, 64, 56, 56), (4, 64, 64, 27, 40), (5, 128, 128, 28, 28), (6, 256, 256, 14, 14),
                                   (3, 256, 256, 13, 20), (4, 64, 128, 28, 28), (7, 512, 512, 7, 7), (2, 128, 256, 14, 14)])
def test_conv_x3_with_the_leading_batchnorm_folded_in(shape):
    """functional.bn_conv_bn_eval_x3 (bn1 -> conv1 -> bn2 -> PReLU of IBasicBlock, backbones/frb/iresnet.py:58-62, eval mode,
    split-bf16): the leading BatchNorm folded into the conv operand + a shift per border class, against f64 torch on the
    values the split input holds, and against the unfolded pair of launches."""
    import torch.nn as nn
    from msml_amd import functional as Fh
    n, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False).cuda()
    bn1, bn2 = nn.BatchNorm2d(cin).cuda().eval(), nn.BatchNorm2d(cout).cuda().eval()
    prelu = nn.PReLU(cout).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5)
        for bn in (bn1, bn2):
            c = bn.num_features
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g) * 0.5)
            bn.running_mean.copy_(torch.randn(c, generator=g) * 0.3)
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        prelu.weight.copy_(torch.rand(cout, generator=g) * 0.5)
    xs = Fh.x3_from_f32(x)
    out = Fh.bn_conv_bn_eval_x3(xs, bn1, conv, bn2, prelu)
    assert out is not None
    got = Fh.x3_to_f32(out)
    two = Fh.x3_to_f32(Fh.conv_bn_eval_x3(Fh.bn_act_x3(xs, bn1), None, conv, bn2, prelu, None, 0, False))
    with torch.no_grad():
        xv = Fh.x3_to_f32(xs).double().permute(0, 3, 1, 2)
        ref = prelu.double()(bn2.double()(conv.double()(bn1.double()(xv)))).permute(0, 2, 3, 1)
    tol = 4e-5 * ref.abs().max().item()
    assert (two.double() - ref).abs().max().item() <= tol          # (the unfolded pair meets the same bar)
    assert (got.double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("ratio", [6.0, 50.0])
def test_conv_x3_batchnorm_fold_under_cancellation(ratio):
    """with running_mean >> sqrt(running_var) on the block input the folded form conv(W s, x) + sum(W t) is a
    difference of two large terms, and the split-bf16 rounding of the first is amplified by |mean| / std.  Block input with
    mean / std = `ratio` per channel, against f64 torch at the 2e-4 parity bar of the embedding path (the output scale):
    ratio 6 is folded and inside the bar; ratio 50 is REFUSED by functional.bn_conv_bn_eval_x3 (X3_FOLD_MAX_RATIO = 8) --
    IBasicBlock.forward then runs the separate pass, which meets the bar at any ratio."""
    import torch.nn as nn
    from msml_amd import functional as Fh
    n, cin, cout, h, w = 4, 128, 128, 28, 28
    g = torch.Generator().manual_seed(int(ratio))
    std = torch.rand(cin, generator=g) + 0.5
    mean = std * ratio * (torch.randint(0, 2, (cin,), generator=g).float() * 2 - 1)
    x = (torch.randn(n, h, w, cin, generator=g) * std + mean).cuda()
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False).cuda()
    bn1, bn2 = nn.BatchNorm2d(cin).cuda().eval(), nn.BatchNorm2d(cout).cuda().eval()
    prelu = nn.PReLU(cout).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5)
        bn1.weight.copy_(torch.rand(cin, generator=g) + 0.5)
        bn1.bias.copy_(torch.randn(cin, generator=g) * 0.5)
        bn1.running_mean.copy_(mean)
        bn1.running_var.copy_(std * std)
        bn2.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn2.bias.copy_(torch.randn(cout, generator=g) * 0.5)
        bn2.running_mean.copy_(torch.randn(cout, generator=g) * 0.3)
        bn2.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
        prelu.weight.copy_(torch.rand(cout, generator=g) * 0.5)
    xs = Fh.x3_from_f32(x)
    with torch.no_grad():
        xv = Fh.x3_to_f32(xs).double().permute(0, 3, 1, 2)
        ref = prelu.double()(bn2.double()(conv.double()(bn1.double()(xv)))).permute(0, 2, 3, 1)
    for mod in (conv, bn1, bn2, prelu):
        mod.float()
    tol = 2e-4 * ref.abs().max().item()
    out = Fh.bn_conv_bn_eval_x3(xs, bn1, conv, bn2, prelu)
    two = Fh.x3_to_f32(Fh.conv_bn_eval_x3(Fh.bn_act_x3(xs, bn1), None, conv, bn2, prelu, None, 0, False))
    assert (two.double() - ref).abs().max().item() <= tol
    if ratio > Fh.X3_FOLD_MAX_RATIO:
        assert out is None                                    # refused: the block falls back to the separate pass
        assert Fh.bn_conv_bn_eval_x3(xs, bn1, conv, bn2, prelu) is None      # (cached decision)
    else:
        assert out is not None
        assert (Fh.x3_to_f32(out).double() - ref).abs().max().item() <= tol


# k_conv_halo2 with the split-bf16 epilogue (config 5): (N, C, H, stride) -- stride-2 layers on 14 x 14 tiles (128- and 256-wide
# channel blocks, ragged 26 -> 13), 14 -> 7 and 7 x 7 maps as four-image mosaics, 4 x 4 maps as six-image mosaics (a mosaic
# needs ~160 workgroups, hence the batch)
@pytest.mark.parametrize("epi", ["bn_prelu", "bn_res", "bn_res_prelu"])
@pytest.mark.parametrize("shape", [(6, 128, 56, 2), (5, 256, 28, 2), (4, 128, 26, 2), (161, 512, 14, 2), (164, 512, 7, 1),
                                   (243, 512, 4, 1)])
def test_conv_x3_stride2_and_mosaics_on_the_halo_tile(shape, epi, monkeypatch):
    import torch.nn as nn
    from msml_amd import functional as Fh
    n, c, h, stride = shape
    ho = h // stride
    g = torch.Generator().manual_seed(sum(shape) + len(epi))
    x = torch.randn(n, h, h, c, generator=g).cuda()
    conv = nn.Conv2d(c, c, 3, stride, 1, bias=False).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(c, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5)
    scale = (torch.rand(c, generator=g) + 0.5).cuda()
    shift = torch.randn(c, generator=g).cuda()
    alpha = (torch.rand(c, generator=g) * 0.5).cuda() if "prelu" in epi else None
    res = torch.randn(n, ho, ho, c, generator=g).cuda() if "res" in epi else None
    res_first = epi == "bn_res_prelu"
    xs = Fh.x3_from_f32(x)
    rs = Fh.x3_from_f32(res) if res is not None else None

    def run():
        return Fh.x3_to_f32(Fh.conv_x3(xs, None, conv, scale, shift, alpha, rs, res_first))

    got = run()
    torch.cuda.synchronize()
    monkeypatch.setenv("MSML_NO_HALO2_X3", "1")
    gen = run()
    torch.cuda.synchronize()
    monkeypatch.delenv("MSML_NO_HALO2_X3")
    xv = Fh.x3_to_f32(xs).double().permute(0, 3, 1, 2)
    wh = conv.weight.detach().to(torch.bfloat16).float()
    wv = (wh + (conv.weight.detach() - wh).to(torch.bfloat16).float()).double()
    cs = (1, c, 1, 1)
    y = F.conv2d(xv, wv, None, stride, 1) * scale.double().view(cs) + shift.double().view(cs)
    rv = Fh.x3_to_f32(rs).double().permute(0, 3, 1, 2) if rs is not None else None
    if rv is not None and res_first:
        y = y + rv
    if alpha is not None:
        y = torch.where(y > 0, y, y * alpha.double().view(cs))
    if rv is not None and not res_first:
        y = y + rv
    ref = y.permute(0, 2, 3, 1)
    tol = 4e-5 * ref.abs().max().item()
    assert (gen.double() - ref).abs().max().item() <= tol
    assert (got.double() - ref).abs().max().item() <= tol
    assert (got - gen).abs().max().item() <= tol
    if stride == 1:                                       # the leading-BatchNorm fold on the mosaic (border classes per image)
        bn1, bn2 = nn.BatchNorm2d(c).cuda().eval(), nn.BatchNorm2d(c).cuda().eval()
        prelu = nn.PReLU(c).cuda()
        with torch.no_grad():
            for bn in (bn1, bn2):
                bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
                bn.bias.copy_(torch.randn(c, generator=g) * 0.5)
                bn.running_mean.copy_(torch.randn(c, generator=g) * 0.3)
                bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
            prelu.weight.copy_(torch.rand(c, generator=g) * 0.5)
            ref2 = prelu.double()(bn2.double()(conv.double()(bn1.double()(xv)))).permute(0, 2, 3, 1)
            conv.float()
            bn1.float(); bn2.float(); prelu.float()
        out = Fh.bn_conv_bn_eval_x3(xs, bn1, conv, bn2, prelu)
        assert out is not None
        assert (Fh.x3_to_f32(out).double() - ref2).abs().max().item() <= 4e-5 * ref2.abs().max().item()


# split-bf16 convs on the small maps (64-row tiles of the general kernel whatever the batch): (N, Cin, Cout, H, (R, S), bias) --
# the deepest GCM line convs (512 -> 18 @ 4x4), 256- / 128-channel 3x3 layers @ 7x7, a bottleneck 1x1
@pytest.mark.parametrize("shape", [(37, 256, 256, 7, (3, 3), False), (50, 512, 18, 4, (7, 1), True), (50, 512, 18, 4, (1, 7), True),
                                   (21, 128, 128, 7, (3, 3), False), (64, 512, 128, 7, (1, 1), False), (3, 256, 18, 7, (7, 1), True)])
def test_conv_x3_small_maps(shape):
    import torch.nn as nn
    from msml_amd import functional as Fh
    n, cin, cout, h, (r, s_), bias = shape
    g = torch.Generator().manual_seed(sum(shape[:4]) + r)
    x = torch.randn(n, h, h, cin, generator=g).cuda()
    conv = nn.Conv2d(cin, cout, (r, s_), 1, (r // 2, s_ // 2), bias=bias).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, r, s_, generator=g) * (2.0 / (cin * r * s_)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g))
    xs = Fh.x3_from_f32(x)
    if bias:
        got = Fh.x3_to_f32(Fh.conv_plain_x3(conv, xs, None, 0))[..., :cout]
        extra = lambda y: y + conv.bias.detach().double().view(1, cout, 1, 1)      # noqa: E731
    else:
        scale = (torch.rand(cout, generator=g) + 0.5).cuda()
        shift = torch.randn(cout, generator=g).cuda()
        alpha = (torch.rand(cout, generator=g) * 0.5).cuda()
        got = Fh.x3_to_f32(Fh.conv_x3(xs, None, conv, scale, shift, alpha, None, False))[..., :cout]

        def extra(y):
            y = y * scale.double().view(1, cout, 1, 1) + shift.double().view(1, cout, 1, 1)
            return torch.where(y > 0, y, y * alpha.double().view(1, cout, 1, 1))
    xv = Fh.x3_to_f32(xs).double().permute(0, 3, 1, 2)
    wh = conv.weight.detach().to(torch.bfloat16).float()
    wv = (wh + (conv.weight.detach() - wh).to(torch.bfloat16).float()).double()
    ref = extra(F.conv2d(xv, wv, None, 1, (r // 2, s_ // 2))).permute(0, 2, 3, 1)
    assert (got.double() - ref).abs().max().item() <= 4e-5 * ref.abs().max().item()


# k_conv_halo_p: the persistent 128-channel tile (several rounds of 14 x 14 tiles per launch): (N, H, W) with >= 512 tiles --
# four tiles per image, ragged tiles (27 x 40: 6 per image), more tiles than twice the workgroups
@pytest.mark.parametrize("shape", [(130, 28, 28), (90, 27, 40), (260, 28, 28), (40, 56, 56, 64), (140, 28, 28, 256)])
def test_conv_halo_persistent_128_channel_tile(shape, monkeypatch):
    """Forward + accumulator-mode statistics and backward-data + fused BatchNorm sums on the persistent kernel: outputs
    bit-identical to the one-tile-per-workgroup kernel (same per-tile arithmetic), sums equal to f32 rounding (the tiles are
    added in another order), both against f64 torch (backbones/frb/iresnet.py:56-67, layer2's 128-channel blocks)."""
    n, h, w = shape[:3]
    cin = shape[3] if len(shape) > 3 else 128         # (64: one slab per tile; 256: four)
    c = 128
    g = torch.Generator().manual_seed(n + h + cin)
    x = torch.randn(n, cin, h, w, generator=g).bfloat16().float()
    wt = (torch.randn(c, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).bfloat16().float()
    xd = ops.to_nhwc(x.cuda(), _lib.BF16)
    wp = ops.pack_weight(wt.cuda(), False, cin, 0, _lib.BF16)
    # the backward-data launch of the test is a 128-output-channel conv of its own: dY with `cin` channels -> dX with 128
    wt2 = (torch.randn(cin, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5).bfloat16().float()
    wpt = ops.pack_weight(wt2.cuda(), True, cin, 0, _lib.BF16)
    dy = torch.randn(n, cin, h, w, generator=g).bfloat16().float()
    dyd = ops.to_nhwc(dy.cuda(), _lib.BF16)
    bnx = torch.randn(n, h, w, c, generator=g).bfloat16().cuda()
    coef = (torch.rand(4, c, generator=g) + 0.5).cuda()
    alpha = torch.full((c,), 0.25, device="cuda")

    def run():
        acc = ops.stats_acc(c, xd.device)
        y = torch.empty(n, h, w, c, dtype=torch.bfloat16, device="cuda")
        _lib.call("msml_conv2d_acc", xd, cin, None, 0, wp, wp.shape[0], None, y, c, acc, n, h, w, h, w, 3, 3, 1, 1, 1, 0,
                  _lib.BF16, _lib.BF16)
        dx, bacc = ops.conv_dgrad_bnbwd(dyd, wpt, c, 3, 3, 1, 1, 1, h, w, bnx, coef, alpha)
        torch.cuda.synchronize()
        return y, acc.clone(), dx, bacc.clone()

    y1, a1, dx1, b1 = run()
    monkeypatch.setenv("MSML_HALO_PERSIST", "0")
    y0, a0, dx0, b0 = run()
    monkeypatch.delenv("MSML_HALO_PERSIST")
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    assert torch.allclose(a1.sum(0), a0.sum(0), rtol=1e-5, atol=1e-5 * a0.sum(0).abs().max().item())
    assert torch.allclose(b1.sum(0), b0.sum(0), rtol=1e-5, atol=1e-5 * b0.sum(0).abs().max().item())
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    got = ops.to_nchw(y1, c).cpu().double()
    assert (got - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    assert torch.allclose(a1.sum(0)[0].cpu(), ref.sum((0, 2, 3)), rtol=0, atol=1e-3 * ref.abs().max().item() * (n * h * w) ** 0.5)
    dref = F.conv_transpose2d(dy.double(), wt2.double(), None, 1, 1)
    assert (ops.to_nchw(dx1, c).cpu().double() - dref).abs().max().item() <= 1.5e-2 * dref.abs().max().item()


# the stems' im2col through LDS (k_stem_im2col_lds) against the gather kernel it replaces: (N, H, W, stride) -- the two stems of
# the step (112 x 112, stride 1 and 2), ragged row groups (P % 4 != 0), odd widths, one image
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(3, 112, 112, 1), (2, 112, 112, 2), (5, 30, 42, 1), (4, 30, 42, 2), (1, 7, 9, 1), (2, 13, 128, 2)])
def test_stem_im2col_through_lds_is_the_gather(shape, dtype, monkeypatch):
    n, h, w_, stride = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, 3, h, w_, generator=g).cuda()
    dt = _lib.BF16 if dtype == "bf16" else _lib.F32
    got = ops.stem_im2col(x, 3, 3, stride, 1, dtype=dt)
    monkeypatch.setenv("MSML_NO_STEM_LDS", "1")
    ref = ops.stem_im2col(x, 3, 3, stride, 1, dtype=dt)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and torch.equal(got, ref)
    # and against unfold: k = (r * 3 + s) * 3 + c, zeros above k = 27
    cols = F.unfold(x, 3, padding=1, stride=stride).view(n, 3, 9, -1).permute(0, 3, 2, 1).reshape(n, got.shape[1], got.shape[2], 27)
    want = cols.bfloat16().float() if dtype == "bf16" else cols
    assert torch.equal(got[..., :27].float(), want) and float(got[..., 27:].float().abs().max()) == 0.0


# msml_bn_fin_bwd_apply_next_act (round 6): the apply pass that writes dx ALSO reduces the three backward sums of the BatchNorm
# + PReLU that produced its input tensor (a stem in front of the first IBasicBlock): (M pixels, C, stride-2 compact add)
@pytest.mark.parametrize("s2", [False, True])
@pytest.mark.parametrize("shape", [(2 * 56 * 56, 64), (3 * 28 * 28, 64), (5 * 14 * 14, 128)])
def test_bn_backward_apply_reduces_the_sums_of_a_stem_in_front_of_it(shape, s2):
    needs_experiments()                   # (measured not faster, msml_amd/ops.py STEM_BWD_SUMS: not in the shipped library)
    m, c = shape
    n = {2 * 56 * 56: 2, 3 * 28 * 28: 3, 5 * 14 * 14: 5}[m]
    h = int(round((m // n) ** 0.5))
    g = torch.Generator().manual_seed(m + c + int(s2))
    rnd = lambda *s: torch.randn(*s, generator=g)       # noqa: E731
    dy = rnd(m, c).cuda().bfloat16()
    x = (rnd(m, c) * 1.3 + 0.2).cuda().bfloat16()                          # input of THIS BatchNorm = output of the stem's PReLU
    coef = torch.stack([torch.rand(c, generator=g) + 0.5, rnd(c) * 0.3, rnd(c) * 0.2, torch.rand(c, generator=g) + 0.5]).cuda()
    acc = (torch.randn(8, 3, c, generator=g, dtype=torch.float64) * (m ** 0.5) / 8).cuda()
    nx = (rnd(m, c) * 0.9 - 0.1).cuda().bfloat16()                         # saved input of the stem's BatchNorm
    ncoef = torch.stack([torch.rand(c, generator=g) + 0.5, rnd(c) * 0.4, rnd(c) * 0.2, torch.rand(c, generator=g) + 0.5]).cuda()
    nalpha = (torch.rand(c, generator=g) * 0.4).cuda()
    add = None
    ah = aw = 0
    if s2:                       # the compact gradient of a 1x1 / stride-2 downsample joins at the even pixels
        add = rnd(n * (h // 2) * (h // 2), c).cuda().bfloat16()
        ah = aw = h

    def run(fn, *extra):
        dx = torch.empty_like(x)
        pg = torch.zeros(3, c, device="cuda")
        nacc = torch.zeros(8, 3, c, dtype=torch.float64, device="cuda")
        _lib.call(fn, dy, x, coef[0], coef[1], None, coef[2], coef[3], acc, None, add, ah, aw, dx, None, pg[0], pg[1], None, 0,
                  m, c, nx, *extra, nacc, _lib.BF16)
        torch.cuda.synchronize()
        return dx, pg, nacc
    dx_a, pg_a, nacc_a = run("msml_bn_fin_bwd_apply", ncoef[2], ncoef[3])                                      # activation-free NEXT
    dx_b, pg_b, nacc_b = run("msml_bn_fin_bwd_apply_next_act", ncoef[0], ncoef[1], nalpha, ncoef[2], ncoef[3])
    assert torch.equal(dx_a, dx_b) and torch.equal(pg_a, pg_b)             # the apply itself is the same arithmetic
    # the three sums through the PReLU mask, from the dx the kernel wrote, in f64
    gq, xn = dx_b.double(), nx.double()
    z = nx.float() * ncoef[0] + ncoef[1]
    neg = z <= 0
    gp = torch.where(neg, gq * nalpha.double(), gq)
    xh = (nx.float() - ncoef[2]) * ncoef[3]
    want = torch.stack((gp.sum(0), (gp * xh.double()).sum(0), torch.where(neg, gq * z.double(), torch.zeros_like(gq)).sum(0)))
    got = nacc_b.sum(0)
    assert torch.allclose(got, want, rtol=2e-3, atol=2e-3 * want.abs().max().item()), (got - want).abs().max()
    assert float(nacc_a[:, 2].abs().max()) == 0.0 and float(nacc_b[:, 2].abs().max()) > 0.0
