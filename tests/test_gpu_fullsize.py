"""Full-size (BASELINE batch 256) checks through size-independent properties, where the CPU
oracle would take minutes: batch-split invariance (tiling must not leak between pixels),
linearity in the weights / in the batch, BatchNorm normalisation, and bf16 end-to-end
batch-composition independence in eval mode."""
import pytest
import torch

from msml_amd import _lib, functional as Fh, ops, synthetic
from msml_amd.backbones import MSML

pytestmark = pytest.mark.gpu
BF = _lib.BF16


def _rand(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda")


@pytest.mark.parametrize("cin,cout,h,stride", [(256, 256, 14, 1), (64, 64, 56, 1), (128, 128, 56, 2)])
def test_conv_full_size_properties(cin, cout, h, stride):
    n = 256
    x = _rand((n, h, h, cin), 1).bfloat16()
    w1 = _rand((cout, cin, 3, 3), 2) * 0.05
    w2 = _rand((cout, cin, 3, 3), 3) * 0.05
    wp1, wp2 = ops.pack_weight(w1, False, cin, 0, BF), ops.pack_weight(w2, False, cin, 0, BF)
    wps = ops.pack_weight(w1.bfloat16().float() + w2.bfloat16().float(), False, cin, 0, BF)
    y1, st = ops.conv2d(x, None, wp1, None, cout, 3, 3, stride, 1, 1, False, want_stats=True)
    # (1) batch-split invariance: bit-identical results for a sub-batch (different tile walk)
    ya, _ = ops.conv2d(x[64:192].contiguous(), None, wp1, None, cout, 3, 3, stride, 1, 1, False)
    assert torch.equal(y1[64:192], ya)
    # (2) linearity in the weights (f32 accumulation, bf16 rounding of the three outputs)
    y2, _ = ops.conv2d(x, None, wp2, None, cout, 3, 3, stride, 1, 1, False)
    ys, _ = ops.conv2d(x, None, wps, None, cout, 3, 3, stride, 1, 1, False)
    ref = y1.float() + y2.float()
    err = (ys.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-2, err
    # (3) epilogue statistics = statistics of the stored tensor (up to its bf16 rounding)
    s = st.sum(0).float()              # partial rows (f32) or the f64 accumulator of ops.ACC_STATS
    yf = y1.float().reshape(-1, cout)
    assert torch.allclose(s[0], yf.sum(0), rtol=2e-2, atol=2.0 * yf.abs().max().item())
    assert torch.allclose(s[1], (yf * yf).sum(0), rtol=2e-2)


def test_dgrad_wgrad_full_size_batch_linearity():
    n, c, h = 256, 256, 14
    x = _rand((n, h, h, c), 4).bfloat16()
    dy = _rand((n, h, h, c), 5).bfloat16()
    w = _rand((c, c, 3, 3), 6) * 0.05
    # weight gradient of the batch == sum over two half batches (f32 slabs, fixed order)
    dw = torch.empty_like(w)
    d1, d2 = torch.empty_like(w), torch.empty_like(w)
    ops.conv_wgrad(dy, x, dw, c, c, c, 0, 3, 3, 1, 1, 1)
    ops.conv_wgrad(dy[:128].contiguous(), x[:128].contiguous(), d1, c, c, c, 0, 3, 3, 1, 1, 1)
    ops.conv_wgrad(dy[128:].contiguous(), x[128:].contiguous(), d2, c, c, c, 0, 3, 3, 1, 1, 1)
    err = (dw - (d1 + d2)).abs().max().item() / dw.abs().max().item()
    assert err < 1e-4, err
    # backward-data: sub-batch bit-identical
    wpt = ops.pack_weight(w, True, c, 0, BF)
    dx, _ = ops.conv2d(dy, None, wpt, None, c, 3, 3, 1, 1, 1, True, p=h, q=h)
    dxa, _ = ops.conv2d(dy[100:164].contiguous(), None, wpt, None, c, 3, 3, 1, 1, 1, True, p=h, q=h)
    assert torch.equal(dx[100:164], dxa)


def test_bn_train_full_size_normalises():
    """Training-mode BatchNorm at 256 x 56 x 56 x 64: per-channel mean of the output == beta and
    variance == gamma^2 (the defining property), running stats move by momentum 0.1."""
    bn = torch.nn.BatchNorm2d(64, eps=1e-5).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(1.0 + 0.1 * _rand((64,), 7))
        bn.bias.copy_(0.1 * _rand((64,), 8))
    x = (_rand((256, 56, 56, 64), 9) * 3.0 + 1.5).bfloat16()
    y = Fh.bn_act(x, None, bn).float().reshape(-1, 64)
    mean, var = y.mean(0), y.var(0, unbiased=False)
    assert torch.allclose(mean, bn.bias, atol=2e-2)
    assert torch.allclose(var, bn.weight ** 2, rtol=3e-2)
    xf = x.float().reshape(-1, 64)
    assert torch.allclose(bn.running_mean, 0.1 * xf.mean(0), rtol=1e-2, atol=1e-3)
    assert int(bn.num_batches_tracked) == 1


def test_eval_batch_composition_independence_bf16():
    """ires50 bf16 eval at batch 256: the embedding / mask of an image does not depend on what
    else is in the batch (bit-identical to a batch of 8 containing it)."""
    torch.manual_seed(0)
    peer = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
    m = MSML("iresnet50", "unet", (1, 1, 1, 1), 8, fp16=True, fm_params=(3, 2, "sigmoid", "mul"),
             header_type="AMArcFace", peer_params=peer).cuda().eval()
    x = synthetic.images(256, 3)
    x, _ = synthetic.rect_occlusion(x, 3)
    x = x.cuda()
    with torch.no_grad():
        f, seg = m(x)
        f8, seg8 = m(x[120:128].contiguous())
    assert torch.equal(f[120:128], f8)
    assert torch.equal(seg[120:128], seg8)
    assert torch.isfinite(f).all()
