"""End-to-end GPU parity of msml_amd.MSML against the golden vectors recorded from the
reference (and against the CPU oracle for per-op pieces).  f32 mode: embeddings within 1e-3 rel
(north_star) -- in practice ~1e-5 -- and occlusion-mask indices bit-exact."""
import numpy as np
import pytest
import torch

from msml_amd import functional as Fh
from msml_amd import synthetic
from msml_amd.backbones import MSML
from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
from oracle import model as om
from oracle.fill import fill_module
from oracle.inputs import eval_inputs, head_inputs, refinit_frb_convs, seg_inputs
from tests.helpers import assert_cs, checksum, elem_err, load, pick, rel_err

pytestmark = pytest.mark.gpu
PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}


def hip_msml(frb, C=1000, fp16=False):
    torch.manual_seed(0)
    m = MSML(frb, "unet", (1, 1, 1, 1), C, fp16=fp16, fm_params=(3, 2, "sigmoid", "mul"),
             header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0), peer_params=dict(PEER_OFF))
    return fill_module(m).cuda()


def eval_check(frb, fname, bs, fp16, feat_tol, exact_mask, precision=None):
    g = load(fname)
    m = hip_msml(frb, fp16=fp16).eval()
    if precision is not None:
        m.eval_precision = precision
    x, _ = eval_inputs(bs)
    with torch.no_grad():
        feat, final_seg = m(x.cuda())
    torch.cuda.synchronize()
    err = rel_err(feat.cpu().numpy(), g["feature"])
    assert err < feat_tol, err
    bits = np.packbits(Fh.mask_index(final_seg).cpu().numpy().reshape(-1))
    mism = int(np.unpackbits(bits ^ g["mask_bits"]).sum())
    if exact_mask:
        assert mism == 0, mism
        assert_cs(final_seg, g["final_seg_cs"], 1e-4, "final_seg")
    else:
        assert mism < 0.02 * bs * 112 * 112, mism
    return err, mism


def test_eval_ires18_f32():
    eval_check("iresnet18", "g1_ires18_eval.npz", 4, False, 1e-3, True)


def test_eval_ires50_f32():
    eval_check("iresnet50", "g2_ires50_eval.npz", 2, False, 1e-3, True)


def test_eval_ires100_f32():
    eval_check("iresnet100", "g2_ires100_eval.npz", 2, False, 1e-3, True)


@pytest.mark.parametrize("frb,fname,bs", [("iresnet18", "g1_ires18_eval.npz", 4),
                                          ("iresnet50", "g2_ires50_eval.npz", 2),
                                          ("iresnet100", "g2_ires100_eval.npz", 2)])
def test_eval_fp16_default_meets_the_parity_bar(frb, fname, bs):
    """fp16=True inference in its DEFAULT precision (split-bf16, 'bf16x3': three bf16 MFMAs per product)
    against the reference's f32 goldens: embedding within 1e-3 rel (north_star), occlusion-mask indices
    bit-exact, final_seg checksums within 1e-4 -- the same bar the exact-f32 mode is held to."""
    m = hip_msml(frb, fp16=True)
    assert m.eval_precision == "bf16x3"
    err, mism = eval_check(frb, fname, bs, True, 1e-3, True)
    print("bf16x3 eval %s: feature rel err %.3e, mask mismatches %d" % (frb, err, mism))
    assert err < 2e-4            # measured ~1e-5: keep an order of magnitude of margin visible


def test_eval_fp16_bn1_fold_against_the_separate_pass(monkeypatch):
    """Split-bf16 inference folds IBasicBlock's eval-mode bn1 into conv1 (functional.bn_conv_bn_eval_x3: operand scaled per
    input channel + a shift per border class of the output pixel, backbones/frb/iresnet.py:58-60).  Against the path that
    keeps bn1 as a pass of its own (X3_FOLD_BN1 off): same embeddings to split-bf16 rounding, same mask indices, and both
    inside the golden's bar."""
    m = hip_msml("iresnet50", fp16=True).eval()
    x, _ = eval_inputs(2)
    with torch.no_grad():
        assert Fh.X3_FOLD_BN1
        f1, s1 = m(x.cuda())
        monkeypatch.setattr(Fh, "X3_FOLD_BN1", False)
        f0, s0 = m(x.cuda())
    assert rel_err(f1.float().cpu().numpy(), f0.float().cpu().numpy()) < 1e-4
    assert torch.equal(Fh.mask_index(s1), Fh.mask_index(s0))
    assert rel_err(s1.float().cpu().numpy(), s0.float().cpu().numpy()) < 1e-4


@pytest.mark.parametrize("fp16", [False, True])
def test_eval_ires34_vs_oracle(fp16):
    """iresnet34 ([3, 4, 6, 3], the third FRB the reference's MSML constructs: backbones/msml.py:106-108) has no
    recorded golden: compare with the CPU oracle (itself pinned to the reference on ires18 / 50 / 100) run here on
    the same fill and inputs -- f32 and the fp16=True default (bf16x3) to the same bar as the goldens."""
    torch.manual_seed(0)
    ref = fill_module(om.MSML("iresnet34", "unet", (1, 1, 1, 1), 1000, fm_params=(3, 2, "sigmoid", "mul"),
                              header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0))).eval()
    m = hip_msml("iresnet34", fp16=fp16).eval()
    x, _ = eval_inputs(2)
    with torch.no_grad():
        rf, rseg = ref(x)
        feat, seg = m(x.cuda())
    err = rel_err(feat.cpu().numpy(), rf.numpy())
    assert err < 1e-3, err
    assert torch.equal(Fh.mask_index(seg).cpu().long(), rseg.float().max(dim=1)[1])       # train.py:357
    assert rel_err(seg.float().cpu().numpy(), rseg.numpy()) < 1e-4


def test_eval_ires18_bf16_fast_mode():
    """eval_precision='bf16' (plain bf16 operands, the throughput mode): documented tolerance -- it does
    NOT meet the 1e-3 bar (DESIGN.md 'precision modes'), which is why it is not the default."""
    err, mism = eval_check("iresnet18", "g1_ires18_eval.npz", 4, True, 5e-2, False, precision="bf16")
    print("bf16 eval: feature rel err %.3e, mask mismatches %d" % (err, mism))


def test_eval_bf16_fused_epilogue_matches_unfused():
    """Inference folds eval BatchNorm / PReLU / residual into the conv epilogue (no_grad); with
    autograd enabled the unfused conv -> bn_act kernels run.  Same arithmetic up to bf16 rounding
    of the intermediate tensors."""
    m = hip_msml("iresnet50", fp16=True).eval()
    m.eval_precision = "bf16"
    x, _ = eval_inputs(4)
    with torch.no_grad():
        f1, s1 = m(x.cuda())
    with torch.enable_grad():
        f2, s2 = m(x.cuda())
    err = rel_err(f1.float().cpu().numpy(), f2.detach().float().cpu().numpy())
    assert err < 3e-2, err
    a, b = Fh.mask_index(s1), Fh.mask_index(s2.detach())
    assert (a != b).float().mean().item() < 0.02
    g = load("g2_ires50_eval.npz")
    x2, _ = eval_inputs(2)
    with torch.no_grad():
        f3, _ = m(x2.cuda())
    assert rel_err(f3.float().cpu().numpy(), g["feature"]) < 5e-2


def test_state_dict_roundtrip_with_oracle():
    m = hip_msml("iresnet18", 10)
    o = om.MSML("iresnet18", num_classes=10, header_type="AMArcFace")
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, strict=True)
    m.load_state_dict(o.state_dict(), strict=True)
    names = [n for n, _ in m.named_parameters()]
    assert any("osb" in n for n in names) and any("classification" in n for n in names)
    assert any("fm_ops" in n for n in names)


def test_seg_loss_g7():
    g = load("g7_seg_loss.npz")
    logit, msk = seg_inputs()
    crit = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")
    lg = logit.cuda().requires_grad_(True)
    loss = crit(lg, msk.cuda(), msk.cuda())
    loss.backward()
    assert abs(loss.item() - g["loss"]) < 1e-5 * abs(g["loss"])
    assert rel_err(pick(lg.grad, 256), g["grad_pick"]) < 1e-4
    assert_cs(lg.grad, g["grad_cs"], 1e-4)
    lg = logit.cuda().requires_grad_(True)
    clean = torch.ones_like(msk).cuda()
    loss = crit(lg, clean, clean)
    loss.backward()
    assert abs(loss.item() - g["loss_clean"]) < 1e-5 * abs(g["loss_clean"])
    assert_cs(lg.grad, g["grad_clean_cs"], 1e-4)
    # the reference's other reductions (consensus_loss.py:42-57: 'all' = H * W / N * H * W)
    for rp, rk in (("all", "idx"), ("idx", "all"), ("all", "all")):
        lg = logit.cuda().requires_grad_(True)
        loss = StructureConsensuLossFunction(10.0, 5.0, rp, rk)(lg, msk.cuda(), msk.cuda())
        loss.backward()
        assert abs(loss.item() - g["loss_%s_%s" % (rp, rk)]) < 1e-5 * abs(g["loss_%s_%s" % (rp, rk)]), (rp, rk)
        # with reduce_pixel='all' the reference's OWN gradient is NaN on images that lack a blob (0 * inf in the autograd
        # of log(0), the loss term itself is masked to 0): the kernel returns the finite limit (0) there
        ref, got = g["grad_pick_%s_%s" % (rp, rk)], pick(lg.grad, 256)
        ok = ~np.isnan(ref)
        assert torch.isfinite(lg.grad).all()
        assert rel_err(got[ok], ref[ok]) < 1e-4, (rp, rk)
        if ok.all():
            assert_cs(lg.grad, g["grad_cs_%s_%s" % (rp, rk)], 1e-4)


def test_seg_loss_refuses_blobs_that_differ_from_target_on_every_call():
    """blobs must equal target (train.py:258 passes a clone).  The first call checks on the host and raises; every
    later call checks on the device without a synchronisation and turns loss AND gradient into NaN on a mismatch
    (the first-call-only check left later callers with a silently wrong loss)."""
    logit, msk = seg_inputs()
    msk = msk.cuda()
    other = 1 - msk
    crit = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")
    with pytest.raises(NotImplementedError):
        crit(logit.cuda(), other, msk)
    crit = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")
    lg = logit.cuda().requires_grad_(True)
    ref = crit(lg, msk.clone(), msk)                      # call 1: equal, host check passes
    lg2 = logit.cuda().requires_grad_(True)
    ok = crit(lg2, msk.clone(), msk)                      # call 2: equal clone, device check
    ok.backward()
    assert torch.equal(ok.detach(), ref.detach()) and torch.isfinite(lg2.grad).all()
    lg3 = logit.cuda().requires_grad_(True)
    bad = crit(lg3, other, msk)                           # call 3: differs -> poisoned, no exception, no sync
    bad.backward()
    assert torch.isnan(bad).item() and torch.isnan(lg3.grad).all()
    with pytest.raises(NotImplementedError):
        crit(logit.cuda(), msk[:, :50], msk)              # a shape mismatch is a host-side fact: raises at once


def test_heads_g5():
    from msml_amd.headers import AMArcFace, AMCosFace, Softmax
    g = load("g5_heads.npz")
    emb, w, label = head_inputs()
    for name, cls, prm in (("arc0", AMArcFace, (64.0, 0.48, 0.0, 0.0)),
                           ("arc1", AMArcFace, (64.0, 0.5, 1.2, 0.1)),
                           ("cos0", AMCosFace, (64.0, 0.4, 0.0, 0.0)),
                           ("cos1", AMCosFace, (64.0, 0.4, 1.2, 0.1))):
        h = cls(512, 8, None, *prm).cuda()
        with torch.no_grad():
            h.weight.copy_(w)
        e = emb.cuda().requires_grad_(True)
        out = h(e, label.cuda())
        out.backward(torch.linspace(-1, 1, out.numel()).reshape(out.shape).cuda())
        assert rel_err(out.detach().cpu().numpy(), g[name + "_out"]) < 1e-5, name
        assert rel_err(e.grad.cpu().numpy(), g[name + "_demb"]) < 1e-4, name
        assert rel_err(h.weight.grad.cpu().numpy(), g[name + "_dw"]) < 1e-4, name
    h = Softmax(512, 8, None).cuda()
    with torch.no_grad():
        h.weight.copy_(w)
        h.bias.copy_(torch.linspace(-0.5, 0.5, 8))
    assert rel_err(h(emb.cuda(), label.cuda()).detach().cpu().numpy(), g["softmax_out"]) < 1e-5


def run_train_step(m, bs, C):
    x, msk = eval_inputs(bs)
    label = synthetic.labels(bs, C, seed=1)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.1 / 512 * bs, momentum=0.9, weight_decay=5e-4)
    final_cls, final_seg, kd = m(x.cuda(), label.cuda(), None)
    seg_loss = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")(final_seg, msk.cuda(), msk.cuda())
    cls_loss = torch.nn.functional.cross_entropy(final_cls, label.cuda())
    total = cls_loss + seg_loss
    total.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), 5, 2)
    return opt, final_cls, seg_loss, cls_loss, gnorm


@pytest.mark.parametrize("variant", ["fill", "refinit"])
def test_train_step_g4(variant):
    """One full training step (train-mode BN, CE + consensus seg loss, clip, SGD) in f32 mode
    against the reference's golden: losses, selected gradients, updated running statistics."""
    g = load("g4_train_%s.npz" % variant)
    m = hip_msml("iresnet18", 1000)
    if variant == "refinit":
        refinit_frb_convs(m)
    opt, final_cls, seg_loss, cls_loss, gnorm = run_train_step(m, 4, 1000)
    tol = 1e-3
    assert abs(seg_loss.item() - g["seg_loss"]) < tol * abs(g["seg_loss"])
    assert abs(cls_loss.item() - g["cls_loss"]) < tol * abs(g["cls_loss"])
    assert abs(float(gnorm) - g["grad_norm"]) < 5e-3 * abs(g["grad_norm"])
    assert_cs(final_cls, g["final_cls_cs"], tol, "final_cls")
    params = dict(m.named_parameters())
    worst, worst_el = 0.0, 0.0
    for key in g.files:
        if key.startswith("grad_pick/"):
            n = key.split("/", 1)[1]
            got = pick(params[n].grad, 32)
            if n == "frb.fc.bias":
                # followed by train-mode BatchNorm1d: the exact gradient is 0, both sides hold
                # rounding noise only
                assert np.abs(got).max() < 1e-5 and np.abs(g[key]).max() < 1e-5
                continue
            e, ee = rel_err(got, g[key]), elem_err(got, g[key])
            worst, worst_el = max(worst, e), max(worst_el, ee)
            assert e < 1e-2 and ee < 1e-2, (n, e, ee)       # norm-wise AND worst single element
    print("train step %s: worst picked-grad rel err %.3e (norm-wise), %.3e (element-wise)" % (variant, worst, worst_el))
    opt.step()
    for key in g.files:
        if key.startswith("stat/"):
            n = key.split("/", 1)[1]
            assert rel_err(m.state_dict()[n].cpu().numpy(), g[key]) < 1e-3, n


def test_partial_fc_w1_g6():
    """PartialFC.forward_backward on the HIP backend, W = 1, against the reference's golden."""
    from msml_amd.headers import ArcMargin, PartialFC
    from oracle.inputs import PFC_B, PFC_C, PFC_E, pfc_inputs
    g = load("g6_partial_fc.npz")
    feat, label, w = pfc_inputs(1, 0)
    p = PartialFC(0, 0, 1, PFC_B, False, ArcMargin(64.0, 0.48, 0.0, 0.0), PFC_C,
                  embedding_size=PFC_E)
    with torch.no_grad():
        p.weight.copy_(w)
    opt = torch.optim.SGD([{"params": p.parameters()}], lr=0.1 / 512 * PFC_B, momentum=0.9,
                          weight_decay=5e-4)
    x_grad, loss_v = p.forward_backward(label.cuda(), feat.cuda(), opt)
    assert abs(loss_v.item() - g["w1/r0/loss"]) < 1e-4 * abs(g["w1/r0/loss"])
    assert rel_err(x_grad.cpu().numpy(), g["w1/r0/x_grad"]) < 1e-4
    assert rel_err(pick(p.sub_weight.grad, 256), g["w1/r0/wgrad_pick"]) < 1e-4
    opt.step()
    assert rel_err(pick(p.sub_weight.data, 256), g["w1/r0/wnew_pick"]) < 1e-5


def test_flat_sgd_matches_torch():
    """FlatSGD (fused clip + momentum SGD on the flat arena) == clip_grad_norm_ + torch SGD."""
    from msml_amd.optim import FlatSGD
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in [(7, 5), (33,), (4, 3, 3, 3)]]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    fo = FlatSGD([{"params": ps[:2], "lr": 0.1}, {"params": ps[2:], "lr": 0.01}], 0.9, 5e-4, 5.0)
    to = torch.optim.SGD([{"params": qs[:2], "lr": 0.1}, {"params": qs[2:], "lr": 0.01}],
                         lr=0.1, momentum=0.9, weight_decay=5e-4)
    for step in range(3):
        fo.zero_grad()
        to.zero_grad()
        gs = [torch.randn_like(p) * 3 for p in ps]
        for p, q, gr in zip(ps, qs, gs):
            p.grad.add_(gr)
            q.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(qs, 5.0, 2)
        fo.step()
        to.step()
        for p, q in zip(ps, qs):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), step


def test_inplace_grads_match_autograd():
    """FlatSGD's in-place gradient accumulation (kernels write into the arena views, autograd
    gets None) gives the same gradients as the standard autograd path."""
    from msml_amd import ops
    from msml_amd.optim import FlatSGD, reference_param_groups
    x, msk = eval_inputs(2)
    label = synthetic.labels(2, 50, seed=1)

    def grads(inplace):
        m = hip_msml("iresnet18", 50).train()
        opt = None
        if inplace:
            opt = FlatSGD(reference_param_groups(m, 2, 1), 0.9, 5e-4, 5.0)
            opt.zero_grad()
        else:
            pass
        cls, seg, _ = m(x.cuda(), label.cuda())
        loss = torch.nn.functional.cross_entropy(cls, label.cuda()) + \
            StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    a = grads(False)
    b = grads(True)
    assert set(a) == set(b)
    for n in a:
        assert torch.allclose(a[n], b[n], rtol=1e-5, atol=1e-7), n


def test_partial_fc_bf16_vs_oracle():
    """bf16 PartialFC (split-K dX GEMM, fused softmax passes) against the CPU oracle backend at a
    shape with many classes; tolerance = bf16 operand rounding."""
    from msml_amd.headers import ArcMargin, PartialFC
    from tests.pfc_cpu_backend import OracleBackend
    torch.manual_seed(0)
    N, C, E = 32, 10000, 512
    feat = torch.nn.functional.normalize(torch.randn(N, E))
    label = torch.randint(0, C, (N,))
    w = torch.randn(C, E) * 0.01
    ref = PartialFC(0, 0, 1, N, False, ArcMargin(64.0, 0.48, 0, 0), C, backend=OracleBackend(),
                    device=torch.device("cpu"))
    ref.weight.copy_(w)
    xg_r, loss_r = ref.forward_backward(label, feat, None)
    p = PartialFC(0, 0, 1, N, False, ArcMargin(64.0, 0.48, 0, 0), C, fp16=True)
    with torch.no_grad():
        p.weight.copy_(w)
    xg, loss = p.forward_backward(label.cuda(), feat.cuda(), None)
    assert abs(loss.item() - loss_r.item()) < 1e-3 * abs(loss_r.item())
    assert rel_err(xg.cpu().numpy(), xg_r.numpy()) < 1e-2
    assert rel_err(p.sub_weight.grad.cpu().numpy(), ref.sub_weight.grad.numpy()) < 1e-2


def test_stream_wait_stream_orders_the_waiter_behind_the_source():
    """msml_stream_wait_stream (the fork of the weight-gradient stream, one re-recorded event per thread): work queued on the
    waiter afterwards sees everything queued on the source before -- twice, so that the second record of the same event is
    exercised while the first wait may still be pending."""
    from msml_amd import _lib
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 26, device="cuda")
    torch.cuda.synchronize()
    seen = []
    for rnd in range(2):
        with torch.cuda.stream(a):
            for _ in range(20):
                x.add_(1.0)
        _lib.call("msml_stream_wait_stream", b.cuda_stream, a.cuda_stream)
        with torch.cuda.stream(b):
            seen.append((x.min(), x.max()))
            _lib.call("msml_stream_wait_stream", a.cuda_stream, b.cuda_stream)      # (a must not overwrite x under b's read)
    torch.cuda.synchronize()
    assert [(float(lo), float(hi)) for lo, hi in seen] == [(20.0, 20.0), (40.0, 40.0)]


def test_side_streams_match_serial():
    """Eager multi-stream issue (OSB on its own stream, weight gradients on a second stream)
    gives bit-identical gradients to the single-stream order (same kernels, same inputs)."""
    from msml_amd import ops
    from msml_amd.optim import FlatSGD, reference_param_groups
    x, msk = eval_inputs(2)
    label = synthetic.labels(2, 50, seed=1)

    def grads(streams):
        m = hip_msml("iresnet18", 50, fp16=True).train()
        opt = FlatSGD(reference_param_groups(m, 2, 1), 0.9, 5e-4, 5.0)
        ops.WGRAD_STREAM = torch.cuda.Stream() if streams else None
        ops.OSB_STREAM = torch.cuda.Stream() if streams else None
        try:
            for _ in range(2):
                opt.zero_grad()
                cls, seg, _ = m(x.cuda(), label.cuda())
                loss = torch.nn.functional.cross_entropy(cls, label.cuda()) + \
                    StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
                loss.backward()
                ops.wgrad_stream_join()
            torch.cuda.synchronize()
            return opt.flat_g.clone()
        finally:
            ops.WGRAD_STREAM = None
            ops.OSB_STREAM = None
    a = grads(False)
    b = grads(True)
    assert torch.equal(a, b)


def test_stem_backward_sums_from_the_first_block_match_the_stems_own_reduce(monkeypatch):
    """Round 6 (opt-in, ops.STEM_BWD_SUMS / MSML_STEM_BWD_SUMS=1: measured not faster, ops.py): the first IBasicBlock's bn1
    apply kernel reduces the stem BatchNorm's three backward sums (through the stem's PReLU mask) while it writes that
    BatchNorm's output gradient (msml_bn_fin_bwd_apply_next_act); the stem's own backward is then an apply pass.  Against the step with the stem's own reduce pass (ops.STEM_BWD_SUMS off): every
    gradient the same to the order of the f32 partial sums -- the stems' conv / BatchNorm / PReLU gradients included --
    and the FRB's stem (112 x 112) really took the short path.  (The OSB's stem output feeds layer1 AND gcm1 -- a fan-out,
    functional.fanout2 -- so the first OSB block sees only part of its gradient and that stem keeps its own reduce.)"""
    from msml_amd import _lib as _l
    from msml_amd import ops
    from msml_amd.optim import FlatSGD, reference_param_groups
    if not _l.value("msml_has_experiments"):
        pytest.skip("kernel variant of an experiment build (tools/experiments_run.sh)")
    x, msk = eval_inputs(8)
    label = synthetic.labels(8, 50, seed=1)

    def grads(on):
        monkeypatch.setattr(ops, "STEM_BWD_SUMS", on)
        m = hip_msml("iresnet18", 50, fp16=True).train()
        opt = FlatSGD(reference_param_groups(m, 8, 1), 0.9, 5e-4, 5.0)
        hits0 = ops.COUNTERS["bn3_partial_hits"]
        try:
            opt.zero_grad()
            cls, seg, _ = m(x.cuda(), label.cuda())
            loss = torch.nn.functional.cross_entropy(cls, label.cuda()) + \
                StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
            loss.backward()
            ops.wgrad_stream_join()
            torch.cuda.synchronize()
            return {n: p.grad.detach().float().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}, \
                ops.COUNTERS["bn3_partial_hits"] - hits0
        finally:
            opt.release()
    a, hits_a = grads(False)
    b, hits_b = grads(True)
    assert hits_b == hits_a + 1                                  # the FRB's stem
    for n in a:
        assert rel_err(b[n], a[n]) < 2e-3, (n, rel_err(b[n], a[n]))
    for n in ("frb.conv1.weight", "frb.bn1.weight", "frb.bn1.bias", "frb.prelu.weight", "osb.conv1.weight", "osb.bn1.weight",
              "osb.prelu.weight"):
        assert n in a and np.abs(a[n]).max() > 0, n


def test_side_streams_switched_on_mid_run():
    """Serial steps first, then the side streams are switched on (what bench.py does): the scratch
    workspaces grow during the first multi-stream steps, and a buffer dropped on growth must not be
    recycled under kernels of the weight-gradient stream (it once was: NaN gradients).  The OSB
    backward is issued first here (its own backward call), the order that exposed it."""
    from msml_amd import ops
    from msml_amd.optim import FlatSGD, reference_param_groups
    x, msk = eval_inputs(8)
    label = synthetic.labels(8, 50, seed=1)

    def run(streams_after):
        ops._WS.clear()
        m = hip_msml("iresnet18", 50, fp16=True).train()
        opt = FlatSGD(reference_param_groups(m, 8, 1), 0.9, 5e-4, 5.0)
        try:
            for it in range(4):
                if it == streams_after:
                    ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
                opt.zero_grad()
                cls, seg, _ = m(x.cuda(), label.cuda())
                seg_loss = StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
                seg_loss.backward()
                torch.nn.functional.cross_entropy(cls, label.cuda()).backward()
                ops.wgrad_stream_join()
                opt.step()
            torch.cuda.synchronize()
            return opt.flat_w.clone()
        finally:
            ops.WGRAD_STREAM = None
            ops.OSB_STREAM = None
    a = run(99)
    b = run(2)
    assert torch.isfinite(b).all()
    assert torch.equal(a, b)


def test_overlapped_allreduce_one_rank():
    """Bucketed gradient all-reduce overlapped with backward (RCCL, one-rank communicator):
    every bucket fires exactly once and the gradients equal the non-overlapped path."""
    import os
    import torch.distributed as dist
    from msml_amd import ops
    from msml_amd.optim import FlatSGD, reference_param_groups
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
    x, msk = eval_inputs(2)
    label = synthetic.labels(2, 50, seed=1)

    def grads(overlap):
        m = hip_msml("iresnet18", 50, fp16=True).train()
        opt = FlatSGD(reference_param_groups(m, 2, 1), 0.9, 5e-4, 5.0)
        if overlap:
            opt.enable_overlap(1, bucket_bytes=8 << 20)
        ops.WGRAD_STREAM = torch.cuda.Stream()
        ops.OSB_STREAM = torch.cuda.Stream()
        try:
            for _ in range(2):
                opt.zero_grad()
                cls, seg, _ = m(x.cuda(), label.cuda())
                loss = torch.nn.functional.cross_entropy(cls, label.cuda()) + \
                    StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
                loss.backward()
                if overlap:
                    assert sum(opt.fired) >= len(opt.buckets) - 1     # fired during backward
                    opt.all_reduce_grads(1)
                else:
                    ops.wgrad_stream_join()
            torch.cuda.synchronize()
            return opt.flat_g.clone(), (len(opt.buckets) if overlap else 0)
        finally:
            ops.WGRAD_STREAM = None
            ops.OSB_STREAM = None
    a, _ = grads(False)
    b, nb = grads(True)
    assert nb >= 4
    assert torch.equal(a, b)
    dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [(4, 64, 64, 14, 1), (3, 64, 128, 14, 2), (128, 256, 256, 14, 1), (5, 32, 64, 9, 2)])
def test_block_function_matches_op_graph(cfg):
    """IBasicBlock as ONE autograd node (blocks.py: BatchNorm backward sums from the dgrad epilogue,
    gradient join inside bn1's apply) against the op-by-op graph of functional.py: same forward
    bits, gradients equal up to the order of the f32 partial sums / one bf16 rounding of the join."""
    import copy
    from torch import nn
    from msml_amd import ops
    from msml_amd.backbones.frb.iresnet import IBasicBlock, conv1x1
    n, cin, cout, h, stride = cfg
    torch.manual_seed(sum(cfg))
    down = None
    if stride != 1 or cin != cout:
        down = nn.Sequential(conv1x1(cin, cout, stride), nn.BatchNorm2d(cout, eps=1e-05))
    blk = IBasicBlock(cin, cout, stride, down)
    for p in blk.parameters():
        if p.dim() == 1:
            nn.init.uniform_(p, 0.5, 1.5)
        else:
            nn.init.normal_(p, 0, (1.0 / (p.shape[1] * 9)) ** 0.5)
    nn.init.uniform_(blk.prelu.weight, 0.1, 0.4)
    blk = blk.cuda().train()
    x0 = ops.to_nhwc(torch.randn(n, cin, h, h).cuda(), 1)
    dout = None
    res = []
    for use_fn in (False, True):
        b = copy.deepcopy(blk)
        x = x0.clone().requires_grad_(True)
        old = ops.BLOCK_FUNCTION
        ops.BLOCK_FUNCTION = use_fn
        try:
            y = b(x)
            if dout is None:
                dout = torch.randn_like(y)
            y.backward(dout)
        finally:
            ops.BLOCK_FUNCTION = old
        torch.cuda.synchronize()
        res.append((y.detach().float(), x.grad.float(), {k: v.grad.float() for k, v in b.named_parameters()},
                    {k: v.clone() for k, v in b.named_buffers()}))
    (y1, dx1, g1, buf1), (y2, dx2, g2, buf2) = res
    assert torch.equal(y1, y2)
    for k in buf1:
        assert torch.equal(buf1[k], buf2[k]), k
    assert (dx1 - dx2).abs().max().item() <= 2e-2 * dx1.abs().max().item()
    assert rel_err(dx2.cpu().numpy(), dx1.cpu().numpy()) < 5e-3
    for k in g1:
        assert rel_err(g2[k].cpu().numpy(), g1[k].cpu().numpy()) < 1e-2, k


@pytest.mark.parametrize("cfg", [(22, 64, 64, 56), (5, 128, 128, 28), (7, 256, 256, 14)])
def test_block_bn_inside_conv_is_bit_neutral(cfg):
    """bn1 / bn2(+PReLU) applied inside conv1 / conv2 and their weight-gradient kernels
    (ops.FUSE_BN_IN, msml_conv2d_bnin) against the same block with the activations materialised:
    every output, gradient and running statistic bit-identical."""
    import copy
    from torch import nn
    from msml_amd import ops
    from msml_amd.backbones.frb.iresnet import IBasicBlock
    n, cin, cout, h = cfg
    from msml_amd import _lib as _l
    if cin == 64 and not _l.value("msml_has_experiments"):
        pytest.skip("the weights-stationary 64-channel kernel takes an input transform in experiment builds only "
                    "(tools/experiments_run.sh)")
    torch.manual_seed(sum(cfg))
    blk = IBasicBlock(cin, cout, 1, None)
    for p in blk.parameters():
        if p.dim() == 1:
            nn.init.uniform_(p, 0.5, 1.5)
        else:
            nn.init.normal_(p, 0, (1.0 / (p.shape[1] * 9)) ** 0.5)
    nn.init.uniform_(blk.prelu.weight, 0.1, 0.4)
    blk = blk.cuda().train()
    x0 = ops.to_nhwc(torch.randn(n, cin, h, h).cuda(), 1)
    from msml_amd import _lib
    assert _lib.value("msml_conv2d_bnin_applies", cin, cout, n, h, h, h, h, 3, 3, 1, 1, 1, 1) == 1
    dout = None
    res = []
    for fuse in (False, True):
        b = copy.deepcopy(blk)
        x = x0.clone().requires_grad_(True)
        old = ops.FUSE_BN_IN
        ops.FUSE_BN_IN = fuse
        ops._BNIN_OK.clear()
        try:
            y = b(x)
            if dout is None:
                dout = torch.randn_like(y)
            y.backward(dout)
        finally:
            ops.FUSE_BN_IN = old
            ops._BNIN_OK.clear()
        torch.cuda.synchronize()
        res.append((y.detach().clone(), x.grad.clone(), {k: v.grad.clone() for k, v in b.named_parameters()},
                    {k: v.clone() for k, v in b.named_buffers()}))
    (y1, dx1, g1, buf1), (y2, dx2, g2, buf2) = res
    if cin == 64:
        # round 5: the materialised side's 64 -> 64 convs run on k_conv_s2r, the in-LDS side's on k_conv_ws (the only kernel
        # with the input transform).  The conv outputs are bit-identical, but the f32 per-workgroup partial sums of the
        # BatchNorm statistics are cut differently, so the coefficients may differ in their last bit: equal to bf16 rounding
        from tests.helpers import rel_err as _re
        assert _re(y2.float().cpu().numpy(), y1.float().cpu().numpy()) < 2e-3
        assert _re(dx2.float().cpu().numpy(), dx1.float().cpu().numpy()) < 2e-3
        for k in g1:
            assert _re(g2[k].float().cpu().numpy(), g1[k].float().cpu().numpy()) < 5e-3, k
        for k in buf1:
            assert torch.allclose(buf1[k].float(), buf2[k].float(), rtol=1e-5, atol=1e-6), k
        return
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    for k in buf1:
        assert torch.equal(buf1[k], buf2[k]), k


@pytest.mark.parametrize("cfg", [(4, 64, 128, 14), (3, 64, 64, 9), (2, 64, 64, 56), (3, 128, 256, 28)])
def test_block_compact_downsample_gradient(cfg):
    """First block of a stage: the input gradient of the 1x1 / stride-2 downsample conv stays compact
    and is scatter-added by bn1's apply kernel (msml_bn_act_bwd_apply_s2) -- same bits as the dense
    gradient + dense add (odd map sizes included)."""
    import copy
    from torch import nn
    from msml_amd import ops
    from msml_amd.backbones.frb.iresnet import IBasicBlock, conv1x1
    n, cin, cout, h = cfg
    torch.manual_seed(sum(cfg))
    down = nn.Sequential(conv1x1(cin, cout, 2), nn.BatchNorm2d(cout, eps=1e-05))
    blk = IBasicBlock(cin, cout, 2, down)
    for p in blk.parameters():
        if p.dim() == 1:
            nn.init.uniform_(p, 0.5, 1.5)
        else:
            nn.init.normal_(p, 0, (1.0 / (p.shape[1] * 9)) ** 0.5)
    blk = blk.cuda().train()
    x0 = ops.to_nhwc(torch.randn(n, cin, h, h).cuda(), 1)
    dout = None
    res = []
    for sparse in (False, True):
        b = copy.deepcopy(blk)
        x = x0.clone().requires_grad_(True)
        old = ops.SPARSE_DOWNSAMPLE_GRAD
        ops.SPARSE_DOWNSAMPLE_GRAD = sparse
        try:
            y = b(x)
            if dout is None:
                dout = torch.randn_like(y)
            y.backward(dout)
        finally:
            ops.SPARSE_DOWNSAMPLE_GRAD = old
        torch.cuda.synchronize()
        res.append((x.grad.clone(), {k: v.grad.clone() for k, v in b.named_parameters()}))
    (dx1, g1), (dx2, g2) = res
    assert torch.isfinite(dx2.float()).all()
    assert torch.equal(dx1, dx2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize("cfg", [(6, 64, 28), (4, 128, 14), (3, 256, 14), (5, 512, 7)])
def test_fm_bottleneck_function_matches_op_graph(cfg):
    """FM resblock_bottle as ONE autograd node (blocks.bottleneck: BatchNorm backward sums of bn1 / bn2
    from the dgrad epilogues) against the op-by-op graph: same forward bits and running statistics,
    gradients equal up to the order of the f32 partial sums."""
    import copy
    from torch import nn
    from msml_amd import ops
    from msml_amd.backbones.fm.fmoperator import resblock_bottle
    n, ch, h = cfg
    torch.manual_seed(sum(cfg))
    blk = resblock_bottle(ch, ch)
    for p in blk.parameters():
        if p.dim() == 1:
            nn.init.uniform_(p, 0.5, 1.5)
        else:
            nn.init.normal_(p, 0, (1.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
    for pr in (blk.prelu1, blk.prelu2, blk.prelu3):
        nn.init.uniform_(pr.weight, 0.1, 0.4)
    blk = blk.cuda().train()
    x0 = ops.to_nhwc(torch.randn(n, ch, h, h).cuda(), 1)
    dout = None
    res = []
    for use_fn in (False, True):
        b = copy.deepcopy(blk)
        x = x0.clone().requires_grad_(True)
        old = ops.BLOCK_FUNCTION
        ops.BLOCK_FUNCTION = use_fn
        try:
            y = b(x)
            if dout is None:
                dout = torch.randn_like(y)
            y.backward(dout)
        finally:
            ops.BLOCK_FUNCTION = old
        torch.cuda.synchronize()
        res.append((y.detach().float(), x.grad.float(), {k: v.grad.float() for k, v in b.named_parameters()},
                    {k: v.clone() for k, v in b.named_buffers()}))
    (y1, dx1, g1, buf1), (y2, dx2, g2, buf2) = res
    assert torch.equal(y1, y2)
    for k in buf1:
        assert torch.equal(buf1[k], buf2[k]), k
    assert (dx1 - dx2).abs().max().item() <= 2e-2 * dx1.abs().max().item()
    assert rel_err(dx2.cpu().numpy(), dx1.cpu().numpy()) < 5e-3
    for k in g1:
        assert rel_err(g2[k].cpu().numpy(), g1[k].cpu().numpy()) < 1e-2, k


@pytest.mark.parametrize("stride", [1, 2])
def test_stem_im2col_path_matches_nhwc_conv(stride):
    """bf16 stems: im2col of the raw image + 1x1 conv (functional.RawImage) == the 3x3 conv on the
    NHWC image padded to 32 channels -- outputs, batch statistics and the weight gradient."""
    import copy
    from torch import nn
    from msml_amd.backbones import _nn
    torch.manual_seed(7 + stride)
    conv = nn.Conv2d(3, 64, 3, stride, 1, bias=False)
    bn, pr = nn.BatchNorm2d(64, eps=1e-5), nn.PReLU(64)
    nn.init.normal_(conv.weight, 0, 0.2)
    nn.init.uniform_(pr.weight, 0.1, 0.4)
    mods = [nn.ModuleList([conv, bn, pr]).cuda().train()]
    mods.append(copy.deepcopy(mods[0]))
    x = torch.randn(6, 3, 30, 22, device="cuda")
    outs = []
    for (c, b, a), inp in zip(mods, (Fh.RawImage(x), Fh.to_nhwc(x, 1))):
        y = _nn.conv_bn(c, b, inp, prelu=a)
        if not outs:
            dy = torch.randn_like(y)
        y.backward(dy)
        outs.append((y.detach().float(), c.weight.grad.clone(), b.running_var.clone(), b.weight.grad.clone()))
    (y1, g1, rv1, bg1), (y2, g2, rv2, bg2) = outs
    assert y1.shape == y2.shape
    assert (y1 - y2).abs().max().item() <= 2e-2 * y2.abs().max().item()
    assert rel_err(g1.cpu().numpy(), g2.cpu().numpy()) < 1e-2
    assert torch.allclose(rv1, rv2, rtol=1e-3)
    assert rel_err(bg1.cpu().numpy(), bg2.cpu().numpy()) < 1e-2
    # inference: fused epilogue on the im2col operand
    for (c, b, a) in mods:
        c.eval(); b.eval(); a.eval()
    with torch.no_grad():
        e1 = _nn.conv_bn(mods[0][0], mods[0][1], Fh.RawImage(x), prelu=mods[0][2]).float()
        e2 = _nn.conv_bn(mods[1][0], mods[1][1], Fh.to_nhwc(x, 1), prelu=mods[1][2]).float()
    assert (e1 - e2).abs().max().item() <= 2e-2 * e2.abs().max().item()


def test_block_chain_statistics_hand_off():
    """make_layer marks every block but the last `emit_stats`: bn3's kernel then also emits the
    statistics of the block output, which the next block's bn1 uses instead of a k_bn_stats pass.
    The chain must equal the op-by-op graph (forward bits, running statistics, gradients)."""
    import copy
    from msml_amd import ops
    from msml_amd.backbones.frb.iresnet import IBasicBlock, make_layer
    torch.manual_seed(11)
    layer = make_layer(IBasicBlock, 64, 128, 3, 2)
    for p in layer.parameters():
        if p.dim() == 1:
            torch.nn.init.uniform_(p, 0.5, 1.5)
        else:
            torch.nn.init.normal_(p, 0, (1.0 / (p.shape[1] * 9)) ** 0.5)
    layer = layer.cuda().train()
    assert layer[0].emit_stats and layer[1].emit_stats and not getattr(layer[2], "emit_stats", False)
    x0 = ops.to_nhwc(torch.randn(6, 64, 28, 28).cuda(), 1)
    res, dout = [], None
    for use_fn in (False, True):
        m = copy.deepcopy(layer)
        x = x0.clone().requires_grad_(True)
        old = ops.BLOCK_FUNCTION
        ops.BLOCK_FUNCTION = use_fn
        try:
            y = m(x)
            if dout is None:
                dout = torch.randn_like(y)
            y.backward(dout)
        finally:
            ops.BLOCK_FUNCTION = old
        res.append((y.detach().float(), x.grad.float(), {k: v.clone() for k, v in m.named_buffers()},
                    {k: v.grad.float() for k, v in m.named_parameters()}))
    (y1, dx1, b1, g1), (y2, dx2, b2, g2) = res
    # the handed-over statistics are summed in another order (f32 partial rows): a few outputs move
    # by one bf16 ulp
    assert rel_err(y2.cpu().numpy(), y1.cpu().numpy()) < 3e-3
    assert (y1 - y2).abs().max().item() <= 2e-2 * y1.abs().max().item()
    # (running statistics: 0.1 x batch statistics of tensors in which a few elements differ by one bf16 ulp -> some
    # 1e-5 absolute on values of 1e-2 .. 3e-1)
    for k in b1:
        assert torch.allclose(b1[k].float(), b2[k].float(), rtol=3e-4, atol=3e-5), k
    assert rel_err(dx2.cpu().numpy(), dx1.cpu().numpy()) < 1.5e-2     # three bf16 blocks deep
    for k in g1:
        assert rel_err(g2[k].cpu().numpy(), g1[k].cpu().numpy()) < 2e-2, k


@pytest.mark.parametrize("cfg", [(6, 64, 56), (5, 128, 28), (4, 256, 14), (6, 512, 7)])
def test_fm_stage_gradient_fan_out_summed_in_the_conv_epilogue(cfg):
    """FMCnn (reference fmoperator.py:277-311): the stage input yf feeds same_conv AND the act / arith / skip kernel.
    Fh.conv_tee sums its two gradients in same_conv's backward-data epilogue; against the plain graph, where autograd adds
    them with an element-wise kernel: same forward bit for bit, d yf equal to one bf16 rounding (the fused sum rounds once,
    the separate add twice), every parameter gradient identical."""
    import copy
    from msml_amd import ops
    from msml_amd.backbones.fm.fmoperator import FMCnn
    n, c, h = cfg
    torch.manual_seed(sum(cfg))
    fm = FMCnn(h, h, c, 3, 2, "sigmoid", "mul", {"use_ori": False}).cuda().train()
    yf0 = ops.to_nhwc(torch.randn(n, c, h, h).cuda(), 1)
    yo = ops.to_nhwc(torch.randn(n, 18, h, h).cuda(), 1)
    dz, res = None, []
    for tee in (False, True):
        m = copy.deepcopy(fm)
        yf = yf0.clone().requires_grad_(True)
        old = ops.FM_TEE
        ops.FM_TEE = tee
        try:
            # a producer node in front, as in the network (a leaf input would take the same path)
            z, _ = m(Fh.add(yf, torch.zeros_like(yf)), yo)
            if dz is None:
                dz = torch.randn_like(z)
            z.backward(dz)
        finally:
            ops.FM_TEE = old
        torch.cuda.synchronize()
        res.append((z.detach().clone(), yf.grad.clone(), {k: v.grad.clone() for k, v in m.named_parameters()}))
    (z1, g1, p1), (z2, g2, p2) = res
    assert torch.equal(z1, z2)
    assert rel_err(g2.float().cpu().numpy(), g1.float().cpu().numpy()) < 4e-3
    assert elem_err(g2.float().cpu().numpy(), g1.float().cpu().numpy()) < 1e-2
    for k in p1:
        assert torch.equal(p1[k], p2[k]), k


@pytest.mark.parametrize("cfg", [(3, 64, 56), (5, 64, 28), (4, 128, 14), (6, 512, 4)])
def test_gcm_gradient_fan_out_summed_in_the_conv_epilogue(cfg):
    """Global-Convolution module of the OSB (reference unet.py:16-38): x feeds conv_l1 (7x1) and conv_r1 (1x7).  With
    ops.GCM_TEE conv_r1's input gradient is the residual of conv_l1's backward-data launch (k_conv_line's epilogue at
    56x56 / 28x28, the general kernel's below) -- against the plain graph, where autograd adds the two with an
    element-wise kernel: same forward, every gradient bit for bit (both round the conv result, then the sum)."""
    import copy
    from msml_amd import ops
    from msml_amd.backbones.osb.unet import _GlobalConvModule
    n, c, h = cfg
    torch.manual_seed(sum(cfg))
    gcm = _GlobalConvModule(c, 18, (7, 7)).cuda().train()
    x0 = ops.to_nhwc(torch.randn(n, c, h, h).cuda(), 1)
    dz, res = None, []
    for tee in (False, True):
        m = copy.deepcopy(gcm)
        x = x0.clone().requires_grad_(True)
        old = ops.GCM_TEE
        ops.GCM_TEE = "all" if tee else False           # "all": also the levels the default leaves to autograd's add
        try:
            z = m(Fh.add(x, torch.zeros_like(x)))
            if dz is None:
                dz = torch.randn_like(z)
            z.backward(dz)
        finally:
            ops.GCM_TEE = old
        torch.cuda.synchronize()
        res.append((z.detach().clone(), x.grad.clone(), {k: v.grad.clone() for k, v in m.named_parameters()}))
    (z1, g1, p1), (z2, g2, p2) = res
    assert torch.equal(z1, z2)
    assert g1.float().abs().max().item() > 0
    assert torch.equal(g1, g2)
    for k in p1:
        assert torch.equal(p1[k], p2[k]), k


@pytest.mark.parametrize("cfg", [(6, 64, 64, 56, 1), (5, 64, 128, 28, 2), (4, 256, 256, 14, 1), (6, 256, 512, 14, 2)])
def test_block_forward_in_one_c_call_is_bit_identical(cfg):
    """msml_iblock_fwd (csrc/block.hip: every launch of an IBasicBlock forward behind ONE call across the ABI, VERDICT r3
    item 7) against the same block issued launch by launch from Python (ops.BLOCK_C_ENTRY off): outputs, every
    gradient, running statistics and the statistics handed to the next block bit for bit."""
    import copy
    from torch import nn
    from msml_amd import ops
    from msml_amd.backbones.frb.iresnet import IBasicBlock
    n, cin, cout, h, stride = cfg
    torch.manual_seed(sum(cfg))
    ds = None
    if stride != 1 or cin != cout:
        ds = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout, eps=1e-05))
    blk = IBasicBlock(cin, cout, stride, ds)
    blk2 = IBasicBlock(cout, cout, 1, None)              # a follower, so that the first block emits its output statistics
    for b in (blk, blk2):
        for p in b.parameters():
            if p.dim() == 1:
                nn.init.uniform_(p, 0.5, 1.5)
            else:
                nn.init.normal_(p, 0, (1.0 / (p.shape[1] * 9)) ** 0.5)
    seq = nn.Sequential(blk, blk2).cuda().train()
    seq[0].emit_stats = True
    x0 = ops.to_nhwc(torch.randn(n, cin, h, h).cuda(), 1)
    dout, res = None, []
    for fast in (False, True):
        s = copy.deepcopy(seq)
        s[0].emit_stats = True
        x = x0.clone().requires_grad_(True)
        old = ops.BLOCK_C_ENTRY
        ops.BLOCK_C_ENTRY = fast
        try:
            y = s(Fh.add(x, torch.zeros_like(x)))
            if dout is None:
                dout = torch.randn_like(y)
            y.backward(dout)
        finally:
            ops.BLOCK_C_ENTRY = old
        torch.cuda.synchronize()
        res.append((y.detach().clone(), x.grad.clone(), {k: v.grad.clone() for k, v in s.named_parameters()},
                    {k: v.clone() for k, v in s.named_buffers()}))
    (y1, dx1, g1, b1), (y2, dx2, g2, b2) = res
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    for k in b1:
        assert torch.equal(b1[k], b2[k]), k
