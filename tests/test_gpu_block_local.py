"""LOCAL check of the deep fused bf16 backward.

The global comparison of a bf16 training step with the reference's f32 golden (tests/test_gpu_deep_step.py) carries
the forward rounding of 50-100 layers: its bounds sit at 20-35 % norm-wise and cannot see a missing term in one block.
Here every hand-written backward of the step bench.py times -- one-node IBasicBlocks of the FRB (reference
backbones/frb/iresnet.py:56-67) and of the OSB encoder (backbones/osb/unet.py:80-91), the FM operators'
resblock_bottle (backbones/fm/fmoperator.py:53-68) -- is checked ON ITS OWN: msml_amd.blocks.TAP hands out the block's
actual bf16 input, the BatchNorm coefficients it saved, the gradient it received and the gradient it returned (module
hooks add the block's forward OUTPUT); the
parameter gradients are read from the flat arena (every parameter is used once, the arena was zeroed).  ONE block is
then recomputed in f64 torch (the oracle's block class, filled with the HIP model's parameters; since round 6 on the
device with MIOpen off -- REF_DEV below -- and cross-checked against the CPU run on the small network) from those
same tensors, and the forward output, dX, dW, dgamma, dbeta, dalpha and the saved statistics must agree to what bf16 storage costs THAT
block on THOSE operands: the bound of every quantity is 2 x the error of the same block recomputed on the CPU under
the bf16 rounding model of oracle/bf16_emul.py (three draws) -- a few per cent for an IBasicBlock, 5-7 % for the FM
bottlenecks' weight gradients (three BatchNorm backward projections in a row), instead of the 20-35 % of the global
comparison.  The error no longer compounds, so a missing or mis-scaled term shows: test_local_check_catches_an_injected_fault runs the same check with the identity-path gradient
deliberately left out of the block-input sum (blocks.FAULT = "skip_join") and requires it to turn red.
"""
import os
import time

import numpy as np
import pytest
import torch

from msml_amd import blocks, ops, synthetic
from msml_amd.backbones import MSML
from msml_amd.optim import FlatSGD
from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
from oracle import bf16_emul
from oracle import model as om
from oracle.fill import fill_module
from oracle.inputs import eval_inputs

pytestmark = pytest.mark.gpu
PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
# bound = FLOOR_X x (the block's own emulated bf16 error, worst of three draws) + an absolute term
# (per-channel sums: 3 x -- single channels of 64-512 are noisier than a norm over a whole tensor)
FLOOR_X, CHAN_X, NORM_ABS, FRAC_ABS, CHAN_ABS, ELEM_TOL, STAT_TOL = 2.0, 3.0, 5e-3, 5e-4, 1e-2, 5e-2, 1e-2
# Where the one-block recomputations (f64 truth + emulated bf16 draws, plain torch on the oracle's module classes) run.
# Round 6: on the device by default, with MIOpen switched off (torch's own im2col + rocBLAS dgemm / sgemm kernels and
# native BatchNorm / PReLU kernels -- no half-precision path anywhere): the same arithmetic as on the CPU, 5-10 x
# faster (the two block-by-block tests took 126 s of the suite's 418 s on 16 CPU threads).  "cpu" restores the CPU
# run; test_local_check_catches_an_injected_fault runs the small network BOTH ways and requires the rows to agree.
REF_DEV = os.environ.get("MSML_LOCAL_REF_DEV", "cuda")


def _np(t):
    return t.detach().cpu().numpy()


def _step_with_taps(frb, bs, fault=""):
    """One training step on the path bench.py times (bf16, one-node blocks, BatchNorm sums from the backward-data
    epilogues, grouped weight gradients + OSB on side streams, in-place gradients into FlatSGD's arena) with the
    block taps on.  Returns (model, taps, gradients by parameter name)."""
    torch.manual_seed(0)
    m = fill_module(MSML(frb, "unet", (1, 1, 1, 1), 1000, fp16=True, fm_params=(3, 2, "sigmoid", "mul"),
                         header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0),
                         peer_params=dict(PEER_OFF))).cuda().train()
    x, msk = eval_inputs(bs)
    label = synthetic.labels(bs, 1000, seed=1)
    opt = FlatSGD([{"params": [p for p in m.parameters() if p.requires_grad], "lr": 0.1 / 512 * bs}], 0.9, 5e-4, 5.0)
    taps = []
    outs = {}                     # id(conv1.weight) -> the block's forward output (module hooks)

    def tap(kind, bp, t):
        taps.append((kind, id(bp["c1"][0]), {k: (v.detach().clone() if v is not None else None) for k, v in t.items()}))
    hooks = [mod.register_forward_hook(lambda md, inp, out: outs.__setitem__(id(md.conv1.weight), out.detach().clone()))
             for name, mod in m.named_modules() if name and hasattr(mod, "conv1") and hasattr(mod, "bn3")]
    assert ops.BLOCK_FUNCTION and ops.FUSE_BN_BWD and ops.BOTTLE_FUNCTION
    ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
    blocks.TAP, blocks.FAULT = tap, fault
    try:
        opt.zero_grad()
        final_cls, final_seg, _ = m(x.cuda(), label.cuda(), None)
        loss = torch.nn.functional.cross_entropy(final_cls, label.cuda()) + \
            StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")(final_seg, msk.cuda(), msk.cuda())
        loss.backward()
        ops.wgrad_stream_join()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        blocks.TAP, blocks.FAULT = None, ""
        ops.WGRAD_STREAM = ops.OSB_STREAM = None
        opt.release()
        for h in hooks:
            h.remove()
    for _, wid, t in taps:
        t["out"] = outs.get(wid)
    return m, taps, grads


def _nchw64(t, c, dev="cpu"):
    """NHWC storage tensor (bf16, padded channels) -> NCHW f64 on `dev`, first c channels."""
    return t[..., :c].permute(0, 3, 1, 2).float().to(dev).double().contiguous()


def _frac_beyond(got, want, tol):
    """Fraction of elements whose error exceeds tol x the largest reference element.  (A maximum over millions of
    elements is not a usable statistic behind a PReLU: where the pre-activation sits within a bf16 rounding of zero the
    two sides take different branches and that element's gradient differs by (1 - alpha) x its incoming gradient.)
    f64 torch tensors on one device (round 6: the comparisons run where the recomputation ran -- numpy on the host spent
    more time on the 25-million-element maps than the recomputation itself)."""
    if not isinstance(got, torch.Tensor):            # (tests/test_gpu_module_local.py passes numpy arrays)
        got, want = torch.as_tensor(np.asarray(got, np.float64)), torch.as_tensor(np.asarray(want, np.float64))
    return float(((got - want).abs() > tol * want.abs().max()).double().mean())


def _rel(got, want):
    """Norm-wise relative error ||got - want|| / ||want|| of two f64 tensors (tests.helpers.rel_err on the device)."""
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


def _ref_block(kind, mod, dtype, dev="cpu"):
    if kind == "iblock":
        cout, cin = mod.conv1.weight.shape[:2]
        ref = om.IBasicBlock(cin, cout, mod.conv2.stride[0], mod.downsample is not None)
    else:
        cin = cout = mod.conv1.weight.shape[1]
        ref = om.ResBottle(cin)
    ref = ref.to(dtype).train()
    ref.load_state_dict({k: v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()
                         for k, v in mod.state_dict().items()}, strict=True)
    return ref.to(dev), cin, cout


def _errors(got, want, terms):
    """got / want: {'dx': array, parameter name: array}.  Tensor gradients (dX, conv weights): (norm-wise error,
    fraction of elements beyond ELEM_TOL of the largest); per-channel gradients (dgamma, dbeta, dalpha): the worst
    channel's error in units of that channel's term norm."""
    tens, chan = {}, {}
    for k, w in want.items():
        if k in terms:
            chan[k] = float(((got[k] - w).abs() / terms[k].clamp_min(1e-300)).max())
        else:
            tens[k] = (_rel(got[k], w), _frac_beyond(got[k], w, ELEM_TOL))
    return tens, chan


def _check_block(kind, name, mod, t, grads, dev=None):
    with torch.backends.cudnn.flags(enabled=False):       # (MIOpen off: torch's native f64 / f32 conv kernels on the device)
        return _check_block_on(kind, name, mod, t, grads, REF_DEV if dev is None else dev)


def _check_block_on(kind, name, mod, t, grads, dev):
    """Recompute ONE block from the tensors its HIP backward consumed: in f64 (the truth) and, for the bound, in f32
    under the bf16 rounding model of oracle/bf16_emul.py (every stored activation / activation gradient rounded, bf16
    conv operands -- plain PyTorch hooks, no HIP code) for three draws of the rounding noise.  The emulation's own
    error against f64 is what bf16 storage costs THIS block on THESE operands; the HIP block must stay within 2 x the
    worst draw (+ a small absolute term).  Returns rows (what, hip error, bound) for tensors (norm-wise), element
    fractions, per-channel gradients, and the saved statistics.
    A per-channel parameter gradient is a sum over N*H*W terms that largely cancel (the shift of a BatchNorm in front
    of conv -> BatchNorm has NO effect beyond the border pixels: its exact gradient is ~0), so its error is measured
    against the root-sum-square of its TERMS -- the scale on which roundings of the terms add up -- not against the
    sum itself."""
    prefix = name + "."
    ref, cin, cout = _ref_block(kind, mod, torch.float64, dev)
    x64 = _nchw64(t["x"], cin, dev).requires_grad_()
    dout64 = _nchw64(t["dout"], cout, dev)
    feats, terms, hooks = {}, {}, []

    def watch(mname, m):
        def fwd(_m, inp, out):
            z = inp[0].detach()
            feats[mname] = z

            def bwd(g):
                if isinstance(m, torch.nn.PReLU):
                    terms[mname + ".weight"] = (g * z.clamp_max(0)).pow(2).sum((0, 2, 3)).sqrt().double()
                else:
                    mu = z.mean((0, 2, 3), keepdim=True)
                    xh = (z - mu) / (z.var((0, 2, 3), unbiased=False, keepdim=True) + m.eps).sqrt()
                    terms[mname + ".bias"] = g.pow(2).sum((0, 2, 3)).sqrt().double()
                    terms[mname + ".weight"] = (g * xh).pow(2).sum((0, 2, 3)).sqrt().double()
            out.register_hook(bwd)
        hooks.append(m.register_forward_hook(fwd))
    for mname, m in ref.named_modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.PReLU)):
            watch(mname, m)
    y64 = ref(x64)
    y64.backward(dout64)
    for h in hooks:
        h.remove()
    want = {"dx": x64.grad.detach()}
    if t.get("out") is not None:           # the block's FORWARD output from the same input (train-mode statistics)
        want["out"] = y64.detach()
    want.update({pn: p.grad.detach() for pn, p in ref.named_parameters()})
    got = {"dx": _nchw64(t["dx"], cin, dev)}
    if "out" in want:
        got["out"] = _nchw64(t["out"], cout, dev)
    got.update({pn: grads[prefix + pn].to(dev).double() for pn in want if pn not in ("dx", "out")})
    h_tens, h_chan = _errors(got, want, terms)
    # the local bf16 floor: the same block, f32, under the rounding model, three draws
    f_tens, f_chan = {}, {}
    for shift in (0.0, 0.31, -0.27):
        emu, _, _ = _ref_block(kind, mod, torch.float32, dev)
        bf16_emul.emulate(emu)
        bf16_emul.GRID_SHIFT = shift
        try:
            xe = x64.detach().float().requires_grad_()
            ye = emu(xe)
            ye.backward(dout64.float())
        finally:
            bf16_emul.GRID_SHIFT = 0.0
        ge = {"dx": bf16_emul._r(xe.grad).double()}
        if "out" in want:
            ge["out"] = bf16_emul._r(ye.detach()).double()
        ge.update({pn: p.grad.double() for pn, p in emu.named_parameters()})
        a, b = _errors(ge, want, terms)
        for k, (e, fr) in a.items():
            f_tens[k] = (max(f_tens.get(k, (0, 0))[0], e), max(f_tens.get(k, (0, 0))[1], fr))
        for k, e in b.items():
            f_chan[k] = max(f_chan.get(k, 0.0), e)
    tens = [(k, e, FLOOR_X * f_tens[k][0] + NORM_ABS) for k, (e, _) in h_tens.items()]
    frac = [(k, fr, FLOOR_X * f_tens[k][1] + FRAC_ABS) for k, (_, fr) in h_tens.items()]
    chan = [(k, e, CHAN_X * f_chan[k] + CHAN_ABS) for k, e in h_chan.items()]
    # saved BatchNorm coefficients [4][C] = scale, shift, mean, invstd against the f64 batch statistics of the tensor
    # the oracle block feeds to the same BatchNorm (mean error in units of the standard deviation, invstd relative)
    stats = []
    for b, key in (("bn1", "k1"), ("bn2", "k2"), ("bn3", "k3")):
        k = t[key].float().cpu().double().numpy()
        z = feats[b]
        c = z.shape[1]
        mean = _np(z.mean((0, 2, 3)))
        var = _np(z.var((0, 2, 3), unbiased=False))
        eps = getattr(ref, b).eps
        stats.append((b + ".mean", float(np.abs(k[2][:c] - mean).max() / np.sqrt(var + eps).max()), STAT_TOL))
        stats.append((b + ".invstd", float(np.abs(k[3][:c] * np.sqrt(var + eps) - 1.0).max()), STAT_TOL))
    return tens, frac, chan, stats


def _run_local_check(frb, bs, fault="", dev=None):
    """Returns (model, rows, n IBasicBlocks, n bottlenecks); rows = {family: [(name, hip error, bound)]}."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    m, taps, grads = _step_with_taps(frb, bs, fault)
    by_w = {id(mod.conv1.weight): (name, mod) for name, mod in m.named_modules()
            if hasattr(mod, "conv1") and hasattr(mod, "bn3") and name}
    rows = {"norm-wise": [], "element fraction": [], "per-channel": [], "statistic": []}
    t0 = time.time()
    for kind, wid, t in taps:
        name, mod = by_w[wid]
        for fam, part in zip(rows, _check_block(kind, name, mod, t, grads, dev)):
            rows[fam] += [(name + "." + what, e, b) for what, e, b in part]
    n_i = sum(1 for k, _, _ in taps if k == "iblock")
    n_b = len(taps) - n_i
    print("local f64 check %s b%d%s: %d IBasicBlocks (FRB + OSB) + %d FM bottlenecks; %d tensor gradients, %d per-channel "
          "parameter gradients; %.0f s on %s (f64 truth + 3 emulated bf16 draws per block)"
          % (frb, bs, " FAULT=" + fault if fault else "", n_i, n_b, len(rows["norm-wise"]), len(rows["per-channel"]),
             time.time() - t0, dev or REF_DEV))
    for fam, rr in rows.items():
        top = sorted(rr, key=lambda r: -r[1] / r[2])[:4]
        print("   %s, closest to their bounds: " % fam + "; ".join("%s %.2e (bound %.2e)" % r for r in top))
        print("      largest error %.3e, largest bound %.3e" % (max(r[1] for r in rr), max(r[2] for r in rr)))
        fo = [r for r in rr if r[0].endswith(".out")]
        if fo and fam == "norm-wise":
            print("      forward outputs (%d blocks): largest error %.3e, its bound %.3e"
                  % ((len(fo),) + max(fo, key=lambda r: r[1])[1:]))
    return m, rows, n_i, n_b


def _bad(rows):
    return [(fam,) + r for fam, rr in rows.items() for r in rr if not r[1] < r[2]]


@pytest.mark.parametrize("frb,bs,n_iblocks", [("iresnet50", 32, 24 + 8), ("iresnet100", 16, 49 + 8)])
def test_deep_bf16_backward_block_by_block_f64(frb, bs, n_iblocks):
    """Every IBasicBlock (FRB stages + OSB encoder) and every FM bottleneck of the deep networks' bf16 fused training
    step against ONE-block f64 recomputations from the block's own operands."""
    m, rows, n_i, n_b = _run_local_check(frb, bs)
    assert n_i == n_iblocks and n_b == 8, (n_i, n_b)
    assert sum(1 for r in rows["norm-wise"] if r[0].endswith(".out")) == n_i + n_b       # every block's forward output too
    assert not _bad(rows), _bad(rows)[:10]
    # the bounds themselves stay small: a local bound has power (the global ones sit at 0.2-0.35)
    assert max(b for _, _, b in rows["norm-wise"]) < 0.16


def test_local_check_catches_an_injected_fault():
    """Power of the local check: the same step with the identity-path gradient left out of every stride-1 block's
    input sum (blocks.FAULT = 'skip_join', a one-term fault of the hand-written backward) must put exactly those
    blocks' dX far outside their bounds -- while every parameter gradient OF THOSE BLOCKS, computed from the (intact)
    incoming gradient, and every other block's dX stay inside."""
    m, rows, n_i, n_b = _run_local_check("iresnet18", 8, fault="skip_join")
    dx = {n[:-3]: (e, b) for n, e, b in rows["norm-wise"] if n.endswith(".dx")}
    faulty = [n for n, mod in m.named_modules() if n in dx and hasattr(mod, "downsample") and mod.downsample is None]
    assert len(faulty) == 4 + 4 and len(dx) == n_i + n_b                 # FRB + OSB: the second block of each stage
    assert all(dx[n][0] > 5 * dx[n][1] and dx[n][0] > 0.3 for n in faulty), {n: dx[n] for n in faulty}
    bad = _bad(rows)
    assert {r[1] for r in bad if r[0] == "norm-wise"} == {n + ".dx" for n in faulty}, bad[:12]
    assert all(r[1].endswith(".dx") and r[1][:-3] in faulty for r in bad), bad[:12]
    # and the unfaulted small network passes in full -- with the one-block recomputations on the device (the default,
    # REF_DEV) AND on the CPU: the f64 truth is the same arithmetic on either, so the HIP path's errors must read the
    # same (the bounds come from f32 emulations whose bf16 roundings may fall differently: same size, not same digits)
    _, rows, _, _ = _run_local_check("iresnet18", 8)
    assert not _bad(rows), _bad(rows)[:10]
    if REF_DEV != "cpu":
        _, rows_cpu, _, _ = _run_local_check("iresnet18", 8, dev="cpu")
        assert not _bad(rows_cpu), _bad(rows_cpu)[:10]
        for fam in rows:
            assert [r[0] for r in rows[fam]] == [r[0] for r in rows_cpu[fam]]
            for (n, e, b), (_, ec, bc) in zip(rows[fam], rows_cpu[fam]):
                assert abs(e - ec) <= 1e-5 + 1e-4 * ec, (fam, n, e, ec)
                assert abs(b - bc) <= 0.25 * bc + 1e-3, (fam, n, b, bc)
