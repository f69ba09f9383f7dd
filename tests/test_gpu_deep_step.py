"""The training step of the DEEP networks (config 3: ires50, config 4: the ires100-variant) under pytest.

* exact-f32 op graph and the bf16 fused path (one-node blocks, side streams, FlatSGD) against the reference's
  own training-step goldens G4c (oracle/make_golden.py g4c: train.py:252-277 on iresnet.py:470-481 / the
  [3,13,30,3] variant), with one early / middle / last block of every stage among the picked gradients;
* bf16 tolerances are DERIVED: 2 x the error floor that the CPU oracle shows under the bf16 rounding model of
  oracle/bf16_emul.py (tests/golden/bf16_floor.npz, recorded by oracle/make_bf16_floor.py), per parameter group;
* one FULL-SIZE step of the headline workload (ires50 + 85 742-id PartialFC, batch 256): bf16 fused path against
  the exact-f32 HIP path on the same inputs, and hipGraph replay == eager issue.
"""
import argparse
import os

import numpy as np
import pytest
import torch

from msml_amd import ops, synthetic
from msml_amd.backbones import MSML
from msml_amd.optim import FlatSGD
from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
from oracle.bf16_emul import param_group
from oracle.fill import fill_module
from oracle.inputs import eval_inputs
from tests.helpers import assert_cs, bf16_tolerances, cosine, elem_err, load, pick, rel_err

pytestmark = pytest.mark.gpu
PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
COS_MIN = 0.94       # cosine similarity of a picked bf16 gradient with the reference's (see test_deep_train_step_bf16_fused)


def hip_msml(frb, C=1000, fp16=False):
    torch.manual_seed(0)
    m = MSML(frb, "unet", (1, 1, 1, 1), C, fp16=fp16, fm_params=(3, 2, "sigmoid", "mul"),
             header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0), peer_params=dict(PEER_OFF))
    return fill_module(m).cuda()


@pytest.mark.parametrize("frb,bs", [("iresnet50", 8), ("iresnet100", 4), ("iresnet50", 256)])
def test_deep_train_step_f32(frb, bs):
    """Exact-f32 path: losses, grad norm, 26 picked gradients (norm-wise AND element-wise), running statistics.
    ires50 at batch 256 is the headline workload's backbone step at its FULL per-GPU batch against the reference's own
    golden (oracle/make_golden.py g4d, 256-element picks)."""
    g = load("g4_train_%s_b%d.npz" % (frb.replace("iresnet", "ires"), bs))
    m = hip_msml(frb, 1000)
    x, msk = eval_inputs(bs)
    label = synthetic.labels(bs, 1000, seed=1)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.1 / 512 * bs, momentum=0.9, weight_decay=5e-4)
    final_cls, final_seg, kd = m(x.cuda(), label.cuda(), None)
    seg_loss = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")(final_seg, msk.cuda(), msk.cuda())
    cls_loss = torch.nn.functional.cross_entropy(final_cls, label.cuda())
    (cls_loss + seg_loss).backward()
    gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), 5, 2)
    assert abs(seg_loss.item() - g["seg_loss"]) < 1e-3 * abs(g["seg_loss"])
    assert abs(cls_loss.item() - g["cls_loss"]) < 1e-3 * abs(g["cls_loss"])
    assert abs(float(gnorm) - g["grad_norm"]) < 5e-3 * abs(g["grad_norm"])
    assert_cs(final_cls, g["final_cls_cs"], 1e-3, "final_cls")
    params = dict(m.named_parameters())
    worst, worst_el, n = 0.0, 0.0, 0
    for key in g.files:
        if key.startswith("grad_pick/"):
            name = key.split("/", 1)[1]
            got = pick(params[name].grad, g[key].size)
            if name == "frb.fc.bias":           # exact gradient 0 (train-mode BatchNorm1d follows): noise on both sides
                assert np.abs(got).max() < 1e-5 and np.abs(g[key]).max() < 1e-5
                continue
            e, ee = rel_err(got, g[key]), elem_err(got, g[key])
            worst, worst_el, n = max(worst, e), max(worst_el, ee), n + 1
            assert e < 1e-2 and ee < 1e-2, (name, e, ee)
    print("f32 deep train step %s b%d: %d picked gradients, worst norm-wise %.3e, worst element-wise %.3e"
          % (frb, bs, n, worst, worst_el))
    assert n >= 25
    opt.step()
    for key in g.files:
        if key.startswith("stat/"):
            name = key.split("/", 1)[1]
            assert rel_err(m.state_dict()[name].cpu().numpy(), g[key]) < 1e-3, name


@pytest.mark.parametrize("frb,bs", [("iresnet50", 8), ("iresnet50", 32), ("iresnet100", 16), ("iresnet50", 256)])
def test_deep_train_step_bf16_fused(frb, bs):
    """The path bench.py times (bf16, one-node blocks, BatchNorm backward sums from the backward-data epilogues,
    side streams, FlatSGD) on the deep FRBs against the reference golden: every picked gradient norm-wise within
    min(3 x median emulated bf16 floor of its group, 0.35), cosine similarity >= 0.94 (what a norm-wise error of 0.35
    orthogonal to the gradient leaves: 1 / sqrt(1 + 0.35^2) = 0.944), element-wise within 2 x the norm-wise bound.  (The ires100 batch-4 golden stays an exact-f32 case only: BatchNorm1d over four samples in front
    of an s = 64 head makes single bf16 rounding draws differ by 2-3 x, no bf16 bound on it means anything.  The
    block-by-block f64 check of the same step is tests/test_gpu_block_local.py.)"""
    short = frb.replace("iresnet", "ires")
    g = load("g4_train_%s_b%d.npz" % (short, bs))
    # (batch 256 = the full per-GPU batch of the headline workload, golden g4d: bounds of the batch-32 floor, the
    # largest batch the emulation was recorded at -- rounding noise averages down with the batch, never up)
    tol = bf16_tolerances("%s_b%d" % (short, min(bs, 32)))
    assert ops.BLOCK_FUNCTION and ops.FUSE_BN_BWD and ops.BOTTLE_FUNCTION
    m = hip_msml(frb, 1000, fp16=True)
    x, msk = eval_inputs(bs)
    label = synthetic.labels(bs, 1000, seed=1)
    m.train()
    opt = FlatSGD([{"params": [p for p in m.parameters() if p.requires_grad], "lr": 0.1 / 512 * bs}], 0.9, 5e-4, 5.0)
    ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
    hits0 = ops.COUNTERS["bn3_partial_hits"]
    try:
        opt.zero_grad()
        final_cls, final_seg, kd = m(x.cuda(), label.cuda(), None)
        seg_loss = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")(final_seg, msk.cuda(), msk.cuda())
        cls_loss = torch.nn.functional.cross_entropy(final_cls, label.cuda())
        (cls_loss + seg_loss).backward()
        ops.wgrad_stream_join()
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        opt.step()
        torch.cuda.synchronize()
    finally:
        ops.WGRAD_STREAM = ops.OSB_STREAM = None
        opt.release()
    # the chained blocks handed their bn3 sums over: every non-first block of the FRB stages and of the OSB
    assert ops.COUNTERS["bn3_partial_hits"] - hits0 >= (16 if frb == "iresnet50" else 40)
    assert abs(seg_loss.item() - g["seg_loss"]) < tol["loss"] * abs(g["seg_loss"])
    assert abs(cls_loss.item() - g["cls_loss"]) < tol["loss"] * abs(g["cls_loss"])
    gnorm = float(opt.grad_norm())
    clip = float(5.0 / (g["grad_norm"] + 1e-6))
    report, bad = [], []
    for key in g.files:
        if key.startswith("grad_pick/"):
            n = key.split("/", 1)[1]
            if n == "frb.fc.bias":
                continue
            got = pick(grads[n], g[key].size) * clip
            e, ee, cs = rel_err(got, g[key]), elem_err(got, g[key]), cosine(got, g[key])
            t = tol[param_group(n)]
            report.append((e / t, e, t, n, ee, cs))
            if e >= t or cs < COS_MIN or ee >= 2 * t:
                bad.append((n, e, t, ee, cs))
    report.sort(reverse=True)
    print("bf16 fused deep step %s b%d: gnorm %.4f vs %.4f (%.2e, tol %.1e); seg %.5f / %.5f cls %.5f / %.5f (tol %.1e)"
          % (frb, bs, gnorm, g["grad_norm"], abs(gnorm / g["grad_norm"] - 1), tol["gnorm"], seg_loss.item(),
             g["seg_loss"], cls_loss.item(), g["cls_loss"], tol["loss"]))
    for r, e, t, n, ee, cs in report:
        print("   %-46s rel err %.3e = %.2f x tol %.3f (%s)  element-wise %.3e  cosine %.4f"
              % (n, e, r, t, param_group(n), ee, cs))
    assert abs(gnorm - g["grad_norm"]) < tol["gnorm"] * abs(g["grad_norm"]), (gnorm, g["grad_norm"])
    assert not bad, bad
    sd = m.state_dict()
    for key in g.files:
        if key.startswith("stat/"):
            n = key.split("/", 1)[1]
            assert rel_err(sd[n].cpu().numpy(), g[key]) < tol["stat"], (n, rel_err(sd[n].cpu().numpy(), g[key]))


# ---- sampled ONE-block f64 checks inside the full-size steps ---------------------------------------
# The two full-size tests below compare the bf16 HIP path with the f32 HIP path -- a self-comparison.  A few blocks of
# the SAME bf16 step (one per FRB stage, one FM bottleneck, one OSB block) are tapped at the FULL per-GPU batch and
# recomputed on their own in f64 on the CPU from the tensors the HIP backward consumed, with the bounds of
# tests/test_gpu_block_local.py (2 x the block's own emulated bf16 error, three draws): a reference-side assertion at the
# batch the headline runs, where the block-by-block test stops at batch 32 / 16.
class _Snap:
    """A tapped block with the parameters it had BEFORE the optimizer step of the tapped training step."""

    def __init__(self, mod):
        self.conv1, self.conv2, self.downsample = mod.conv1, getattr(mod, "conv2", None), getattr(mod, "downsample", None)
        self._sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}

    def state_dict(self):
        return self._sd


def _tapped_step(model, names, step_fn):
    """Run step_fn() with block taps on `names`; returns (step_fn's result, rows of the local f64 check)."""
    from msml_amd import blocks
    from tests.test_gpu_block_local import _check_block
    mods = dict(model.named_modules())
    snaps = {n: _Snap(mods[n]) for n in names}
    by_w = {id(mods[n].conv1.weight): n for n in names}
    taps = {}

    def tap(kind, bp, t):
        n = by_w.get(id(bp["c1"][0]))
        if n is not None:
            taps[n] = (kind, {k: (v.detach().clone() if v is not None else None) for k, v in t.items()})
    blocks.TAP = tap
    try:
        out = step_fn()
        torch.cuda.synchronize()
    finally:
        blocks.TAP = None
    assert set(taps) == set(names), (sorted(taps), names)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rows = {"norm-wise": [], "element fraction": [], "per-channel": [], "statistic": []}
    for n in names:
        kind, t = taps[n]
        grads = {n + "." + pn: p.grad.detach().float().cpu() for pn, p in mods[n].named_parameters()}
        for fam, part in zip(rows, _check_block(kind, n, snaps[n], t, grads)):
            rows[fam] += [(n + "." + what, e, b) for what, e, b in part]
    return out, rows


def _report_rows(tag, rows):
    bad = [(fam,) + r for fam, rr in rows.items() for r in rr if not r[1] < r[2]]
    for fam, rr in rows.items():
        top = sorted(rr, key=lambda r: -r[1] / r[2])[:3]
        print("   %s one-block f64 check, %s, closest to their bounds: " % (tag, fam) +
              "; ".join("%s %.2e (bound %.2e)" % r for r in top))
    return bad


def _bench_args(dtype, frb="iresnet50", classes=85742, emulate_world=1):
    return argparse.Namespace(frb=frb, batch=256, classes=classes, dtype=dtype, emulate_world=emulate_world,
                              data="resident")


STAGE_PICKS = ["frb.conv1.weight", "frb.layer1.2.conv2.weight", "frb.layer2.3.conv1.weight",
               "frb.layer3.6.conv1.weight", "frb.layer3.13.conv2.weight", "frb.layer4.2.conv1.weight",
               "frb.fm_ops.0.same_conv.weight", "frb.fm_ops.2.res_block.0.conv2.weight", "frb.fc.weight",
               "osb.conv1.weight", "osb.layer4.1.conv2.weight", "osb.gcm1.conv_l1.weight", "osb.deconv5.weight"]


def test_full_size_step_bf16_vs_f32_and_graph_replay():
    """Config 3 at its full single-GPU size -- ires50-MSML + 85 742-id PartialFC, batch 256, the step of bench.py
    (train.py:282-318): (1) the bf16 fused path against the exact-f32 HIP path on the same weights and batch
    (head loss, seg loss, clipped-gradient norm, one picked gradient per stage, the head's dW), bounded by 2 x the
    emulated bf16 floor of the batch-32 ires50 golden; (2) three hipGraph replays of the bf16 step equal three
    eager steps."""
    import bench
    # (a SELF-comparison, bf16 HIP against f32 HIP: what it adds is the 85 742-id PartialFC at full size and graph
    # replay == eager issue; parity of the full-batch step with the REFERENCE is test_deep_train_step_f32 / _bf16_fused
    # [iresnet50-256] (golden g4d).  Both sides of the difference carry rounding here and the picks include the FM
    # bottleneck 3x3 weights, the worst-conditioned gradients of the network -- their ONE-block bf16 error is already
    # 6 %, tests/test_gpu_block_local.py -- measured 0.38: cap 0.45)
    tol = bf16_tolerances("ires50_b32", cap=0.45)

    local_rows = {}

    def run(dtype, steps=1, graph=False, streams=True, local=None):
        torch.manual_seed(0)
        tr = bench.Trainer(_bench_args(dtype), 0, 0, 1)
        batch = tr.batches[0]
        if streams and dtype == "bf16":
            ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
        try:
            losses = []
            if graph:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        tr.step(batch)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    out = tr.step(batch)
                for _ in range(steps):
                    gr.replay()
                    losses.append((float(out[0]), float(out[1])))
            elif local:
                (lv, sl), rows = _tapped_step(tr.model, local, lambda: tr.step(batch))
                losses.append((float(lv), float(sl)))
                local_rows.update(rows)
            else:
                for _ in range(steps + (2 if steps > 1 else 0)):
                    lv, sl = tr.step(batch)
                    losses.append((float(lv), float(sl)))
                losses = losses[-steps:]
            torch.cuda.synchronize()
            names = dict(tr.model.named_parameters())
            picks = {n: pick(names[n].grad, 256) for n in STAGE_PICKS}
            picks["pfc.sub_weight"] = pick(tr.pfc.sub_weight.grad, 1024)
            res = {"losses": losses, "gnorm": float(tr.opt.grad_norm()), "picks": picks,
                   "w": tr.opt.flat_w.clone(), "hw": pick(tr.opt_pfc.flat_w, 4096)}
        finally:
            ops.WGRAD_STREAM = ops.OSB_STREAM = None
            tr.opt.release()
            tr.opt_pfc.release()
        del tr
        torch.cuda.empty_cache()
        return res

    f32 = run("f32")
    # the bf16 step with six blocks tapped at the full batch (one per FRB stage, an FM bottleneck, an OSB block)
    b16 = run("bf16", local=["frb.layer1.2", "frb.layer2.3", "frb.layer3.6", "frb.layer4.2", "frb.fm_ops.2.res_block.0",
                             "osb.layer2.1"])
    bad = _report_rows("batch 256", local_rows)
    assert not bad, bad[:10]
    (lv32, sl32), (lv16, sl16) = f32["losses"][0], b16["losses"][0]
    print("full-size step: head loss f32 %.5f bf16 %.5f | seg loss %.5f / %.5f | gnorm %.4f / %.4f"
          % (lv32, lv16, sl32, sl16, f32["gnorm"], b16["gnorm"]))
    assert abs(lv16 - lv32) < tol["loss"] * abs(lv32)
    assert abs(sl16 - sl32) < tol["loss"] * abs(sl32)
    assert abs(b16["gnorm"] - f32["gnorm"]) < max(tol["gnorm"], 2e-2) * f32["gnorm"]
    for n, ref in f32["picks"].items():
        e = rel_err(b16["picks"][n], ref)
        t = tol["head"] if n == "pfc.sub_weight" else tol[param_group(n)]
        print("   %-46s bf16 vs f32 rel err %.3e (tol %.3f)" % (n, e, t))
        assert e < t, (n, e, t)
    # (2) graph replay == eager issue, three steps from identical state
    eg = run("bf16", steps=3)
    gr = run("bf16", steps=3, graph=True)
    print("eager losses", eg["losses"], "graph losses", gr["losses"])
    # both runs took 2 warm-up steps + 3 steps on the same batch: same trajectory
    for (a, b), (c, d) in zip(eg["losses"], gr["losses"]):
        assert abs(a - c) <= 1e-4 * abs(a) and abs(b - d) <= 1e-4 * abs(b), (eg["losses"], gr["losses"])
    assert float((eg["w"] - gr["w"]).abs().max()) <= 1e-5 * float(eg["w"].abs().max())
    assert rel_err(gr["hw"], eg["hw"]) < 1e-5


def test_config4_full_size_step_ires100_2m_ids_shard():
    """Config 4 at its full per-GPU size: the ires100-variant [3, 13, 30, 3] + rank 0 of an 8-way
    class-parallel 2 000 000-id PartialFC (250 000 local rows x 2048 gathered feature rows, partial_fc.py:34-36 sizing),
    batch 256 -- the step `bench.py --frb iresnet100 --classes 2000000 --emulate-world 8` times.  bf16 fused path
    against the exact-f32 HIP path on the same weights and batch (a self-comparison like the config-3 test above: the
    reference parity of this network is the b4 / b16 goldens and the block-by-block check), plus size-independent
    properties: every gradient finite, the clipped-gradient norm of both paths agrees, the head shard has the survey's
    shape, the whole step stays far inside 288 GB."""
    import bench
    tol = bf16_tolerances("ires100_b16", cap=0.45)
    picks = [n for n in STAGE_PICKS if n != "frb.layer3.13.conv2.weight"] + ["frb.layer3.29.conv2.weight",
                                                                              "frb.layer2.12.conv1.weight"]

    LOCAL4 = ["frb.layer1.2", "frb.layer2.12", "frb.layer3.29", "frb.layer4.2", "frb.fm_ops.1.res_block.1", "osb.layer3.1"]
    local_rows = {}

    def run(dtype):
        torch.manual_seed(0)
        torch.cuda.reset_peak_memory_stats()
        tr = bench.Trainer(_bench_args(dtype, "iresnet100", 2000000, 8), 0, 0, 1)
        assert tr.pfc.num_local == 250000 and tuple(tr.pfc.sub_weight.shape) == (250000, 512)
        if dtype == "bf16":
            ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
        try:
            if dtype == "bf16":
                (lv, sl), rows = _tapped_step(tr.model, LOCAL4, lambda: tr.step(tr.batches[0]))
                local_rows.update(rows)
            else:
                lv, sl = tr.step(tr.batches[0])
            torch.cuda.synchronize()
            names = dict(tr.model.named_parameters())
            finite = all(bool(torch.isfinite(p.grad).all()) for p in names.values() if p.grad is not None)
            res = {"loss": (float(lv), float(sl)), "gnorm": float(tr.opt.grad_norm()), "finite": finite,
                   "picks": {n: pick(names[n].grad, 256) for n in picks},
                   "head": pick(tr.pfc.sub_weight.grad, 4096), "peak_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
        finally:
            ops.WGRAD_STREAM = ops.OSB_STREAM = None
            tr.opt.release()
            tr.opt_pfc.release()
        del tr
        torch.cuda.empty_cache()
        return res

    f32, b16 = run("f32"), run("bf16")
    bad = _report_rows("config 4, batch 256", local_rows)
    assert not bad, bad[:10]
    print("config 4 full size: head loss f32 %.5f bf16 %.5f | seg %.5f / %.5f | gnorm %.4f / %.4f | peak %.1f / %.1f GB"
          % (f32["loss"][0], b16["loss"][0], f32["loss"][1], b16["loss"][1], f32["gnorm"], b16["gnorm"],
             f32["peak_gb"], b16["peak_gb"]))
    assert f32["finite"] and b16["finite"]
    assert np.isfinite(f32["gnorm"]) and f32["gnorm"] > 0
    assert abs(b16["loss"][0] - f32["loss"][0]) < tol["loss"] * abs(f32["loss"][0])
    assert abs(b16["loss"][1] - f32["loss"][1]) < tol["loss"] * abs(f32["loss"][1])
    assert abs(b16["gnorm"] - f32["gnorm"]) < max(tol["gnorm"], 2e-2) * f32["gnorm"]
    # (picked gradients: with 2 M classes behind s = 64 the per-sample gradient is tiny -- clipped norm 16 against 40 at
    # 85 742 ids -- and BOTH sides of this difference carry rounding: what is asserted is direction and scale, cosine
    # >= 0.80 and norm-wise <= 0.65; the arithmetic is pinned block by block in tests/test_gpu_block_local.py.  The worst
    # tensor, an FM bottleneck weight, sits at 0.52-0.56 / 0.84-0.86: a change that only reorders f64 additions of the
    # BatchNorm sums -- the pointwise conv kernel on or off -- redraws the rounding noise of everything behind it and moves
    # that cosine by 0.02, which is why the bound is not drawn at the first measured value)
    for n, ref in list(f32["picks"].items()) + [("pfc.sub_weight", f32["head"])]:
        got = b16["head"] if n == "pfc.sub_weight" else b16["picks"][n]
        e, cs = rel_err(got, ref), cosine(got, ref)
        print("   %-46s bf16 vs f32 rel err %.3e  cosine %.4f" % (n, e, cs))
        assert e < 0.65 and cs > 0.80, (n, e, cs)
    assert b16["peak_gb"] < 60 and f32["peak_gb"] < 120          # 288 GB per GPU: > 150 GB of headroom in either mode
