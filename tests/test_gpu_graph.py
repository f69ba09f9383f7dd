"""hipGraph capture of the whole training step (what `bench.py --launch graph` replays): replays must walk the same
trajectory as eager steps.  Besides the kernels this pins the per-step host-side state that a replay does not re-run:
the pack-cache refresh, the padded-bias refresh and the zero fill of the BatchNorm accumulator arena (ops._AccArena) --
an accumulator that a replay did not re-zero would add the statistics of consecutive steps."""
import pytest
import torch

from msml_amd import ops, synthetic
from msml_amd.backbones import MSML
from msml_amd.optim import FlatSGD
from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
from oracle.fill import fill_module

pytestmark = pytest.mark.gpu
PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}


def _run(use_graph, steps=3):
    torch.manual_seed(0)
    m = fill_module(MSML("iresnet18", "unet", (1, 1, 1, 1), 50, fp16=True, fm_params=(3, 2, "sigmoid", "mul"),
                         header_type="AMArcFace", peer_params=dict(PEER_OFF))).cuda().train()
    opt = FlatSGD([{"params": [p for p in m.parameters() if p.requires_grad], "lr": 0.01}], 0.9, 5e-4, 5.0)
    x = synthetic.images(8, seed=5)
    x, msk = synthetic.rect_occlusion(x, seed=5)
    x, msk, lab = x.cuda(), msk.cuda(), synthetic.labels(8, 50, seed=5).cuda()
    crit = StructureConsensuLossFunction(10.0, 5.0)

    def step():
        opt.zero_grad()
        cls, seg, _ = m(x, lab)
        loss = torch.nn.functional.cross_entropy(cls, lab) + crit(seg, msk, msk)
        loss.backward()
        opt.step()
        return loss.detach()

    try:
        side = torch.cuda.Stream()                       # PyTorch's capture recipe: warm up on a side stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        losses = []
        if use_graph:
            if use_graph == "second":                    # a first capture that is never replayed, then -- with no
                first = torch.cuda.CUDAGraph()           # eager step in between -- the capture that is
                with torch.cuda.graph(first):
                    step()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
            losses.append(float(out))                    # (the capture itself does not run the step)
            losses.clear()
            for _ in range(steps):
                graph.replay()
                losses.append(float(out))
        else:
            for _ in range(steps):
                losses.append(float(step()))
        torch.cuda.synchronize()
        return losses, opt.flat_w.clone(), {n: b.clone() for n, b in m.named_buffers() if "running_var" in n}
    finally:
        opt.release()


def test_graph_replay_walks_the_eager_trajectory():
    assert ops.ACC_STATS
    le, we, re_ = _run(False)
    lg, wg, rg = _run(True)
    print("eager losses", le, "graph losses", lg)
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(le, lg)), (le, lg)
    assert le[-1] < le[0]                                # and it trains
    assert float((we - wg).abs().max()) <= 1e-6 * float(we.abs().max())
    for n in re_:
        assert torch.allclose(re_[n], rg[n], rtol=1e-5, atol=1e-7), n


def test_second_capture_gets_its_own_accumulators():
    """Two captures back to back: the second graph's BatchNorm accumulators must come from a chunk whose
    zero fill the SECOND graph replays; cut from the first capture's chunk they would keep the sums of the previous
    replay and corrupt mean / var from the second replay on."""
    le, we, _ = _run(False)
    lg, wg, _ = _run("second")
    print("eager losses", le, "second-graph losses", lg)
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(le, lg)), (le, lg)
    assert float((we - wg).abs().max()) <= 1e-6 * float(we.abs().max())


def test_captured_step_follows_lr_changes_and_redraws_dropout():
    """a captured step must not bake host-side scalars that change between steps -- the learning rate
    (LambdaLR.step() -> FlatSGD.set_lr_factor, train.py:193-196) lives in device memory, and a captured dropout
    (iresnet.py:231) reads a device seed that the graph itself advances: replays draw different masks."""
    from msml_amd import functional as Fh
    w = torch.nn.Parameter(torch.ones(1024, device="cuda"))
    opt = FlatSGD([{"params": [w], "lr": 0.5}], 0.0, 0.0, None)
    x = torch.ones(8, 1, 1, 512, device="cuda", dtype=torch.bfloat16)

    def step():
        opt.zero_grad()
        w.grad.add_(1.0)
        opt.step()
        return Fh.dropout(x, 0.5)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                           # eager warm-up: w = 0.5
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = step()
        g.replay()
        m1 = y.clone()
        assert abs(float(w[0]) - 0.0) < 1e-6                 # 0.5 - 0.5 * 1
        opt.set_lr_factor(0.1)                               # LambdaLR.step() between replays
        g.replay()
        m2 = y.clone()
        assert abs(float(w[0]) + 0.05) < 1e-6                # lr 0.05 now: 0 - 0.05
        assert not torch.equal(m1, m2)                       # a new mask per replay
        keep = float((m2 != 0).float().mean())
        assert 0.4 < keep < 0.6 and set(m2.float().unique().tolist()) <= {0.0, 2.0}
    finally:
        opt.release()
