"""One rank of the two-ranks-on-one-GPU PartialFC test (started by tests/conftest.py as a FRESH process
before the pytest process touches the GPU; see test_partial_fc_hip_two_ranks_one_gpu).  Each rank
initialises the GPU itself, joins a gloo group, runs PartialFC.forward_backward on the HIP backend
(f32 and bf16) against device 0 and writes its results."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _diff_report(a, b, what):
    """One line on where two flat gradients differ (count, largest difference, first / last offset, non-finite values)."""
    ne = (a != b) | (a.isnan() != b.isnan())
    k = int(ne.sum().item())
    if k == 0:
        return "%s: identical" % what
    idx = ne.nonzero().flatten()
    d = (a.double() - b.double()).abs()
    d = torch.where(d.isnan(), torch.zeros_like(d), d)
    return ("%s: %d of %d elements differ, max |diff| %.3e (|a| max %.3e), offsets %d..%d, non-finite a %d b %d"
            % (what, k, a.numel(), float(d.max().item()), float(a.abs().nan_to_num().max().item()), int(idx[0].item()),
               int(idx[-1].item()), int((~a.isfinite()).sum().item()), int((~b.isfinite()).sum().item())))


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from msml_amd.headers.partial_fc import ArcMargin, PartialFC
    from oracle.inputs import PFC_B, PFC_C, PFC_E, pfc_inputs
    from tests.helpers import pick
    feat, label, w = pfc_inputs(world, rank)
    out = {}
    for tag, fp16 in (("f32", False), ("bf16", True)):
        p = PartialFC(rank, 0, world, PFC_B, False, ArcMargin(64.0, 0.48, 0.0, 0.0), PFC_C,
                      embedding_size=PFC_E, fp16=fp16)
        assert p.weight.shape == w.shape and p.weight.is_cuda
        with torch.no_grad():
            p.weight.copy_(w)
        opt = torch.optim.SGD([{"params": p.parameters()}], lr=0.1 / 512 * PFC_B * world, momentum=0.9,
                              weight_decay=5e-4)
        p.prefetch_labels(label.cuda())              # the side-stream label gather
        x_grad, loss_v = p.forward_backward(label.cuda(), feat.cuda(), opt)
        wgrad = p.sub_weight.grad.clone()
        opt.step()
        torch.cuda.synchronize()
        out[tag + "_loss"] = loss_v.item()
        out[tag + "_x_grad"] = x_grad.cpu().numpy()
        out[tag + "_wgrad_pick"] = pick(wgrad, 256)
        out[tag + "_wnew_pick"] = pick(p.sub_weight.data, 256)
    # ---- bucketed gradient all-reduce overlapped with the backward (FlatSGD.enable_overlap) at world size 2,
    # both side streams on, against the plain all_reduce_grads path on the same gradients (a bucket
    # fired from a side-stream autograd node must also wait for the training stream)
    from msml_amd import ops, synthetic
    from msml_amd.backbones import MSML
    from msml_amd.optim import FlatSGD, reference_param_groups
    from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
    from oracle.fill import fill_module
    peer = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}

    def grads(overlap, comm_dtype=None):
        torch.manual_seed(0)
        m = fill_module(MSML("iresnet18", "unet", (1, 1, 1, 1), 50, fp16=True, fm_params=(3, 2, "sigmoid", "mul"),
                             header_type="AMArcFace", peer_params=dict(peer))).cuda().train()
        opt = FlatSGD(reference_param_groups(m, 2, world), 0.9, 5e-4, 5.0)
        opt.comm_dtype = comm_dtype
        if overlap:
            opt.enable_overlap(world, bucket_bytes=8 << 20)
        ops.WGRAD_STREAM, ops.OSB_STREAM = torch.cuda.Stream(), torch.cuda.Stream()
        x = synthetic.images(2, seed=10 + rank)               # different data per rank
        x, msk = synthetic.rect_occlusion(x, seed=10 + rank)
        lab = synthetic.labels(2, 50, seed=10 + rank)
        try:
            opt.zero_grad()
            cls, seg, _ = m(x.cuda(), lab.cuda())
            loss = torch.nn.functional.cross_entropy(cls, lab.cuda()) + \
                StructureConsensuLossFunction(10.0, 5.0)(seg, msk.cuda(), msk.cuda())
            loss.backward()
            fired = sum(opt.fired) if overlap else 0
            if overlap:
                out["ddp_duplicate_reports"] = out.get("ddp_duplicate_reports", 0) + opt.duplicate_reports
            opt.all_reduce_grads(world)
            torch.cuda.synchronize()
            return opt.averaged_grad().clone(), fired, (len(opt.buckets) if overlap else 0)
        finally:
            ops.WGRAD_STREAM = ops.OSB_STREAM = None
            opt.release()
    ga, _, _ = grads(False)
    gb, fired, nb = grads(True)
    out["ddp_equal"] = int(torch.equal(ga, gb))
    # what differs, should the two ever differ (the compute is run-to-run deterministic: static tile schedules, f64
    # accumulators): a second plain run separates "the backward itself is not reproducible" from "the overlapped reduce
    # raced", the offsets say which parameters
    ga2, _, _ = grads(False)
    out["ddp_plain_reproducible"] = int(torch.equal(ga, ga2))
    out["ddp_diag"] = _diff_report(ga, gb, "plain vs overlapped") + " | " + _diff_report(ga, ga2, "plain vs plain")
    out["ddp_fired_during_backward"] = fired
    out["ddp_buckets"] = nb
    out["ddp_gsum"] = float(ga.double().abs().sum().item())
    # bf16 gradient messages (MSML_GRAD_COMM=bf16 / opt.comm_dtype), overlap off and on (the copy-back of an
    # overlapped bf16 bucket must be ordered behind its all-reduce): both equal the f32 average to bf16 rounding and
    # equal each other bit for bit (same messages, same order of the two summands)
    gh, _, _ = grads(False, torch.bfloat16)
    gho, fired_h, _ = grads(True, torch.bfloat16)
    den = float(ga.double().norm().item())
    out["ddp_bf16_rel"] = float((gh.double() - ga.double()).norm().item()) / den
    out["ddp_bf16_overlap_rel"] = float((gho.double() - ga.double()).norm().item()) / den
    out["ddp_bf16_equal"] = int(torch.equal(gh, gho))
    out["ddp_bf16_fired_during_backward"] = fired_h
    np.savez(os.path.join(outdir, "r%d.npz" % rank), **out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
