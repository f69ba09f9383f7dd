"""PartialFC on the HIP path: class-parallel ArcFace/CosFace head over RCCL.

Mirrors headers/partial_fc.py of the reference: same constructor, attributes (weight,
weight_mom, sub_weight, sub_weight_mom, index, stream), `forward_backward(label, features,
optimizer) -> (x_grad, loss_v)`, `update()`, `save_params()`, `parameters()`, checkpoint file
names (`rank:{r}_softmax_weight[_mom].pt`).  Differences, all forced by the reference itself:

* `margin_softmax` (no such callable exists in the reference) is a margin
  DESCRIPTOR -- `ArcMargin(s, m, a, k)` / `CosMargin(...)` below -- because the margin is fused
  into the softmax kernels (the logits are never materialised); its math restates
  AMArcFace.forward / AMCosFace.forward (margin_losses.py:390-418 / :277-303).
* the three tiny all-reduces (row max, denominator, target prob; partial_fc.py:136,141,162)
  become: one online (max, sumexp) local pass -> all_reduce(MAX) -> rescale -> all_reduce(SUM),
  and the loss all-reduce; two passes over the local logits instead of three.
* sample_rate < 1 (negative sampling, partial_fc.py:82-94) is not built.

The distributed logic lives here; the local arithmetic is behind `self.backend`, which is the
HIP library in the product (the only backend msml_amd ships) -- tests inject a CPU oracle
backend to exercise the collectives under gloo.
"""
import logging
import os

import torch
import torch.distributed as dist
from torch.nn import Module
from torch.nn.parameter import Parameter


_FORCE = bool(os.environ.get("MSML_FORCE_DIST"))   # run the collectives even at world size 1


class _Margin:
    kind = "arc"

    def __init__(self, s=64.0, m=0.5, a=0.0, k=0.0):
        self.s, self.m, self.a, self.k = float(s), float(m), float(a), float(k)

    def __call__(self, logits, labels):
        raise TypeError("msml_amd margins are descriptors consumed by the fused HIP kernels")


class ArcMargin(_Margin):
    """s*cos(theta + m - k(theta - a)) on the target logit (margin_losses.py:390-418)."""
    kind = "arc"


class CosMargin(_Margin):
    """s*(cos(theta) - m + k(theta - a)) on the target logit (margin_losses.py:277-303)."""
    kind = "cos"


class HipBackend:
    """Local arithmetic of one rank on libmsml_hip.so."""

    def __init__(self, dtype):
        from .. import ops
        from .._lib import TORCH_DTYPE, call
        self.ops, self.call, self.dtype = ops, call, dtype
        self.tdt = TORCH_DTYPE[dtype]

    def local_stats(self, total_features, sub_weight, labels, margin):
        """normalize(W), cos = X Wn^T, online (rowmax, rowsumexp) with the margin applied."""
        ops, call = self.ops, self.call
        from .. import functional as Fh
        n, e = total_features.shape
        c = sub_weight.shape[0]
        dev = total_features.device
        cp = ops.cpad(c)
        kop = (cp + ops.tile_n(cp) - 1) // ops.tile_n(cp) * ops.tile_n(cp)
        wn = torch.empty(kop, e, dtype=self.tdt, device=dev)
        inv_w = torch.empty(c, dtype=torch.float32, device=dev)
        call("msml_rownorm_fwd", sub_weight.detach(), c, kop, e, wn, e, inv_w, self.dtype)
        xs = total_features.to(self.tdt).reshape(n, 1, 1, e).contiguous()
        cosm, _ = ops.conv2d(xs, None, wn, None, cp, 1, 1, 1, 0, 0, False, out_dtype=0)
        cosm = cosm.reshape(n, cp)
        rowmax = torch.empty(n, dtype=torch.float32, device=dev)
        rowsum = torch.empty(n, dtype=torch.float32, device=dev)
        kind = Fh.HEAD_KIND[margin.kind]
        call("msml_pfc_rowstats", cosm, cp, n, c, labels, kind, margin.s, margin.m, margin.a,
             margin.k, rowmax, rowsum)
        state = (xs, wn, inv_w, cosm, kind)
        return state, rowmax, rowsum

    def local_grads(self, state, sub_weight, labels, margin, gmax, gsum, n_total, eps_ls):
        """grad = (p - y_smooth)/N through the margin; returns (ptarget, dX (N,E), dW (C,E))."""
        ops, call = self.ops, self.call
        xs, wn, inv_w, cosm, kind = state
        n, cp = cosm.shape
        c, e = sub_weight.shape
        dev = cosm.device
        dcos = torch.empty(n, 1, 1, cp, dtype=self.tdt, device=dev)
        ptarget = torch.empty(n, dtype=torch.float32, device=dev)
        call("msml_pfc_grad", cosm, cp, n, c, labels, kind, margin.s, margin.m, margin.a, margin.k,
             gmax, gsum, eps_ls, 1.0 / n_total, dcos, cp, ptarget, self.dtype)
        # Wn^T [E][Cp]: classes become the contiguous K of the dX GEMM
        wnt = torch.empty(e, ops.kpad(cp), dtype=self.tdt, device=dev)     # K padded to 32
        call("msml_transpose", wn, c, e, e, wnt, ops.kpad(cp), self.dtype)
        if self.dtype == 1:        # bf16: K = classes is huge and there are only 8 output tiles
            dx = ops.gemm_splitk(dcos.reshape(n, cp), wnt, e)
        else:
            dx, _ = ops.conv2d(dcos, None, wnt, None, e, 1, 1, 1, 0, 0, False, out_dtype=0)
        dwn = torch.empty(c, e, dtype=torch.float32, device=dev)
        ops.conv_wgrad(dcos, xs, dwn, c, e, e, 0, 1, 1, 1, 0, 0)
        dw = torch.empty(c, e, dtype=torch.float32, device=dev)
        call("msml_rownorm_bwd", sub_weight.detach(), inv_w, dwn, e, c, e, dw, 0)
        return ptarget, dx.reshape(n, e), dw


class PartialFC(Module):
    @torch.no_grad()
    def __init__(self, rank, local_rank, world_size, batch_size, resume, margin_softmax, num_classes,
                 sample_rate=1.0, embedding_size=512, prefix="./", fp16=False, backend=None,
                 device=None):
        super().__init__()
        if int(sample_rate) != 1:
            raise NotImplementedError("msml_amd: PartialFC negative sampling (sample_rate<1) is not built")
        if not isinstance(margin_softmax, _Margin):
            raise TypeError("margin_softmax must be msml_amd.headers.ArcMargin / CosMargin "
                            "(the margin is fused into the HIP softmax kernels)")
        self.num_classes, self.rank, self.local_rank = num_classes, rank, local_rank
        self.device = device if device is not None else torch.device("cuda:{}".format(local_rank))
        self.world_size, self.batch_size = world_size, batch_size
        self.margin_softmax = margin_softmax
        self.sample_rate, self.embedding_size, self.prefix = sample_rate, embedding_size, prefix
        self.num_local = num_classes // world_size + int(rank < num_classes % world_size)
        self.class_start = num_classes // world_size * rank + min(rank, num_classes % world_size)
        self.num_sample = int(self.sample_rate * self.num_local)
        self.weight_name = os.path.join(prefix, "rank:{}_softmax_weight.pt".format(rank))
        self.weight_mom_name = os.path.join(prefix, "rank:{}_softmax_weight_mom.pt".format(rank))
        self.weight = self.weight_mom = None
        if resume:
            try:
                self.weight = torch.load(self.weight_name).to(self.device)
                logging.info("softmax weight resume successfully!")
            except (FileNotFoundError, KeyError, IndexError):
                logging.info("softmax weight resume fail!")
            try:
                self.weight_mom = torch.load(self.weight_mom_name).to(self.device)
                logging.info("softmax weight mom resume successfully!")
            except (FileNotFoundError, KeyError, IndexError):
                logging.info("softmax weight mom resume fail!")
        if self.weight is None:
            self.weight = torch.normal(0, 0.01, (self.num_local, embedding_size), device=self.device)
        if self.weight_mom is None:
            self.weight_mom = torch.zeros_like(self.weight)
        self.stream = torch.cuda.Stream(local_rank) if self.device.type == "cuda" else None
        self.index = None
        self.update = lambda: 0
        self.sub_weight = Parameter(self.weight)
        self.sub_weight_mom = self.weight_mom
        if backend is None:
            from .._lib import BF16, F32
            backend = HipBackend(BF16 if fp16 else F32)
        self.backend = backend
        self.eps_ls = 0.1

    def save_params(self):
        torch.save(self.weight.data, self.weight_name)
        torch.save(self.weight_mom, self.weight_mom_name)

    @torch.no_grad()
    def sample(self, total_label):
        # in place like the reference (partial_fc.py:78-81) but without boolean-mask indexing,
        # which would synchronise with the host (and cannot be captured into a hipGraph)
        index_positive = (self.class_start <= total_label) & (total_label < self.class_start + self.num_local)
        total_label.copy_(torch.where(index_positive, total_label - self.class_start,
                                      torch.full_like(total_label, -1)))

    def _all_gather(self, x):
        if self.world_size == 1 and not _FORCE:
            return x.clone()
        out = torch.zeros((self.batch_size * self.world_size,) + tuple(x.shape[1:]), dtype=x.dtype,
                          device=x.device)
        dist.all_gather(list(out.chunk(self.world_size, dim=0)), x.contiguous())
        return out

    def prepare(self, label, optimizer):
        total_label = self._all_gather(label.to(self.device, torch.long))
        self.sample(total_label)
        if optimizer is not None:
            optimizer.state.pop(optimizer.param_groups[-1]["params"][0], None)
            optimizer.param_groups[-1]["params"][0] = self.sub_weight
            optimizer.state[self.sub_weight]["momentum_buffer"] = self.sub_weight_mom
        return total_label

    @torch.no_grad()
    def forward_backward(self, label, features, optimizer):
        total_label = self.prepare(label, optimizer)
        total_features = self._all_gather(features.data.float())
        n_total = self.batch_size * self.world_size
        state, rowmax, rowsum = self.backend.local_stats(total_features, self.sub_weight, total_label,
                                                         self.margin_softmax)
        if self.world_size > 1 or _FORCE:
            gmax = rowmax.clone()
            dist.all_reduce(gmax, dist.ReduceOp.MAX)
            gsum = rowsum * torch.exp(rowmax - gmax)
            dist.all_reduce(gsum, dist.ReduceOp.SUM)
        else:
            gmax, gsum = rowmax, rowsum
        ptarget, dx_total, dw = self.backend.local_grads(state, self.sub_weight, total_label,
                                                        self.margin_softmax, gmax, gsum, n_total,
                                                        self.eps_ls)
        if self.world_size > 1 or _FORCE:
            dist.all_reduce(ptarget, dist.ReduceOp.SUM)
        loss_v = ptarget.clamp_min(1e-30).log().mean() * (-1)
        self.sub_weight.grad = dw
        if self.world_size > 1 or _FORCE:
            x_grad = torch.zeros_like(features, dtype=torch.float32)
            dist.reduce_scatter(x_grad, list(dx_total.contiguous().chunk(self.world_size, dim=0)))
            x_grad = x_grad * self.world_size
        else:
            x_grad = dx_total
        return x_grad, loss_v
