"""PartialFC on the HIP path: class-parallel ArcFace/CosFace head over RCCL.

Mirrors headers/partial_fc.py of the reference: same constructor, attributes (weight,
weight_mom, sub_weight, sub_weight_mom, index, stream), `forward_backward(label, features,
optimizer) -> (x_grad, loss_v)`, `update()`, `save_params()`, `parameters()`, checkpoint file
names (`rank:{r}_softmax_weight[_mom].pt`), negative sampling (`sample_rate < 1`, :82-94,101-104).
Differences, all forced by the reference itself:

* `margin_softmax` (no such callable exists in the reference) is a margin
  DESCRIPTOR -- `ArcMargin(s, m, a, k)` / `CosMargin(...)` below -- because the margin is fused
  into the softmax kernels (the logits are never materialised); its math restates
  AMArcFace.forward / AMCosFace.forward (margin_losses.py:390-418 / :277-303).
* collectives (partial_fc.py:106-177 has five on the critical path: labels, features, row max,
  denominator, target prob, then the reduce-scatter):
    - labels are gathered on the side stream `self.stream` like the reference (:107-110) -- they
      are known before the backbone forward, `prefetch_labels()` lets a caller start it there;
    - features: one `all_gather_into_tensor`;
    - (row max, row sum-exp) of the local online-softmax pass travel in ONE all-gather of [N, 2]
      pairs and are combined locally (gmax = max_r, gsum = sum_r s_r exp(m_r - gmax)) -- one
      collective instead of the dependent MAX + SUM all-reduces (:136,141);
    - dX: `reduce_scatter_tensor`, issued BEFORE the target-probability all-reduce (:162), which
      only feeds the logged loss.
  xGMI is a point-to-point mesh: these are latency-bound 8 KB - 4 MB messages, so what matters is
  their count in front of the backbone backward (3: features, pairs, dX) and that the OSB backward,
  which does not depend on the head, is already running meanwhile (bench.py / INTEGRATION.md issue
  it first when world_size > 1).

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

    def local_grads(self, state, sub_weight, labels, margin, gmax, gsum, n_total, eps_ls, dw_out=None):
        """grad = (p - y_smooth)/N through the margin; returns (ptarget, dX (N,E), dW (C,E)).
        dw_out: write dW there (the parameter's .grad view of a flat gradient arena) instead of a
        fresh tensor."""
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
        dw = dw_out if dw_out is not None else torch.empty(c, e, dtype=torch.float32, device=dev)
        call("msml_rownorm_bwd", sub_weight.detach(), inv_w, dwn, e, c, e, dw, 0)
        return ptarget, dx.reshape(n, e), dw


class PartialFC(Module):
    @torch.no_grad()
    def __init__(self, rank, local_rank, world_size, batch_size, resume, margin_softmax, num_classes,
                 sample_rate=1.0, embedding_size=512, prefix="./", fp16=False, backend=None,
                 device=None):
        super().__init__()
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
        self.full = int(self.sample_rate) == 1
        if self.full:
            self.sub_weight = Parameter(self.weight)
            self.sub_weight_mom = self.weight_mom
        else:
            self.sub_weight = Parameter(torch.empty((0, 0), device=self.device))
            self.sub_weight_mom = None
        self._k = None                # sampled mode on a flat-arena optimizer: rows of sub_weight in use this step
        self._flat_param = None       # the fixed-capacity parameter handed out by flat_parameter()
        if backend is None:
            from .._lib import BF16, F32
            backend = HipBackend(BF16 if fp16 else F32)
        self.backend = backend
        self.eps_ls = 0.1
        self._stage = None            # collectives staged through the host (gloo group, device tensors)
        self._flat = None             # FlatSGD that re-homed sub_weight (adopt_flat_optimizer)
        self._label_job = None        # (id, total_label, event, label) of a prefetched label gather
        self.perm_fn = None           # tests: replaces torch.rand in sample()

    # ---- checkpoint ----------------------------------------------------------------------------
    def save_params(self):
        """rank:{r}_softmax_weight.pt / _mom.pt as plain tensors (partial_fc.py:73-75).  Saved from
        the LIVE tensors: with a flat-arena optimizer the parameter and its momentum are views into
        the arenas (clone: torch.save would otherwise serialise the whole arena storage)."""
        if not self.full and self.index is not None:
            self.update()
        w = self.sub_weight.data if self.full else self.weight
        mom = self.sub_weight_mom if self.full else self.weight_mom
        torch.save(w.detach().clone(), self.weight_name)
        torch.save(mom.detach().clone(), self.weight_mom_name)

    def adopt_flat_optimizer(self, opt):
        """After `opt = FlatSGD([{'params': [pfc.sub_weight], ...}])` re-homed the parameter: point
        weight / weight_mom / sub_weight_mom at the arena views and seed the arena momentum from
        the (possibly resumed) momentum, so that save_params() and resume see the trained state."""
        if not self.full:
            # negative sampling (partial_fc.py:82-94,101-116 swaps the optimizer's parameter and momentum buffer for
            # the sampled rows every step): here the optimizer keeps ONE fixed-capacity parameter in its arenas
            # (flat_parameter()), sample() gathers the sampled rows and their momentum into the arena views and
            # update() scatters them back -- no re-registration, no host sync on the static branch
            if self._flat_param is None or self.sub_weight is not self._flat_param or id(self.sub_weight) not in opt.offsets:
                raise ValueError("negative sampling on a flat-arena optimizer: build it over pfc.flat_parameter() "
                                 "(and adopt it before the first prefetch_labels() / forward_backward())")
            if opt.max_norm is not None:
                # rows [k, capacity) of the arena parameter are not part of the step: they must not enter a norm
                raise ValueError("the head's FlatSGD must be built with max_norm=None (the reference never clips the "
                                 "PartialFC weight, train.py:267-277)")
            self._flat_mom = opt.momentum_view(self.sub_weight)
            opt.steps = max(opt.steps, 1)          # the gathered momentum is never a "first step" buffer
            self._flat = opt
            return
        mview = opt.momentum_view(self.sub_weight)
        mview.copy_(self.weight_mom)
        if self.weight_mom.abs().sum().item() != 0:
            opt.steps = max(opt.steps, 1)          # a resumed buffer is not the first step
        self.weight = self.sub_weight.data
        self.weight_mom = self.sub_weight_mom = mview
        self._flat = opt

    def flat_capacity(self):
        """Rows a flat-arena optimizer must hold in sampled mode: num_sample, or the number of positives a batch can
        bring when that is larger (the reference then keeps exactly the positives, partial_fc.py:89-90)."""
        return max(self.num_sample, min(self.batch_size * self.world_size, self.num_local))

    def flat_parameter(self):
        """Sampled mode on the fast optimizer: the fixed-capacity parameter to build FlatSGD over
        (`FlatSGD([{'params': [pfc.flat_parameter()], ...}])`, then `pfc.adopt_flat_optimizer(opt)`); it becomes
        `pfc.sub_weight`, whose first len(index) rows are the sampled classes of the current step."""
        if self.full:
            return self.sub_weight
        self.sub_weight = self._flat_param = Parameter(torch.zeros((self.flat_capacity(), self.embedding_size),
                                                                   device=self.device))
        return self.sub_weight

    def _active(self):
        """(weight, grad) of the classes in use this step: the whole parameter, or the first k rows of the
        fixed-capacity arena parameter in sampled + flat mode."""
        w = self.sub_weight
        if self._flat is None or self.full or self._k is None or self._k == w.shape[0]:
            return w, w.grad
        return w.data[:self._k], (w.grad[:self._k] if w.grad is not None else None)

    # ---- label mapping / negative sampling ------------------------------------------------------
    @torch.no_grad()
    def sample(self, total_label):
        """partial_fc.py:77-94.  In place like the reference but without boolean-mask indexing or
        torch.unique, which synchronise with the host (and cannot be captured into a hipGraph)."""
        index_positive = (self.class_start <= total_label) & (total_label < self.class_start + self.num_local)
        local = torch.where(index_positive, total_label - self.class_start, torch.full_like(total_label, -1))
        if self.full:
            total_label.copy_(local)
            return
        n = total_label.numel()
        if self.num_sample >= min(n, self.num_local):
            # every positive class is kept and the rest of the budget is filled with random
            # negatives: rand -> positives forced to 2.0 -> top-k -> sorted (:84-88).  The scatter of
            # duplicates replaces torch.unique; the result is the same index set.
            perm = (self.perm_fn or torch.rand)(self.num_local, device=self.device)
            hit = torch.zeros(self.num_local + 1, dtype=torch.bool, device=self.device)
            hit[torch.where(index_positive, local, torch.full_like(local, self.num_local))] = True
            perm = torch.where(hit[:self.num_local], torch.full_like(perm, 2.0), perm)
            index = torch.topk(perm, k=self.num_sample)[1].sort()[0]
        else:
            # fewer samples than labels in the batch can mean fewer than positives: the reference
            # then keeps exactly the positives (:89-90) -- data-dependent size, host sync
            positive = torch.unique(local[index_positive], sorted=True)
            if self.num_sample - positive.size(0) >= 0:
                perm = (self.perm_fn or torch.rand)(self.num_local, device=self.device)
                perm[positive] = 2.0
                index = torch.topk(perm, k=self.num_sample)[1].sort()[0]
            else:
                index = positive
        self.index = index
        mapped = torch.searchsorted(index, local.clamp_min(0))
        total_label.copy_(torch.where(index_positive, mapped, torch.full_like(total_label, -1)))
        if self._flat is not None:
            # fast optimizer: gather the rows and their momentum into the arena views of the fixed parameter
            k = self._k = int(index.shape[0])      # static on the first branch; the second one already synchronised
            torch.index_select(self.weight, 0, index, out=self.sub_weight.data[:k])
            torch.index_select(self.weight_mom, 0, index, out=self._flat_mom[:k])
            self.sub_weight_mom = self._flat_mom[:k]
            return
        if self._flat_param is not None:
            # flat_parameter() was handed to a FlatSGD that has not been adopted yet (prefetch_labels() before the first
            # forward_backward(): ADVICE r3) -- replacing sub_weight here would orphan the arena parameter
            raise RuntimeError("PartialFC: call pfc.adopt_flat_optimizer(opt) right after building FlatSGD over "
                               "pfc.flat_parameter(), before prefetch_labels()")
        self.sub_weight = Parameter(self.weight[index])
        self.sub_weight_mom = self.weight_mom[index]

    @torch.no_grad()
    def update(self):
        """Write the sampled rows back (partial_fc.py:101-104); nothing to do at sample_rate 1."""
        if self.full or self.index is None:
            return 0
        if self._flat is not None:
            self.weight_mom.index_copy_(0, self.index, self._flat_mom[:self._k])
            self.weight.index_copy_(0, self.index, self.sub_weight.data[:self._k])
            return 0
        self.weight_mom[self.index] = self.sub_weight_mom
        self.weight[self.index] = self.sub_weight.data
        return 0

    # ---- collectives ---------------------------------------------------------------------------
    def _dist(self):
        return self.world_size > 1 or _FORCE

    def _staged(self):
        """gloo moves host memory only (its CUDA support covers broadcast / all_reduce): under a gloo
        group with device tensors -- the two-ranks-on-one-GPU test, debugging without RCCL -- the
        collectives are staged through the host.  RCCL (backend 'nccl') works on device memory."""
        if self._stage is None:
            self._stage = self.device.type == "cuda" and dist.is_initialized() and dist.get_backend() == "gloo"
        return self._stage

    def _all_gather(self, x):
        if not self._dist():
            return x.clone()
        if self._staged():
            xc = x.contiguous().cpu()
            out = torch.empty((xc.shape[0] * self.world_size,) + tuple(xc.shape[1:]), dtype=xc.dtype)
            dist.all_gather_into_tensor(out, xc)
            return out.to(x.device)
        out = torch.empty((x.shape[0] * self.world_size,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out

    def _reduce_scatter(self, dx_total, like):
        if self._staged():
            out = torch.empty(like.shape, dtype=torch.float32)
            dist.reduce_scatter_tensor(out, dx_total.contiguous().cpu())
            return out.to(like.device)
        out = torch.empty_like(like, dtype=torch.float32)
        dist.reduce_scatter_tensor(out, dx_total.contiguous())
        return out

    def _all_reduce_sum(self, t):
        if self._staged():
            c = t.cpu()
            dist.all_reduce(c, dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, dist.ReduceOp.SUM)

    def prefetch_labels(self, label):
        """Start the label all-gather + local mapping on the side stream (partial_fc.py:107-110); call it
        before the backbone forward -- prepare() picks the result up when it is handed the SAME tensor object,
        unmodified (identity + version counter; a label buffer refilled in place, or another tensor that happens to
        live at the same address, gathers again).  The label must not be written to between the two calls."""
        given = label
        label = label.to(self.device, torch.long)
        if self.stream is None or not self._dist():
            total = self._all_gather(label)
            self.sample(total)
            self._label_job = ((given, given._version), total, None, label)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            total = self._all_gather(label)
            self.sample(total)
            ev = self.stream.record_event()
        label.record_stream(self.stream)
        self._label_job = ((given, given._version), total, ev, label)

    def prepare(self, label, optimizer):
        if optimizer is not None and self._flat is not optimizer:
            from ..optim import FlatSGD
            if isinstance(optimizer, FlatSGD):       # before sample(): it gathers into the optimizer's arenas
                self.adopt_flat_optimizer(optimizer)
        job, self._label_job = self._label_job, None
        if job is None or job[0][0] is not label or job[0][1] != label._version:
            self.prefetch_labels(label)
            job, self._label_job = self._label_job, None
        _, total_label, ev, _ = job
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)       # partial_fc.py:97
            total_label.record_stream(torch.cuda.current_stream())
        if optimizer is not None:
            from ..optim import FlatSGD
            if isinstance(optimizer, FlatSGD):
                if self._flat is not optimizer:
                    self.adopt_flat_optimizer(optimizer)
            else:
                optimizer.state.pop(optimizer.param_groups[-1]["params"][0], None)
                optimizer.param_groups[-1]["params"][0] = self.sub_weight
                optimizer.state[self.sub_weight]["momentum_buffer"] = self.sub_weight_mom
        return total_label

    @torch.no_grad()
    def forward_backward(self, label, features, optimizer):
        total_label = self.prepare(label, optimizer)
        total_features = self._all_gather(features.data.float())
        n_total = self.batch_size * self.world_size
        sub_w, g = self._active()
        state, rowmax, rowsum = self.backend.local_stats(total_features, sub_w, total_label, self.margin_softmax)
        if self._dist():
            # ONE collective for the softmax denominator: gather every rank's (max, sum-exp) pair and
            # combine locally (same value on every rank, fixed summation order r = 0..W-1)
            pair = torch.stack((rowmax, rowsum), 1)                          # [N, 2]
            allp = self._all_gather(pair).view(self.world_size, -1, 2)       # [W, N, 2]
            gmax = allp[:, :, 0].max(0)[0]
            gsum = (allp[:, :, 1] * torch.exp(allp[:, :, 0] - gmax)).sum(0)
        else:
            gmax, gsum = rowmax, rowsum
        dw_out = g if (self._flat is not None and g is not None and g.shape == sub_w.shape
                       and g.is_contiguous()) else None       # straight into the flat gradient arena
        ptarget, dx_total, dw = self.backend.local_grads(state, sub_w, total_label,
                                                        self.margin_softmax, gmax, gsum, n_total,
                                                        self.eps_ls, dw_out=dw_out)
        if dw_out is None:
            self.sub_weight.grad = dw
        if self._dist():
            # the critical collective first: dX feeds the backbone backward
            x_grad = self._reduce_scatter(dx_total, features) * self.world_size
            # loss: the target probability lives on exactly one rank per row -> SUM (8 KB, logged only)
            self._all_reduce_sum(ptarget)
        else:
            x_grad = dx_total
        loss_v = ptarget.clamp_min(1e-30).log().mean() * (-1)
        return x_grad, loss_v
