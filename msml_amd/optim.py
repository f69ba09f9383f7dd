"""Flat-arena SGD with fused gradient clipping (the training glue of train.py:153-196,267-277).

All parameters of a model are re-homed as views into ONE flat f32 buffer per learning-rate
group, their .grad as views into a matching flat gradient buffer and their momentum likewise,
so that per step the optimizer is: one L2-norm reduction over the flat gradient
(clip_grad_norm_(5, 2)) + one fused clip*grad -> weight-decay -> momentum -> update kernel per
group, with no host synchronisation; DDP reduces the same flat gradient in a few large
messages (sized for xGMI, not for NVSwitch).  Semantics equal torch.optim.SGD(momentum,
weight_decay) + torch.nn.utils.clip_grad_norm_.
"""
import torch
import torch.distributed as dist

from ._lib import call


FORCE_COLLECTIVES = bool(__import__("os").environ.get("MSML_FORCE_DIST"))


def reference_param_groups(model, batch_size, world_size, lr=0.1):
    """LR groups of the reference (train.py:153-178, conf.pretrained False): parameters whose
    name contains 'osb' train at 0.01/512*bs*W, everything else at lr/512*bs*W."""
    base = lr / 512 * batch_size * world_size
    osb = 0.01 / 512 * batch_size * world_size
    groups = {"osb": ([], osb), "rest": ([], base)}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        groups["osb" if "osb" in name else "rest"][0].append(p)
    return [{"params": ps, "lr": glr} for ps, glr in groups.values() if ps]


def lr_factor_ms1m(epoch):
    """LambdaLR factor of the reference's ms1m recipe (config.py:35-39): 0.1 ** #{m in (11, 17, 22):
    m - 1 <= epoch}, warm-up disabled (warmup_epoch = -1)."""
    return 0.1 ** sum(1 for m in (11, 17, 22) if m - 1 <= epoch)


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR for FlatSGD (train.py:193-196): step() once per epoch."""

    def __init__(self, optimizer, lr_lambda, last_epoch=-1):
        self.optimizer, self.lr_lambda, self.last_epoch = optimizer, lr_lambda, last_epoch
        self.step()

    def step(self):
        self.last_epoch += 1
        self.optimizer.set_lr_factor(self.lr_lambda(self.last_epoch))

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.groups]


class FlatSGD:
    def __init__(self, param_groups, momentum=0.9, weight_decay=5e-4, max_norm=5.0):
        self.momentum, self.weight_decay, self.max_norm = momentum, weight_decay, max_norm
        self.groups = []
        params = [p for g in param_groups for p in g["params"]]
        dev = params[0].device
        total = sum((p.numel() + 3) // 4 * 4 for p in params)
        self.flat_w = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.offsets = {}
        for g in param_groups:
            start = off
            for p in g["params"]:
                n = p.numel()
                self.flat_w[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_w[off:off + n].view_as(p.data)
                p.grad = self.flat_g[off:off + n].view_as(p.data)
                self.offsets[id(p)] = off
                off += (n + 3) // 4 * 4
            self.groups.append({"start": start, "end": off, "lr": g["lr"], "base_lr": g["lr"]})
        self.params = params
        self.steps = 0
        for p in params:              # backward kernels accumulate into the arena views directly
            p._msml_arena = self
        self.norm_coef = torch.ones(2, dtype=torch.float32, device=dev)
        self.ws = torch.empty(1024, dtype=torch.float32, device=dev)
        # learning rates live on the device: a step captured into a hipGraph follows set_lr_factor() / LambdaLR
        self.lr_dev = torch.tensor([g["lr"] for g in self.groups], dtype=torch.float32, device=dev)
        # flat_g holds grad_scale^-1 times the gradient: the SUM over the ranks after all_reduce_grads(), whose
        # division by the world size is folded into the clip coefficient (no extra pass over the arena)
        self.grad_scale = 1.0
        self.comm_dtype = torch.bfloat16 if __import__("os").environ.get("MSML_GRAD_COMM", "") == "bf16" else None

    def momentum_view(self, p):
        """The momentum buffer of parameter `p` as a view into the flat momentum arena."""
        off = self.offsets[id(p)]
        return self.flat_m[off:off + p.numel()].view_as(p.data)

    def set_lr_factor(self, factor):
        for g in self.groups:
            g["lr"] = g["base_lr"] * factor
        self.lr_dev.copy_(torch.tensor([g["lr"] for g in self.groups], dtype=torch.float32), non_blocking=True)

    def averaged_grad(self):
        """The flat gradient as the optimizer will apply it before clipping: flat_g x grad_scale (a copy when a
        scale is pending, flat_g itself otherwise)."""
        return self.flat_g if self.grad_scale == 1.0 else self.flat_g * self.grad_scale

    def release(self):
        """Detach the parameters from the arena protocol (in-place gradients, overlap callbacks): they
        behave like plain parameters again (their data / grad stay views of the arenas).  A pending rank-sum scale
        (all_reduce_grads() without step()) is folded into the arena first, so that the .grad handed out is the
        averaged gradient."""
        if self.grad_scale != 1.0:
            self.flat_g.mul_(self.grad_scale)
            self.grad_scale = 1.0
        for p in self.params:
            p.__dict__.pop("_msml_arena", None)
            p.__dict__.pop("_msml_ready", None)

    def zero_grad(self):
        from . import ops
        ops.wgrad_drop()                     # leftovers of a backward pass that raised must not reach this step
        if ops.WGRAD_STREAM is not None:     # the side stream must see the zeroed arena
            ops.WGRAD_STREAM.wait_stream(torch.cuda.current_stream())
        if self.flat_g.is_cuda:
            self.train_stream = torch.cuda.current_stream()   # the stream backward kernels are issued on
        self.grad_scale = 1.0
        if getattr(self, "buckets", None) is not None and (self.reported or any(self.fired)):
            # a backward pass that raised left reports behind: this step starts from a clean slate
            self.pending = [b[2] for b in self.buckets]
            self.fired = [False] * len(self.buckets)
            self.reported = set()
        self.flat_g.zero_()
        base = self.flat_g.data_ptr()
        for p in self.params:        # re-attach the views if something replaced .grad
            off = self.offsets[id(p)]
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self.flat_g[off:off + p.numel()].view_as(p.data)

    # ---- gradient all-reduce overlapped with backward -------------------------------------
    def enable_overlap(self, world_size, bucket_bytes=32 << 20):
        """Bucket the arena (contiguous ranges of ~bucket_bytes) and all-reduce every bucket on
        a communication stream as soon as the backward kernels of all its parameters have been
        enqueued (per-parameter `_msml_ready` callbacks from the in-place gradient path).  xGMI is a
        point-to-point mesh: a few tens of MB per message keep every link busy without
        serialising the tail of the backward behind one huge ring pass."""
        from . import ops
        self.ov_world = world_size
        self.comm = torch.cuda.Stream()
        self.buckets = []          # [start, end, n_params]
        self.bucket_of = {}
        start, count, cur = 0, 0, 0
        for p in self.params:
            off = self.offsets[id(p)]
            end = off + (p.numel() + 3) // 4 * 4
            self.bucket_of[id(p)] = len(self.buckets)
            count += 1
            cur = end
            if (cur - start) * 4 >= bucket_bytes:
                self.buckets.append([start, cur, count])
                start, count = cur, 0
        if count:
            self.buckets.append([start, cur, count])
        self.pending = [b[2] for b in self.buckets]
        self.fired = [False] * len(self.buckets)
        self.reported = set()
        self.duplicate_reports = 0
        self.works = []
        self._half = []
        self.train_stream = torch.cuda.current_stream()     # refreshed by zero_grad()
        for p in self.params:
            p._msml_ready = self._grad_ready

    def _fire(self, bi):
        from . import ops
        self.fired[bi] = True
        s, e, _ = self.buckets[bi]
        # A bucket may be fired from an autograd node that runs on a side stream (the OSB nodes run
        # last, on the OSB stream): current_stream() is then NOT the stream the FRB's in-place
        # gradient kernels were enqueued on.  Always wait for the training stream remembered by
        # zero_grad(), the current stream and both side streams.
        cur = torch.cuda.current_stream()
        self.comm.wait_stream(cur)
        for st in (getattr(self, "train_stream", None), ops.WGRAD_STREAM, ops.OSB_STREAM):
            if st is not None and st != cur:
                self.comm.wait_stream(st)
        with torch.cuda.stream(self.comm):
            self._reduce(self.flat_g[s:e], async_op=True)

    def _reduce(self, buf, async_op=False):
        """SUM all-reduce of one slice of the gradient arena.  comm_dtype = bfloat16 (MSML_GRAD_COMM=bf16 or
        `opt.comm_dtype`): the message travels as bf16 -- half the xGMI bytes for one conversion pass each way;
        the averaged gradient then carries bf16 rounding."""
        if self.comm_dtype is None:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
            if async_op:
                self.works.append(w)
            return
        half = buf.to(self.comm_dtype)
        w = dist.all_reduce(half, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            self.works.append(w)
            self._half.append((buf, half))
        else:
            buf.copy_(half)

    def _grad_ready(self, p):
        bi = self.bucket_of.get(id(p))
        if bi is None or self.fired[bi]:
            return
        # one report per parameter and step: a parameter whose module ran twice in the forward would report after its FIRST
        # gradient kernel -- count it (tests assert zero) and never let it release the bucket early
        if id(p) in self.reported:
            self.duplicate_reports += 1
            self.pending[bi] = max(self.pending[bi], 1) + (1 << 20)      # this bucket waits for all_reduce_grads()
            return
        self.reported.add(id(p))
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._fire(bi)

    def all_reduce_grads(self, world_size, bucket_bytes=64 << 20):
        """DDP gradient reduction on the flat arena: a few large SUM all-reduces (overlapped with
        the backward when enable_overlap() was called, otherwise issued here).

        Contract: afterwards flat_g -- and every p.grad, which is a view of it -- holds the SUM over the ranks,
        NOT the average; the division by the world size lives in `grad_scale` (1 / W), which step() folds into the
        coefficient of the fused clip + SGD kernel and zero_grad() resets.  Code that reads .grad between this call
        and step() (logging, an external clip_grad_norm_, a torch optimizer after release()) must use
        averaged_grad() / release(), which apply the scale."""
        from . import ops
        if getattr(self, "buckets", None) is not None:
            for bi in range(len(self.buckets)):          # parameters that never reported
                if not self.fired[bi]:
                    self._fire(bi)
            # Work.wait() orders the CURRENT stream behind the collective (ProcessGroupNCCL runs it on an internal
            # stream): wait on the communication stream, where the bf16 -> f32 copy-back is enqueued, and let the
            # training stream wait for that stream afterwards (waiting on the main stream only let the
            # copy-back read `half` before its all-reduce had finished)
            with torch.cuda.stream(self.comm):
                for w in self.works:
                    w.wait()
                for buf, half in self._half:          # bf16 messages: back into the f32 arena
                    buf.copy_(half)
            self._half = []
            torch.cuda.current_stream().wait_stream(self.comm)
            ops.wgrad_stream_join()
            self.grad_scale = 1.0 / self.ov_world     # applied by step() through the clip coefficient
            self.pending = [b[2] for b in self.buckets]
            self.fired = [False] * len(self.buckets)
            self.reported = set()
            self.works = []
            return
        ops.wgrad_stream_join()
        if world_size == 1 and not FORCE_COLLECTIVES:
            return
        n = self.flat_g.numel()
        step = bucket_bytes // 4
        for s in range(0, n, step):
            self._reduce(self.flat_g[s:min(n, s + step)])
        self.grad_scale = 1.0 / world_size            # applied by step() through the clip coefficient

    def step(self):
        from . import ops
        ops.wgrad_stream_join()       # weight gradients may still be in flight on the side stream
        n = self.flat_g.numel()
        clip = None
        if self.max_norm is not None:
            call("msml_grad_norm_clip_scaled", self.flat_g, n, float(self.max_norm), float(self.grad_scale),
                 self.norm_coef, self.ws, self.ws.numel())
            clip = self.norm_coef[1:]
        elif self.grad_scale != 1.0:
            if getattr(self, "_scale_set", None) != self.grad_scale:       # constant for the job: filled once
                self.norm_coef[1] = self.grad_scale
                self._scale_set = self.grad_scale
            clip = self.norm_coef[1:]
        # (a zero momentum buffer makes mu * buf + g the first step of torch.optim.SGD: no first-step flag, nothing
        # about the step count is baked into a captured graph)
        for i, g in enumerate(self.groups):
            s, e = g["start"], g["end"]
            call("msml_sgd_momentum_dev", self.flat_w[s:e], self.flat_g[s:e], self.flat_m[s:e], e - s,
                 self.lr_dev[i:i + 1], float(self.momentum), float(self.weight_decay), clip)
        self.steps += 1
        from . import ops
        ops.WEIGHT_EPOCH += 1        # parameters changed behind torch's version counters

    def grad_norm(self):
        """L2 norm of the (rank-averaged) gradient as seen by the last step()."""
        return self.norm_coef[0]
