"""Autograd functions over the C ABI.  Every tensor that flows between them is an NHWC storage
tensor [N, H, W, Cp] (f32 or bf16); parameters stay f32 in the reference's layouts, so
state dicts and optimizers see exactly what they see with the reference modules.

Backward functions run on PyTorch's autograd thread; the library is reentrant and every call
passes torch's current stream explicitly.
"""
import os
import torch

from . import _lib, ops
from ._lib import BF16, DTYPE_OF, F32, TORCH_DTYPE, call
from .ops import cpad


class _ToNHWC(torch.autograd.Function):
    """MSML.forward boundary: NCHW f32 image -> NHWC storage (backbones/msml.py:150)."""

    @staticmethod
    def forward(ctx, x, dtype):
        return ops.to_nhwc(x.float(), dtype)

    @staticmethod
    def backward(ctx, g):
        return None, None


def to_nhwc(x, dtype):
    return _ToNHWC.apply(x, dtype)


def _slice_w(w, deconv, off, n):
    if deconv:
        return w[off:off + n]
    return w[:, off:off + n].contiguous()


class _Conv(torch.autograd.Function):
    """Conv2d / ConvTranspose2d (+bias) on up to two channel-concatenated inputs, with optional
    per-channel (sum, sumsq) partials of the output for a following training-mode BatchNorm."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, cfg, wp):
        deconv = cfg["deconv"]
        c0, c1 = cfg["c0"], cfg["c1"]
        cout = cfg["cout"]
        r, s = weight.shape[2], weight.shape[3]
        dtype = DTYPE_OF[x0.dtype]
        if wp is None:
            wp = ops.PACKS.get(weight, deconv, 0, weight.shape[0], 0, weight.shape[1], c0, c1, dtype)
        bp = None
        if bias is not None:
            bp = ops.padded_bias(bias, cpad(cout))
        y, stats = ops.conv2d(x0, x1, wp, bp, cpad(cout), r, s, cfg["stride"], cfg["pad_h"],
                              cfg["pad_w"], deconv, out_dtype=cfg.get("out_dtype"),
                              want_stats=cfg.get("want_stats", False), real=(c0 + c1, cout))
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)       # no zeros tensor for the statistics output's "gradient"
        ctx.has_bias = bias is not None
        ctx.wparam = cfg.get("grad_param", weight)      # the leaf whose .grad receives dW
        ctx.wobj = weight                               # the parameter object (pack-cache key)
        ctx.bparam = bias
        ctx.save_for_backward(x0, x1, weight)
        if stats is None:
            stats = x0.new_empty(0)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        return _Conv._bwd(ctx, dy, ctx.needs_input_grad)

    @staticmethod
    def _bwd(ctx, dy, need):
        """need: needs_input_grad as the caller wants it honoured (_ConvTee computes the first input's gradient itself)."""
        x0, x1, weight = ctx.saved_tensors
        cfg = ctx.cfg
        if dy is None:
            return None, None, None, None, None, None
        deconv, c0, c1, cout = cfg["deconv"], cfg["c0"], cfg["c1"], cfg["cout"]
        stride, ph, pw = cfg["stride"], cfg["pad_h"], cfg["pad_w"]
        r, s = weight.shape[2], weight.shape[3]
        dy = dy.contiguous()
        if DTYPE_OF[dy.dtype] != DTYPE_OF[x0.dtype]:
            dy = dy.to(x0.dtype)
        dtype = DTYPE_OF[x0.dtype]
        w = weight.detach()
        n, h, wd, _ = x0.shape
        grads = [None, None]
        if (deconv and x1 is not None and need[0] and need[1] and dtype == BF16
                and r == 4 and s == 4 and stride == 2 and ph == 1 and pw == 1 and x0.shape[3] == 32 and x1.shape[3] == 32
                and dy.shape[3] == 32 and h == wd and h in (14, 28, 56)):
            # OSB decoder: both segments' input gradients from ONE pass over dY (msml_deconv4_bwd_data)
            wp0 = ops.PACKS.get(w, False, 0, c0, 0, cout, cout, 0, dtype, owner=ctx.wobj)
            wp1 = ops.PACKS.get(w, False, c0, c1, 0, cout, cout, 0, dtype, owner=ctx.wobj)
            g0, g1 = torch.empty_like(x0), torch.empty_like(x1)
            name = "conv_igemm"
            if ops.PROFILE.on:
                name = "conv N c32+0->32+32 %dx%d k4x4 s2 n%d [k_deconv4_bwd<both segments>]" % (2 * h, 2 * h, n)
            with ops.PROFILE.rec(name, 2.0 * n * h * wd * (c0 + c1) * cout * 16):
                rc = _lib.try_call("msml_deconv4_bwd_data", dy, wp0, wp1, g0, g1, n, h)
            if rc == 0:
                grads = [g0, g1]
        for i, (x, off, ci) in enumerate(((x0, 0, c0), (x1, c0, c1))):
            if grads[i] is not None:
                continue
            if x is None or not need[i]:
                continue
            if deconv:      # backward-data of a transposed conv = strided conv of dy
                wp = ops.PACKS.get(w, False, off, ci, 0, cout, cout, 0, dtype, owner=ctx.wobj)
                grads[i], _ = ops.conv2d(dy, None, wp, None, cpad(ci), r, s, stride, ph, pw, False,
                                         p=h, q=wd, real=(cout, ci))
            else:           # backward-data of a conv = transposed gather of dy
                wp = ops.PACKS.get(w, True, 0, cout, off, ci, cout, 0, dtype, owner=ctx.wobj)
                grads[i], _ = ops.conv2d(dy, None, wp, None, cpad(ci), r, s, stride, ph, pw, True,
                                         p=h, q=wd, real=(cout, ci))
        dw = None
        if need[2]:
            inplace = ops.inplace(ctx.wparam)
            dw = ctx.wparam.grad.view(w.shape) if inplace else torch.empty_like(w)
            side = ops.WGRAD_STREAM if inplace else None
            if side is not None:                       # dW has no consumer before the optimizer
                _lib.stream_wait_current(side)
                dy.record_stream(side)
                for xs in (x0, x1):                    # saved activations are freed after this node
                    if xs is not None:
                        xs.record_stream(side)
            for x, off, ci in ((x0, 0, c0), (x1, c0, c1)):
                if x is None:
                    continue
                if deconv:
                    ops.conv_wgrad(x, dy, dw[off:off + ci], ci, cout, cout, 0, r, s, stride, ph, pw,
                                   accumulate=inplace, stream=side)
                else:
                    ops.conv_wgrad(dy, x, dw, cout, ci, c0 + c1, off, r, s, stride, ph, pw,
                                   accumulate=inplace, stream=side)
            if inplace:
                dw = None
                ops.grad_ready(ctx.wparam)
        db = None
        if ctx.has_bias and need[3]:
            m = dy.numel() // dy.shape[-1]
            cp = dy.shape[-1]
            rows = ops.bn_stats_rows(m, cp)
            wsb = ops.workspace(rows * 2 * cp * 4, dy.device)
            binplace = ops.inplace(ctx.bparam)
            db = ctx.bparam.grad if binplace else torch.empty(cout, dtype=torch.float32, device=dy.device)
            call("msml_bias_grad", dy, m, cp, cout, db, int(binplace), wsb, wsb.numel() // 4, dtype)
            if binplace:
                db = None
                ops.grad_ready(ctx.bparam)
        return grads[0], grads[1], dw, db, None, None


def conv(x0, x1, weight, bias, cfg, wp=None):
    y, stats = _Conv.apply(x0, x1, weight, bias, cfg, wp)
    return y, (stats if stats.numel() else None)


# Test instrumentation (tests/test_gpu_module_local.py), default off: a deliberately wrong backward ("skip_tee": _ConvTee
# drops the gradient of its second consumer) that the module-level f64 check must flag.  Never set by product code; bench.py
# asserts it is empty.
FAULT = ""


class _ConvTee(torch.autograd.Function):
    """_Conv whose first input has a SECOND consumer: returns (y, statistics, alias of x0).  The caller hands the alias
    to the other consumer; its gradient then arrives HERE and is summed with the conv's input gradient inside the
    backward-data kernel's epilogue (msml_conv2d_fused with the other gradient as residual) instead of by a separate
    element-wise add that autograd would issue for the fan-out (FMCnn: yf feeds same_conv AND the act / arith / skip
    kernel, backbones/fm/fmoperator.py:284-306 of the reference)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, cfg, wp):
        y, stats = _Conv.forward(ctx, x0, x1, weight, bias, cfg, wp)
        ctx.mark_non_differentiable(stats)
        return y, stats, x0.view_as(x0)

    @staticmethod
    def backward(ctx, dy, _dstats, dtee):
        cfg = ctx.cfg
        x0, x1, weight = ctx.saved_tensors
        if FAULT == "skip_tee":      # test instrumentation (tests/test_gpu_module_local.py): the second gradient is dropped
            dtee = None
        fused = None
        if (dy is not None and dtee is not None and ctx.needs_input_grad[0] and not cfg["deconv"] and cfg["stride"] == 1
                and dy.dtype == torch.bfloat16 and dtee.dtype == torch.bfloat16 and dtee.shape == x0.shape):
            c0, cout = cfg["c0"], cfg["cout"]
            r, s = weight.shape[2], weight.shape[3]
            dyc = dy.contiguous()
            wp = ops.PACKS.get(weight.detach(), True, 0, cout, 0, c0, cout, 0, BF16, owner=ctx.wobj)
            n, h, wd, c0p = x0.shape
            unit = _unit_coef(c0p, dy.device)
            fused = torch.empty_like(x0)
            name = "conv_igemm"
            if ops.PROFILE.on:
                name = ops.conv_label("T+add", dyc.shape[3], 0, c0p, n, dyc.shape[1], dyc.shape[2], h, wd, r, s, 1,
                                      cfg["pad_h"], cfg["pad_w"], 1, BF16, BF16, False)
            with ops.PROFILE.rec(name, 2.0 * n * h * wd * cout * c0 * r * s):
                call("msml_conv2d_fused", dyc, dyc.shape[3], None, 0, wp, wp.shape[0], unit[0], unit[1], None,
                     dtee.contiguous(), 0, fused, c0p, n, dyc.shape[1], dyc.shape[2], h, wd, r, s, 1, cfg["pad_h"],
                     cfg["pad_w"], 1)
        if fused is not None:
            # the remaining gradients (second segment, dW, bias) as _Conv computes them, without the first input's
            g = _Conv._bwd(ctx, dy, (False,) + tuple(ctx.needs_input_grad[1:]))
            return (fused,) + tuple(g[1:])
        g = _Conv._bwd(ctx, dy, ctx.needs_input_grad)
        if dtee is not None and g[0] is not None:
            g0 = torch.empty_like(g[0])
            call("msml_add", g[0], dtee.contiguous(), g0, g0.numel(), DTYPE_OF[g0.dtype])
            return (g0,) + tuple(g[1:])
        return ((dtee if g[0] is None else g[0]),) + tuple(g[1:])


def _unit_coef(cp, device):
    """(ones, zeros) f32 [cp]: unit scale / zero shift of a fused conv epilogue that only adds its residual
    (ops.unit_coef: one cache for every stream, filled host-synchronously)."""
    return ops.unit_coef(cp, device)


def conv_tee(x0, x1, weight, bias, cfg, wp=None):
    """conv() that also returns an alias of x0 for x0's other consumer (see _ConvTee)."""
    y, stats, x0b = _ConvTee.apply(x0, x1, weight, bias, cfg, wp)
    return y, (stats if stats.numel() else None), x0b


class _BnAct(torch.autograd.Function):
    """y = prelu(batch_norm(x)) + residual (PReLU and residual optional)."""

    @staticmethod
    def forward(ctx, x, stats, gamma, beta, alpha, residual, rmean, rvar, training, momentum, eps,
                res_first, emit=False):
        # emit (accumulator mode only): also return the (sum, sumsq) accumulator of the OUTPUT -- the statistics the next
        # BatchNorm needs (the stem's output feeds layer1's first bn1: no separate statistics pass over 64 x 112 x 112)
        c = x.shape[-1]
        m = x.numel() // c
        dtype = DTYPE_OF[x.dtype]
        dev = x.device
        coef = torch.empty(4, c, dtype=torch.float32, device=dev)   # scale, shift, mean, invstd
        if training and stats is None and ops.acc_applies(c, dtype):
            stats = ops.stats_acc(c, dev)
            with ops.PROFILE.rec("bn_stats", 0.0, x.numel() * x.element_size()):
                call("msml_bn_stats_acc", x, m, c, stats, dtype)
        if training and stats is not None and stats.dtype == torch.float64:
            # accumulator-mode statistics (ops.ACC_STATS): finalize + apply in one launch
            y = torch.empty_like(x)
            yacc = ops.stats_acc(c, dev) if emit else None
            with ops.PROFILE.rec("bn_act_fwd", 0.0, x.numel() * x.element_size() * (3 if residual is not None else 2)):
                call("msml_bn_fin_act_fwd", stats, float(m), gamma, beta, rmean, rvar, momentum, eps, coef[0], coef[1],
                     coef[2], coef[3], x, alpha, residual, int(res_first), y, m, c, yacc, dtype)
            ctx.training = training
            ctx.has = (gamma is not None, beta is not None, alpha is not None, residual is not None)
            ctx.res_first = bool(res_first) and residual is not None and alpha is not None
            ctx.params = (gamma, beta, alpha)
            ctx.save_for_backward(x, coef, alpha, residual if ctx.res_first else None)
            if emit:
                ctx.mark_non_differentiable(yacc, coef)
                return y, yacc, coef
            return y
        if training:
            if stats is None:
                rows = ops.bn_stats_rows(m, c)
                stats = torch.empty(rows, 2, c, dtype=torch.float32, device=dev)
                with ops.PROFILE.rec("bn_stats", 0.0, x.numel() * x.element_size()):
                    call("msml_bn_stats", x, m, c, stats, dtype)
            call("msml_bn_finalize", stats, stats.shape[0], c, float(m), gamma, beta, rmean, rvar,
                 momentum, eps, coef[0], coef[1], coef[2], coef[3])
        else:
            call("msml_bn_finalize", None, 0, c, 0.0, gamma, beta, rmean, rvar, momentum, eps,
                 coef[0], coef[1], coef[2], coef[3])
        y = torch.empty_like(x)
        with ops.PROFILE.rec("bn_act_fwd", 0.0, x.numel() * x.element_size() * (3 if residual is not None else 2)):
            call("msml_bn_act_fwd", x, coef[0], coef[1], alpha, residual, int(res_first), y, m, c, dtype)
        ctx.training = training
        ctx.has = (gamma is not None, beta is not None, alpha is not None, residual is not None)
        ctx.res_first = bool(res_first) and residual is not None and alpha is not None
        ctx.params = (gamma, beta, alpha)
        ctx.save_for_backward(x, coef, alpha, residual if ctx.res_first else None)
        return y

    @staticmethod
    def backward(ctx, dy, *_dacc):
        x, coef, alpha, res = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("msml_amd: backward through eval-mode BatchNorm is not supported")
        c = x.shape[-1]
        m = x.numel() // c
        dtype = DTYPE_OF[x.dtype]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        pg = torch.empty(3, c, dtype=torch.float32, device=x.device)
        rows = ops.bn_stats_rows(m, c)
        need = (rows * 3 * c + 2 * c) * 4
        ws = ops.workspace(need, x.device)
        dres = torch.empty_like(x) if ctx.res_first else None
        has_g, has_b, has_a, has_r = ctx.has
        gamma, beta, alpha_p = ctx.params
        want = (has_g and ctx.needs_input_grad[2], has_b and ctx.needs_input_grad[3],
                has_a and ctx.needs_input_grad[4])
        # parameter gradients go straight into the flat arena when FlatSGD owns the .grad views
        inplace = all((not w) or ops.inplace(prm) for w, prm in zip(want, (gamma, beta, alpha_p)))
        if inplace:
            tg = [prm.grad if w else None for w, prm in zip(want, (gamma, beta, alpha_p))]
        else:
            tg = [pg[0], pg[1], pg[2] if alpha is not None else None]
        # the three sums may already sit in an accumulator: the first IBasicBlock's bn1 apply kernel, which WROTE this dy,
        # reduced them on the way (blocks._bn_bwd with nxt = (x, coef, alpha); msml_bn_fin_bwd_apply_next_act).  The
        # attribute travels on the gradient tensor object, as blocks' `_msml_bn3_partial` does between chained blocks.
        part = dy.__dict__.pop("_msml_bn3_partial", None)
        if part is not None and part.dtype == torch.float64 and not ctx.res_first and ops.acc_applies(c, dtype):
            ops.COUNTERS["bn3_partial_hits"] += 1
            with ops.PROFILE.rec("bn_act_bwd_apply", 0.0, x.numel() * x.element_size() * 3):
                call("msml_bn_fin_bwd_apply", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], part, None, None, 0, 0, dx,
                     None, tg[0], tg[1], tg[2], int(inplace), m, c, None, None, None, None, dtype)
            if inplace:
                ops.grad_ready(*[prm for w, prm in zip(want, (gamma, beta, alpha_p)) if w])
            return (dx, None,
                    pg[0] if want[0] and not inplace else None,
                    pg[1] if want[1] and not inplace else None,
                    pg[2] if want[2] and not inplace else None,
                    dy if has_r else None, None, None, None, None, None, None, None)
        with ops.PROFILE.rec("bn_act_bwd", 0.0, x.numel() * x.element_size() * (5 + (3 if ctx.res_first else 0))):
            if ops.acc_applies(c, dtype):
                call("msml_bn_act_bwd_acc", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], res, None, dx, dres,
                     tg[0], tg[1], tg[2], int(inplace), m, c, ops.stats_acc(c, x.device, 3), dtype)
            else:
                call("msml_bn_act_bwd", dy, x, coef[0], coef[1], alpha, coef[2], coef[3], res, dx, dres,
                     tg[0], tg[1], tg[2], int(inplace), m, c, ws, ws.numel() // 4, dtype)
        if ctx.res_first:
            dy = dres
        if inplace:
            ops.grad_ready(*[prm for w, prm in zip(want, (gamma, beta, alpha_p)) if w])
        return (dx, None,
                pg[0] if want[0] and not inplace else None,
                pg[1] if want[1] and not inplace else None,
                pg[2] if want[2] and not inplace else None,
                dy if has_r else None, None, None, None, None, None, None, None)


def bn_act(x, stats, bn, prelu=None, residual=None, res_first=False, emit_stats=False):
    """Apply an nn.BatchNorm module `bn` (+ optional nn.PReLU, + residual) to an NHWC tensor.
    res_first: prelu(bn(x) + residual) instead of prelu(bn(x)) + residual.
    emit_stats: (training, accumulator-mode statistics) attach the output's (sum, sumsq) accumulator to the result as
    `_msml_stats`, where a following one-node IBasicBlock picks it up for its bn1 (blocks.iblock)."""
    if isinstance(x, SplitT):
        return bn_act_x3(x, bn, prelu, residual, res_first)
    training = bn.training
    if training:
        ops.bn_counter(bn)
    c = x.shape[-1]
    emit = bool(emit_stats and training and ops.EMIT_STEM_STATS and x.dtype == torch.bfloat16 and ops.acc_applies(c, BF16)
                and (stats is None or stats.dtype == torch.float64) and 256 % (c // 8) == 0)
    out = _BnAct.apply(x, stats, bn.weight, bn.bias, prelu.weight if prelu is not None else None,
                       residual, bn.running_mean, bn.running_var, training,
                       0.1 if bn.momentum is None else bn.momentum, bn.eps, res_first, emit)
    if emit:
        y, yacc, coef = out
        y._msml_stats = yacc
        # (for the one-node IBasicBlock that follows -- blocks.iblock: its bn1 apply kernel writes THIS BatchNorm's output
        # gradient and can reduce this BatchNorm's backward sums on the way, PReLU mask included)
        if residual is None and ops.STEM_BWD_SUMS:
            y._msml_bn3 = (x, coef, prelu.weight if prelu is not None else None)
        return y
    return out


class _FmFuse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, yf, act, arith):
        z = torch.empty_like(yf)
        with ops.PROFILE.rec("fm_fuse_fwd", 0.0, 3 * x.numel() * x.element_size()):
            call("msml_fm_fuse_fwd", x, yf, z, x.numel(), act, arith, DTYPE_OF[x.dtype])
        ctx.save_for_backward(x, yf)
        ctx.mode = (act, arith)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, yf = ctx.saved_tensors
        dz = dz.contiguous()
        dx, dyf = torch.empty_like(x), torch.empty_like(yf)
        with ops.PROFILE.rec("fm_fuse_bwd", 0.0, 5 * x.numel() * x.element_size()):
            call("msml_fm_fuse_bwd", dz, x, yf, dx, dyf, x.numel(), ctx.mode[0], ctx.mode[1],
                 DTYPE_OF[x.dtype])
        return dx, dyf, None, None


ACTS = {"tanh": 0, "sigmoid": 1}
ARITHS = {"add": 0, "sub": 1, "mul": 2, "div": 3}


def fm_fuse(x, yf, act, arith):
    if isinstance(x, SplitT):
        return fm_fuse_x3(x, yf, act, arith)
    return _FmFuse.apply(x, yf, ACTS[act], ARITHS[arith])


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        call("msml_add", a, b, out, a.numel(), DTYPE_OF[a.dtype])
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    if isinstance(a, SplitT):
        return add_x3(a, b)
    return _Add.apply(a, b)


class _FanOut2(torch.autograd.Function):
    """A tensor with two consumers (the OSB's encoder outputs feed the next stage AND a GCM, backbones/osb/unet.py:205-226;
    a GCM input feeds both of its branches, :29-38): the two gradients are summed by msml_add on the backward's stream
    instead of by the autograd engine's own element-wise add."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        ga, gb = ga.contiguous(), gb.contiguous()
        out = torch.empty_like(ga)
        call("msml_add", ga, gb, out, out.numel(), DTYPE_OF[ga.dtype])
        return out


def fanout2(x):
    """(x, x) for a tensor that two branches consume; with autograd on, their gradients meet in msml_add."""
    if (torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad and x.is_cuda
            and x.dtype in DTYPE_OF):
        a, b = _FanOut2.apply(x)
        # what a producer attached for a following IBasicBlock (the first branch is the one that continues the backbone):
        # the forward statistics of x.  NOT `_msml_bn3` (round 6): the block behind a fan-out sees only ITS share of x's
        # gradient, so the sums it would reduce for x's BatchNorm while writing that share are not that BatchNorm's sums
        # -- they were dropped at the msml_add above anyway, after costing the block's apply kernel another input stream
        if "_msml_stats" in x.__dict__:
            a.__dict__["_msml_stats"] = x.__dict__["_msml_stats"]
        return a, b
    return x, x


class _Dap(torch.autograd.Function):
    """NHWC 18-channel map -> final_seg NCHW f32 (N, 2, H, W)."""

    @staticmethod
    def forward(ctx, x):
        n, h, w, cp = x.shape
        seg = torch.empty(n, 2, h, w, dtype=torch.float32, device=x.device)
        call("msml_dap_fwd", x, seg, None, n, h, w, cp, DTYPE_OF[x.dtype])
        ctx.meta = (x.shape, x.dtype)
        return seg

    @staticmethod
    def backward(ctx, dseg):
        shape, dt = ctx.meta
        dx = torch.empty(shape, dtype=dt, device=dseg.device)
        dseg = dseg.contiguous().float()
        # boundary of the OSB graph: this node runs on the OSB stream, its incoming gradient was
        # produced (and is freed) by the loss on the main stream -- keep the allocator from handing
        # the block to the FRB backward while this stream still reads it
        dseg.record_stream(_lib.current_stream())
        call("msml_dap_bwd", dseg, dx, shape[0], shape[1], shape[2], shape[3], DTYPE_OF[dt])
        return dx


def dap(x):
    if isinstance(x, SplitT):
        x = x3_to_f32(x)
    return _Dap.apply(x)


def mask_index(final_seg):
    """Occlusion-mask index (train.py:357): uint8 (N, H, W), 0 = occluded, 1 = clean."""
    n, _, h, w = final_seg.shape
    a, b = final_seg[:, 0], final_seg[:, 1]
    return (b > a).to(torch.uint8)


class _ToVec(torch.autograd.Function):
    """NHWC [N,1,1,C] storage -> (N, C) f32 feature (and back for the gradient)."""

    @staticmethod
    def forward(ctx, x, c):
        ctx.meta = (x.shape, x.dtype)
        return x.reshape(x.shape[0], -1)[:, :c].float()

    @staticmethod
    def backward(ctx, g):
        shape, dt = ctx.meta
        out = torch.zeros(shape, dtype=dt, device=g.device)
        out.reshape(shape[0], -1)[:, :g.shape[1]] = g.to(dt)
        return out, None


def to_vec(x, c):
    return _ToVec.apply(x, c)


class _SegLoss(torch.autograd.Function):
    """StructureConsensuLossFunction(alpha, beta, 'idx', 'idx')(logit, msk, msk)."""

    @staticmethod
    def forward(ctx, logit, msk, alpha, beta, mean_all=False, kl_all=False):
        n, c, h, w = logit.shape
        assert c == 2
        logit = logit.contiguous().float()
        msk = msk.to(device=logit.device, dtype=torch.int64).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=logit.device)
        dlogit = torch.empty_like(logit)
        ws = torch.empty(20 * n, dtype=torch.float32, device=logit.device)
        call("msml_seg_consensus_loss_r", logit, msk, n, h, w, alpha, beta, int(mean_all), int(kl_all), loss, dlogit, ws,
             ws.numel())
        ctx.save_for_backward(dlogit)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlogit,) = ctx.saved_tensors
        return dlogit * g, None, None, None, None, None


def seg_consensus_loss(logit, msk, alpha=10.0, beta=5.0, reduce_pixel="idx", reduce_pixel_kl="idx"):
    return _SegLoss.apply(logit, msk, alpha, beta, reduce_pixel == "all", reduce_pixel_kl == "all")


# --------------------------------------------------------------------------- cosine heads
HEAD_KIND = {"arc": 0, "cos": 1}


class _CosMarginHead(torch.autograd.Function):
    """s * margin(normalize(emb) @ normalize(W).T) -- AMArcFace / AMCosFace forward
    (headers/margin_losses.py:371-418 / :257-303) with explicit backward."""

    @staticmethod
    def forward(ctx, emb, weight, label, kind, s, m, a, k, dtype):
        b, e = emb.shape
        c = weight.shape[0]
        dev = emb.device
        tdt = TORCH_DTYPE[dtype]
        cp = cpad(c)
        kop = (cp + ops.tile_n(cp) - 1) // ops.tile_n(cp) * ops.tile_n(cp)
        xn = torch.empty(b, 1, 1, e, dtype=tdt, device=dev)
        inv_x = torch.empty(b, dtype=torch.float32, device=dev)
        call("msml_rownorm_fwd", emb.detach().contiguous(), b, b, e, xn, e, inv_x, dtype)
        wn = torch.empty(kop, e, dtype=tdt, device=dev)
        inv_w = torch.empty(c, dtype=torch.float32, device=dev)
        call("msml_rownorm_fwd", weight.detach(), c, kop, e, wn, e, inv_w, dtype)
        cosm, _ = ops.conv2d(xn, None, wn, None, cp, 1, 1, 1, 0, 0, False, out_dtype=F32)
        cosm = cosm.reshape(b, cp)
        label = label.to(device=dev, dtype=torch.int64).contiguous()
        cos_t = torch.empty(b, dtype=torch.float32, device=dev)
        call("msml_gather_target", cosm, cp, label, b, cos_t)
        call("msml_margin_fwd", cosm, label, b, c, cp, kind, s, m, a, k)
        ctx.save_for_backward(emb, weight, label, xn, wn, inv_x, inv_w, cos_t)
        ctx.prm = (kind, s, m, a, k, dtype)
        return cosm[:, :c]

    @staticmethod
    def backward(ctx, dlogit):
        emb, weight, label, xn, wn, inv_x, inv_w, cos_t = ctx.saved_tensors
        kind, s, m, a, k, dtype = ctx.prm
        b, e = emb.shape
        c = weight.shape[0]
        dev = emb.device
        tdt = TORCH_DTYPE[dtype]
        cp = cpad(c)
        dlogit = dlogit.contiguous().float()
        dcos = torch.empty(b, 1, 1, cp, dtype=tdt, device=dev)
        call("msml_margin_bwd", dlogit, dlogit.shape[1], label, cos_t, b, c, dcos, cp, kind, s, m,
             a, k, dtype)
        demb = dw = None
        if ctx.needs_input_grad[0]:
            # dXn = dcos @ Wn : GEMM with K = classes -> weights packed transposed [E][Cp]
            wnt = torch.empty(e, ops.kpad(cp), dtype=tdt, device=dev)       # K padded to 32
            call("msml_transpose", wn, c, e, e, wnt, ops.kpad(cp), dtype)
            dxn, _ = ops.conv2d(dcos, None, wnt, None, e, 1, 1, 1, 0, 0, False, out_dtype=F32)
            demb = torch.empty_like(emb)
            call("msml_rownorm_bwd", emb.detach().contiguous(), inv_x, dxn.reshape(b, e), e, b, e,
                 demb, 0)
        if ctx.needs_input_grad[1]:
            dwn = torch.empty(c, e, dtype=torch.float32, device=dev)
            ops.conv_wgrad(dcos, xn, dwn, c, e, e, 0, 1, 1, 1, 0, 0)
            dw = torch.empty_like(weight)
            call("msml_rownorm_bwd", weight.detach(), inv_w, dwn, e, c, e, dw, 0)
        return demb, dw, None, None, None, None, None, None, None


def cos_margin_head(emb, weight, label, kind, s, m, a, k, dtype):
    return _CosMarginHead.apply(emb, weight, label, HEAD_KIND[kind], s, m, a, k, dtype)


class _Linear(torch.autograd.Function):
    """Plain F.linear(emb, W, b) (Softmax head, headers/margin_losses.py:53) on the MFMA GEMM."""

    @staticmethod
    def forward(ctx, emb, weight, bias, dtype):
        b, e = emb.shape
        c = weight.shape[0]
        tdt = TORCH_DTYPE[dtype]
        x = emb.detach().to(tdt).reshape(b, 1, 1, e).contiguous()
        wp = ops.pack_weight(weight.detach().reshape(c, e, 1, 1), False, e, 0, dtype)
        bp = torch.zeros(cpad(c), dtype=torch.float32, device=emb.device)
        bp[:c] = bias.detach()
        y, _ = ops.conv2d(x, None, wp, bp, cpad(c), 1, 1, 1, 0, 0, False, out_dtype=F32)
        ctx.save_for_backward(x, weight)
        ctx.dtype = dtype
        return y.reshape(b, -1)[:, :c]

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dtype = ctx.dtype
        tdt = TORCH_DTYPE[dtype]
        b, e = x.shape[0], x.shape[3]
        c = weight.shape[0]
        cp = cpad(c)
        g = torch.zeros(b, 1, 1, cp, dtype=tdt, device=dy.device)
        g.reshape(b, cp)[:, :c] = dy.to(tdt)
        wt = ops.pack_weight(weight.detach().reshape(c, e, 1, 1), True, c, 0, dtype)
        dx, _ = ops.conv2d(g, None, wt, None, e, 1, 1, 1, 0, 0, False, out_dtype=F32)
        dw = torch.empty_like(weight)
        ops.conv_wgrad(g, x, dw, c, e, e, 0, 1, 1, 1, 0, 0)
        db = dy.float().sum(0)
        return dx.reshape(b, e), dw, db, None


def linear(emb, weight, bias, dtype):
    return _Linear.apply(emb, weight, bias, dtype)


class _Normalize(torch.autograd.Function):
    """F.normalize(x, dim=1) on (B, E) f32 embeddings (rownorm kernels)."""

    @staticmethod
    def forward(ctx, x):
        b, e = x.shape
        x = x.contiguous().float()
        y = torch.empty_like(x)
        inv = torch.empty(b, dtype=torch.float32, device=x.device)
        call("msml_rownorm_fwd", x, b, b, e, y, e, inv, F32)
        ctx.save_for_backward(x, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, inv = ctx.saved_tensors
        b, e = x.shape
        dx = torch.empty_like(x)
        call("msml_rownorm_bwd", x, inv, dy.contiguous().float(), e, b, e, dx, 0)
        return dx


def normalize(x):
    return _Normalize.apply(x)


class _FlatFc(torch.autograd.Function):
    """flatten(C,H,W) + Linear(C*H*W -> E) on an NHWC map (backbones/frb/iresnet.py:230-232).

    The packed operand of the equivalent HxW 'valid' window conv is exactly the Linear weight with
    its columns in NHWC flatten order, so forward is a skinny GEMM (M = batch, K = 25 088) run
    with split-K, backward-data a plain GEMM against the transposed operand (the window-conv
    formulation would visit 49 taps per output pixel with one valid), and the weight gradient the
    window wgrad that writes straight into the parameter's (E, C, H, W) layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, grad_param):
        n, h, w, c = x.shape
        e = weight.shape[0]
        dtype = DTYPE_OF[x.dtype]
        wp = ops.PACKS.get(weight, False, 0, e, 0, c, c, 0, dtype, owner=grad_param)      # [E][H*W*C]
        xf = x.reshape(n, h * w * c)
        if dtype == BF16:
            y = ops.gemm_splitk(xf, wp, cpad(e))
        else:
            y, _ = ops.conv2d(x, None, wp, None, cpad(e), h, w, 1, 0, 0, False, out_dtype=F32)
            y = y.reshape(n, cpad(e))
        y = y[:, :e] + bias.detach()
        ctx.save_for_backward(x, weight, bias)
        ctx.grad_param = grad_param
        ctx.wp = wp
        # The embedding tail stays f32 in every mode: the GEMM accumulates in f32 anyway, and BatchNorm1d
        # (`features`, iresnet.py:232) subtracts the batch mean -- a bf16 rounding of its INPUT is amplified
        # by |y| / |y - mean| (measured: 2.2 x the emulated bf16 floor on the fc gradient at batch 4).
        return y.contiguous().reshape(n, 1, 1, e)

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        n, h, w, c = x.shape
        e = weight.shape[0]
        dtype = DTYPE_OF[x.dtype]
        dy = dy.contiguous().to(x.dtype)         # the MFMA operand of the two backward GEMMs (f32 tail -> bf16)
        k = h * w * c
        dx = None
        if ctx.needs_input_grad[0]:
            if dtype == BF16:
                wpt = torch.empty(k, ops.kpad(e), dtype=x.dtype, device=x.device)   # [H*W*C][E]
                call("msml_transpose", ctx.wp, e, k, k, wpt, ops.kpad(e), dtype)
                dxf, _ = ops.conv2d(dy, None, wpt, None, k, 1, 1, 1, 0, 0, False, real=(e, k))
                dx = dxf.reshape(n, h, w, c)
            else:
                wt = ops.PACKS.get(weight, True, 0, e, 0, c, e, 0, dtype, owner=ctx.grad_param)
                dx, _ = ops.conv2d(dy, None, wt, None, c, h, w, 1, 0, 0, True, p=h, q=w, real=(e, c))
        gp = ctx.grad_param
        inplace = ops.inplace(gp)
        dw = gp.grad.view(weight.shape) if inplace else torch.empty_like(weight)
        ops.conv_wgrad(dy, x, dw, e, c, c, 0, h, w, 1, 0, 0, accumulate=inplace)
        binplace = ops.inplace(bias)
        rows = ops.bn_stats_rows(n, e)
        wsb = ops.workspace(rows * 2 * e * 4, dy.device)
        db = bias.grad if binplace else torch.empty(e, dtype=torch.float32, device=dy.device)
        call("msml_bias_grad", dy, n, e, e, db, int(binplace), wsb, wsb.numel() // 4, dtype)
        if binplace:
            ops.grad_ready(bias)
            db = None
        if inplace:
            ops.grad_ready(gp)
            dw = None
        return dx, dw, db, None


def flat_fc(x, weight4d, bias, grad_param):
    """x: NHWC [N,H,W,C]; weight4d: the Linear weight viewed as (E, C, H, W)."""
    return _FlatFc.apply(x, weight4d, bias, grad_param)


def _eval_bn_coef(bn_m, c):
    """(scale, shift) of an eval-mode BatchNorm, cached on the module until a parameter, a running
    statistic or the flat weight arena changes."""
    stamp = (bn_m.weight._version, bn_m.bias._version, bn_m.running_mean._version,
             bn_m.running_var._version, ops.WEIGHT_EPOCH, bn_m.weight.data_ptr())
    hit = getattr(bn_m, "_msml_eval_coef", None)
    if hit is None or hit[0] != stamp:
        coef = torch.empty(2, c, dtype=torch.float32, device=bn_m.weight.device)
        call("msml_bn_finalize", None, 0, c, 0.0, bn_m.weight, bn_m.bias, bn_m.running_mean,
             bn_m.running_var, 0.1, bn_m.eps, coef[0], coef[1], None, None)
        hit = (stamp, coef)
        bn_m._msml_eval_coef = hit
    return hit[1]


def conv_bn_eval(x0, x1, conv_m, bn_m, prelu, residual, c1, res_first):
    """Inference: conv with eval-mode BatchNorm (+PReLU, +residual) folded into its epilogue
    (msml_conv2d_fused).  bf16 NHWC tensors, no autograd graph."""
    n, h, w, c0p = x0.shape
    cin, cout = conv_m.in_channels, conv_m.out_channels
    c0 = cin - c1
    r, s = conv_m.kernel_size
    stride, (ph, pw) = conv_m.stride[0], conv_m.padding
    coutp = cpad(cout)
    assert coutp == cout == bn_m.num_features
    wp = ops.PACKS.get(conv_m.weight, False, 0, cout, 0, cin, c0, c1, BF16)
    coef = _eval_bn_coef(bn_m, coutp)
    p = ops.conv_out_size(h, r, stride, ph, False)
    q = ops.conv_out_size(w, s, stride, pw, False)
    c1p = 0 if x1 is None else x1.shape[3]
    out = torch.empty(n, p, q, coutp, dtype=torch.bfloat16, device=x0.device)
    name = "conv_fused"
    if ops.PROFILE.on:
        name = ops.conv_label("N+bn", c0p, c1p, coutp, n, h, w, p, q, r, s, stride, ph, pw, 0, BF16, BF16, False)
    with ops.PROFILE.rec(name, 2.0 * n * p * q * cin * cout * r * s):
        call("msml_conv2d_fused", x0, c0p, x1, c1p, wp, wp.shape[0], coef[0], coef[1],
             prelu.weight if prelu is not None else None, residual, int(res_first), out, coutp,
             n, h, w, p, q, r, s, stride, ph, pw, 0)
    return out


# ---------------------------------------------------------------------------------------------
# Stems on the raw image (bf16 mode): im2col once, then a 1x1 conv over 32 channels.
class RawImage:
    """The NCHW f32 input image as handed to MSML.forward, for the stems' im2col path.  The padded
    NHWC tensor the generic conv path needs is only built if somebody asks for it."""

    def __init__(self, x, x3=False):
        self.raw = x
        self.dtype = torch.bfloat16
        self.x3 = x3                # inference in split-bf16 (bf16x3) storage
        self._nhwc = None

    def nhwc(self):
        if self._nhwc is None:
            self._nhwc = to_nhwc(self.raw, BF16)
        return self._nhwc

    def record_stream(self, stream):
        self.raw.record_stream(stream)


def _stem_pack(conv_m):
    """Packed [Cout][R][S][C] (K = 27 -> 32) weight of a stem conv, refreshed when the parameter changes."""
    w = conv_m.weight
    stamp = (w._version, ops.WEIGHT_EPOCH, w.data_ptr())
    hit = getattr(conv_m, "_msml_stem_pack", None)
    if hit is None or hit[0] != stamp:
        cout, cin, r, s = w.shape
        w2 = w.detach().permute(0, 2, 3, 1).reshape(cout, r * s * cin, 1, 1).contiguous()
        hit = (stamp, ops.pack_weight(w2, False, r * s * cin, 0, BF16))
        conv_m._msml_stem_pack = hit
    return hit[1]


class _StemConv(torch.autograd.Function):
    """3x3 conv on the 3-channel image as im2col + 1x1 conv; only the weight gets a gradient."""

    @staticmethod
    def forward(ctx, weight, conv_m, raw):
        cout, cin, r, s = weight.shape
        col = ops.stem_im2col(raw, r, s, conv_m.stride[0], conv_m.padding[0])
        wp = _stem_pack(conv_m)
        y, stats = ops.conv2d(col, None, wp, None, cpad(cout), 1, 1, 1, 0, 0, False, want_stats=True,
                              real=(r * s * cin, cout))
        ctx.set_materialize_grads(False)
        ctx.wparam = weight
        ctx.save_for_backward(col)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return None, None, None
        (col,) = ctx.saved_tensors
        w = ctx.wparam
        cout, cin, r, s = w.shape
        k = r * s * cin
        dw2 = torch.empty(cout, k, 1, 1, dtype=torch.float32, device=dy.device)
        ops.conv_wgrad(dy.contiguous(), col, dw2, cout, k, k, 0, 1, 1, 1, 0, 0)
        dw = dw2.view(cout, r, s, cin).permute(0, 3, 1, 2)          # back to the parameter's OIHW
        if ops.inplace(w):
            w.grad.view(w.shape).add_(dw)
            ops.grad_ready(w)
            return None, None, None
        return dw.contiguous(), None, None


def stem_conv_bn(raw, conv_m, bn_m, prelu):
    """Stem conv -> BatchNorm -> PReLU on a RawImage (training and inference)."""
    if raw.x3:
        return stem_conv_bn_x3(raw, conv_m, bn_m, prelu)
    if not bn_m.training and not torch.is_grad_enabled():
        cout, cin, r, s = conv_m.weight.shape
        col = ops.stem_im2col(raw.raw, r, s, conv_m.stride[0], conv_m.padding[0])
        wp = _stem_pack(conv_m)
        coef = _eval_bn_coef(bn_m, cpad(cout))
        n, p, q, kp = col.shape
        out = torch.empty(n, p, q, cpad(cout), dtype=torch.bfloat16, device=col.device)
        name = "conv_fused"
        if ops.PROFILE.on:
            name = ops.conv_label("N+bn", kp, 0, cpad(cout), n, p, q, p, q, 1, 1, 1, 0, 0, 0, BF16, BF16, False)
        with ops.PROFILE.rec(name, 2.0 * n * p * q * r * s * cin * cout):
            call("msml_conv2d_fused", col, kp, None, 0, wp, wp.shape[0], coef[0], coef[1],
                 prelu.weight if prelu is not None else None, None, 0, out, cpad(cout), n, p, q, p, q, 1, 1,
                 1, 0, 0, 0)
        return out
    y, stats = _StemConv.apply(conv_m.weight, conv_m, raw.raw)
    # (the stem's output is the input of layer1's first bn1: its statistics ride on this BatchNorm's apply pass)
    return bn_act(y, stats if stats.numel() else None, bn_m, prelu, emit_stats=True)


# ---------------------------------------------------------------------------------------------
# Split-bf16 ("bf16x3") inference: f32-class accuracy on the bf16 MFMA (csrc/x3.hip, msml_conv2d_x3).
# Every tensor between the ops is a SplitT; the ops below mirror conv / conv_bn_eval / bn_act /
# fm_fuse / add / dap / flat_fc for that storage.  Inference only (no autograd graph).
class SplitT:
    """[N, H, W, 3*Cp] bf16 = planes [hi | lo | hi] of an f32-valued NHWC tensor with Cp channels."""
    __slots__ = ("t", "c")
    dtype = "bf16x3"

    def __init__(self, t, c):
        self.t, self.c = t, c

    @property
    def shape(self):
        n, h, w, _ = self.t.shape
        return (n, h, w, self.c)

    @property
    def device(self):
        return self.t.device

    def detach(self):
        return self

    def record_stream(self, stream):
        self.t.record_stream(stream)

    def numel(self):
        n, h, w, _ = self.t.shape
        return n * h * w * self.c


def _x3_empty(n, h, w, cp, dev):
    return SplitT(torch.empty(n, h, w, 3 * cp, dtype=torch.bfloat16, device=dev), cp)


def x3_from_f32(x):
    """NHWC f32 [N,H,W,Cp] -> SplitT."""
    n, h, w, cp = x.shape
    out = _x3_empty(n, h, w, cp, x.device)
    call("msml_x3_from_f32", x.contiguous(), out.t, n * h * w, cp)
    return out


def x3_to_f32(s):
    n, h, w, cp = s.shape
    out = torch.empty(n, h, w, cp, dtype=torch.float32, device=s.device)
    call("msml_x3_to_f32", s.t, out, n * h * w, cp)
    return out


def to_nchw_any(t, c):
    """NHWC storage tensor of any precision mode -> NCHW f32 with the first c channels."""
    if isinstance(t, SplitT):
        t = x3_to_f32(t)
    return ops.to_nchw(t, c)


def _x3_expand(w, axis, segs):
    """f32 weight -> the split operand [wh | wh | wl] per input segment along `axis` (the input-channel
    axis), each segment zero-padded to its storage channel count first."""
    parts = []
    off = 0
    for c, cp_ in segs:
        ws = w.narrow(axis, off, c)
        if cp_ != c:
            pad = list(ws.shape)
            pad[axis] = cp_ - c
            ws = torch.cat((ws, ws.new_zeros(pad)), axis)
        wh = ws.to(torch.bfloat16).float()
        parts += [wh, wh, ws - wh]
        off += c
    return torch.cat(parts, axis).contiguous()


def _x3_pack(owner, key, w, transpose, segs):
    """Packed split operand of parameter `w`, cached on the module until the parameter changes."""
    stamp = (w._version, ops.WEIGHT_EPOCH, w.data_ptr(), key)
    cache = owner.__dict__.setdefault("_msml_x3_pack", {})
    hit = cache.get(key)
    if hit is None or hit[0] != stamp:
        wexp = _x3_expand(w.detach().float(), 0 if transpose else 1, segs)
        c1 = 3 * segs[0][1]
        c2 = 3 * segs[1][1] if len(segs) > 1 else 0
        hit = (stamp, ops.pack_weight(wexp, transpose, c1, c2, BF16))
        cache[key] = hit
    return hit[1]


def conv_x3(x0, x1, conv_m, scale, shift, alpha, residual, res_first, c1=0, weight=None, pack_key="fwd"):
    """conv / deconv on split tensors with the affine + PReLU + residual epilogue (msml_conv2d_x3)."""
    import torch.nn as nn
    deconv = isinstance(conv_m, nn.ConvTranspose2d)
    w = conv_m.weight if weight is None else weight
    cin, cout = conv_m.in_channels, conv_m.out_channels
    c0 = cin - c1
    r, s = conv_m.kernel_size
    stride, (ph, pw) = conv_m.stride[0], conv_m.padding
    n, h, wd, c0p = x0.shape
    segs = [(c0, c0p)] + ([(c1, x1.shape[3])] if x1 is not None else [])
    wp = _x3_pack(conv_m, pack_key, w, deconv, segs)
    coutp = cpad(cout)
    p = ops.conv_out_size(h, r, stride, ph, deconv)
    q = ops.conv_out_size(wd, s, stride, pw, deconv)
    out = _x3_empty(n, p, q, coutp, x0.device)
    pix = n * h * wd if deconv else n * p * q
    name = "conv_x3"
    if ops.PROFILE.on and os.environ.get("MSML_PROFILE_X3_SHAPES"):
        name = "conv_x3 c%d+%d->%d %dx%d k%dx%d s%d%s n%d" % (c0p, x1.shape[3] if x1 is not None else 0, coutp, h, wd, r, s,
                                                             stride, "T" if deconv else "", n)
    with ops.PROFILE.rec(name, 2.0 * pix * cin * cout * r * s):
        call("msml_conv2d_x3", x0.t, c0p, x1.t if x1 is not None else None, x1.shape[3] if x1 is not None else 0,
             wp, wp.shape[0], scale, shift, alpha, residual.t if residual is not None else None, int(res_first),
             out.t, coutp, n, h, wd, p, q, r, s, stride, ph, pw, int(deconv))
    return out


def _pad_vec(v, cp):
    """Per-channel f32 vector zero-padded to the storage channel count."""
    if v.numel() == cp:
        return v.detach()
    out = torch.zeros(cp, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v.detach()
    return out


def conv_bn_eval_x3(x0, x1, conv_m, bn_m, prelu, residual, c1, res_first):
    coutp = cpad(conv_m.out_channels)
    coef = _eval_bn_coef(bn_m, coutp)
    return conv_x3(x0, x1, conv_m, coef[0], coef[1], prelu.weight if prelu is not None else None, residual,
                   res_first, c1)


# IBasicBlock's bn1 sits IN FRONT of conv1 (backbones/frb/iresnet.py:58-60): in eval mode it is an affine map per input
# channel, so conv1(bn1(x)) = conv(W s1, x) + sum over the taps inside the map of W t1 -- a constant per BORDER CLASS of the
# output pixel (the zero padding follows the BatchNorm).  The split-bf16 path folds it into conv1's operand and a [9][Cout]
# shift table (msml_conv2d_x3_border) instead of a pass over the block input.  MSML_X3_NO_BN1_FOLD=1 restores the pass.
X3_FOLD_BN1 = os.environ.get("MSML_X3_NO_BN1_FOLD") is None
X3_FOLD_MAX_RATIO = 8.0          # largest |running_mean| / sqrt(running_var + eps) of a channel the fold accepts


def _bn_stamp(bn_m):
    return (bn_m.weight._version, bn_m.bias._version, bn_m.running_mean._version, bn_m.running_var._version,
            bn_m.weight.data_ptr())


def bn_conv_bn_eval_x3(x, bn_in, conv_m, bn_m, prelu):
    """bn_in -> conv3x3 (stride 1, pad 1, no bias) -> bn_m -> PReLU on a split tensor in ONE launch; None when the conv
    is not of that kind (the caller then runs bn_act + conv_bn)."""
    n, h, w, cp = x.shape
    cout, cin = conv_m.out_channels, conv_m.in_channels
    if (not X3_FOLD_BN1 or conv_m.kernel_size != (3, 3) or conv_m.stride != (1, 1) or conv_m.padding != (1, 1)
            or conv_m.bias is not None or h < 2 or w < 2):
        return None
    coutp = cpad(cout)
    cw = conv_m.weight
    stamp = (_bn_stamp(bn_in), _bn_stamp(bn_m), cw._version, cw.data_ptr(), ops.WEIGHT_EPOCH, cp)
    hit = conv_m.__dict__.get("_msml_x3_fold")
    if hit is None or hit[0] != stamp:
        # conv(W s, x) and sum(W t) cancel when a channel's running mean is large against its standard deviation: the
        # split-bf16 rounding of W s x (2^-17 of ITS size) is then |mean| / std times larger relative to the result than in
        # the separate pass, which rounds s x + t after the subtraction.  Fold only while that amplification
        # stays below X3_FOLD_MAX_RATIO (8 x 2^-17 = 6e-5 of the output scale, inside the 2e-4 parity bar); checked once
        # per parameter version, where the operand is rebuilt anyway.
        ratio = float((bn_in.running_mean.detach().abs() / (bn_in.running_var.detach() + bn_in.eps).sqrt()).max())
        if not ratio <= X3_FOLD_MAX_RATIO:
            conv_m.__dict__["_msml_x3_fold"] = (stamp, None, None)
            return None
        c_in = _eval_bn_coef(bn_in, cp).double()
        c_out = _eval_bn_coef(bn_m, coutp).double()
        wd = cw.detach().double()
        wf = (wd * c_in[0, :cin].view(1, cin, 1, 1)).float()
        wp = ops.pack_weight(_x3_expand(wf, 1, [(cin, cp)]), False, 3 * cp, 0, BF16)
        tap = (wd * c_in[1, :cin].view(1, cin, 1, 1)).sum(1)                  # [cout][3][3]: W t1 per tap
        rows = ((1, 2), (0, 1, 2), (0, 1))                                    # taps inside the map: first / inner / last
        b9 = torch.zeros(9, coutp, dtype=torch.float64, device=cw.device)
        for cy in range(3):
            for cx in range(3):
                b9[cy * 3 + cx, :cout] = tap[:, rows[cy], :][:, :, rows[cx]].sum((1, 2))
        shift9 = (b9 * c_out[0].view(1, coutp) + c_out[1].view(1, coutp)).float().contiguous()
        hit = (stamp, wp, shift9)
        conv_m.__dict__["_msml_x3_fold"] = hit
    _, wp, shift9 = hit
    if wp is None:               # (this BatchNorm's mean / std ratio rules the fold out: the caller runs the separate pass)
        return None
    scale = _eval_bn_coef(bn_m, coutp)[0]
    out = _x3_empty(n, h, w, coutp, x.device)
    with ops.PROFILE.rec("conv_x3", 2.0 * n * h * w * cin * cout * 9):
        call("msml_conv2d_x3_border", x.t, cp, wp, wp.shape[0], scale, shift9, prelu.weight if prelu is not None else None,
             None, 0, out.t, coutp, n, h, w)
    return out


def conv_plain_x3(conv_m, x0, x1, c1):
    """Conv / deconv (+ bias) without BatchNorm (OSB decoder, FM same_conv)."""
    shift = _pad_vec(conv_m.bias, cpad(conv_m.out_channels)) if conv_m.bias is not None else None
    return conv_x3(x0, x1, conv_m, None, shift, None, None, 0, c1)


def bn_act_x3(x, bn_m, prelu=None, residual=None, res_first=False):
    n, h, w, cp = x.shape
    coef = _eval_bn_coef(bn_m, cp)
    y = _x3_empty(n, h, w, cp, x.device)
    name = "bn_act_fwd"
    if ops.PROFILE.on and os.environ.get("MSML_PROFILE_X3_SHAPES"):
        name = "bn_act_x3 c%d %dx%d n%d%s%s" % (cp, h, w, n, " +prelu" if prelu is not None else "", " +res" if residual is not None else "")
    with ops.PROFILE.rec(name, 0.0, n * h * w * cp * 2 * (7 if residual is not None else 5)):
        call("msml_x3_bn_act_fwd", x.t, coef[0], coef[1], prelu.weight if prelu is not None else None,
             residual.t if residual is not None else None, int(res_first), y.t, n * h * w, cp)
    return y


def fm_fuse_x3(x, yf, act, arith):
    n, h, w, cp = x.shape
    z = _x3_empty(n, h, w, cp, x.device)
    with ops.PROFILE.rec("fm_fuse_fwd", 0.0, 3 * x.t.numel() * 2):
        call("msml_x3_fm_fuse_fwd", x.t, yf.t, z.t, n * h * w, cp, ACTS[act], ARITHS[arith])
    return z


def add_x3(a, b):
    n, h, w, cp = a.shape
    out = _x3_empty(n, h, w, cp, a.device)
    call("msml_x3_add", a.t, b.t, out.t, n * h * w, cp)
    return out


def stem_conv_bn_x3(raw, conv_m, bn_m, prelu):
    """Stem on the raw image: f32 im2col -> split -> 1x1 split conv with the folded BatchNorm + PReLU."""
    cout, cin, r, s = conv_m.weight.shape
    col = ops.stem_im2col(raw.raw, r, s, conv_m.stride[0], conv_m.padding[0], dtype=F32)
    xs = x3_from_f32(col)
    n, p, q, kp = col.shape
    k = r * s * cin
    w2 = conv_m.weight.detach().permute(0, 2, 3, 1).reshape(cout, k, 1, 1)
    stamp = (conv_m.weight._version, ops.WEIGHT_EPOCH, conv_m.weight.data_ptr())
    hit = conv_m.__dict__.get("_msml_x3_stem")
    if hit is None or hit[0] != stamp:
        hit = (stamp, ops.pack_weight(_x3_expand(w2.float(), 1, [(k, kp)]), False, 3 * kp, 0, BF16))
        conv_m.__dict__["_msml_x3_stem"] = hit
    wp = hit[1]
    coutp = cpad(cout)
    coef = _eval_bn_coef(bn_m, coutp)
    out = _x3_empty(n, p, q, coutp, col.device)
    with ops.PROFILE.rec("conv_x3", 2.0 * n * p * q * k * cout):
        call("msml_conv2d_x3", xs.t, kp, None, 0, wp, wp.shape[0], coef[0], coef[1],
             prelu.weight if prelu is not None else None, None, 0, out.t, coutp, n, p, q, p, q, 1, 1, 1, 0, 0, 0)
    return out


def flat_fc_x3(x, fc_m):
    """flatten(C,H,W) + Linear on a split map: split-K GEMM over K = H*W*3C, f32 result [N, E]."""
    n, h, w, c = x.shape
    e = fc_m.out_features
    stamp = (fc_m.weight._version, ops.WEIGHT_EPOCH, fc_m.weight.data_ptr())
    hit = fc_m.__dict__.get("_msml_x3_fc")
    if hit is None or hit[0] != stamp:
        w4 = fc_m.weight.detach().float().view(e, c, h, w)
        hit = (stamp, ops.pack_weight(_x3_expand(w4, 1, [(c, c)]), False, 3 * c, 0, BF16))      # [E][H*W*3C]
        fc_m.__dict__["_msml_x3_fc"] = hit
    wp = hit[1]
    y = ops.gemm_splitk(x.t.reshape(n, h * w * 3 * c), wp, cpad(e))
    return (y[:, :e] + fc_m.bias.detach()).reshape(n, 1, 1, e)


# ---------------------------------------------------------------------------------------------
# Peer-guided FM branch (fmoperator.py:293-308), MSE distillation loss, dropout.
class _FmAct(torch.autograd.Function):
    """M = act(x) materialised (the input of conv_m)."""

    @staticmethod
    def forward(ctx, x, act):
        m = torch.empty_like(x)
        call("msml_fm_act_fwd", x, m, x.numel(), act, DTYPE_OF[x.dtype])
        ctx.save_for_backward(x)
        ctx.act = act
        return m

    @staticmethod
    def backward(ctx, dm):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        call("msml_fm_act_bwd", dm.contiguous(), x, dx, x.numel(), ctx.act, DTYPE_OF[x.dtype])
        return dx, None


def fm_act(x, act):
    return _FmAct.apply(x, ACTS[act])


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        call("msml_mul_fwd", a, b, out, a.numel(), DTYPE_OF[a.dtype])
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        call("msml_mul_bwd", g.contiguous(), a, b, da, db, a.numel(), DTYPE_OF[a.dtype])
        return da, db


def mul(a, b):
    return _Mul.apply(a, b)


class _Axpb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, b):
        y = torch.empty_like(x)
        call("msml_axpb", x, y, x.numel(), float(a), float(b), DTYPE_OF[x.dtype])
        ctx.a = a
        return y

    @staticmethod
    def backward(ctx, g):
        dx = torch.empty_like(g)
        call("msml_axpb", g.contiguous(), dx, g.numel(), float(ctx.a), 0.0, DTYPE_OF[g.dtype])
        return dx, None, None


def axpb(x, a, b):
    return _Axpb.apply(x, a, b)


class _Mse(torch.autograd.Function):
    """torch.nn.MSELoss()(a, b) over the REAL channels of two NHWC storage tensors (pad channels are zero
    in both and do not contribute to the sum; `count` is the real element count)."""

    @staticmethod
    def forward(ctx, a, b, count):
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        ws = torch.empty(2048, dtype=torch.float64, device=a.device)
        call("msml_mse_fwd", a, b, a.numel(), float(count), loss, ws, ws.numel(), DTYPE_OF[a.dtype])
        ctx.save_for_backward(a, b)
        ctx.count = count
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        gs = g.reshape(1).float().contiguous()
        call("msml_mse_bwd", a, b, gs, float(ctx.count), da, db, a.numel(), DTYPE_OF[a.dtype])
        return da, db, None


def mse(a, b, real_channels):
    n, h, w, _ = a.shape
    return _Mse.apply(a, b, n * h * w * real_channels)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        y = torch.empty_like(x)
        if isinstance(seed, torch.Tensor):       # device-resident seed (graph capture)
            call("msml_dropout_dev", x, y, x.numel(), float(p), seed, DTYPE_OF[x.dtype])
        else:
            call("msml_dropout", x, y, x.numel(), float(p), int(seed), DTYPE_OF[x.dtype])
        ctx.cfg = (p, seed)
        return y

    @staticmethod
    def backward(ctx, g):
        p, seed = ctx.cfg
        dx = torch.empty_like(g)
        if isinstance(seed, torch.Tensor):
            call("msml_dropout_dev", g.contiguous(), dx, g.numel(), float(p), seed, DTYPE_OF[g.dtype])
        else:
            call("msml_dropout", g.contiguous(), dx, g.numel(), float(p), int(seed), DTYPE_OF[g.dtype])
        return dx, None, None


_DROP_COUNTER = [0]
_DROP_STATE = {}           # device -> [torch.initial_seed() the tensor was derived from, int64 seed tensor]
_M63 = 0x7FFFFFFFFFFFFFFF


def _capture_seed_base(s0):
    """Start of the device-resident seed sequence captured steps draw from: a different affine map of the torch seed
    than the eager sequence `s0 * 1000003 + counter` (the two would otherwise coincide step for step: replay k re-drew
    the mask of eager call k -- ADVICE r3), with 2^40 calls of headroom before the int64 wraps."""
    return ((s0 * 2862933555777941757 + 3037000493) & (_M63 >> 1)) | (1 << 40)


def dropout(x, p, seed=None):
    """nn.Dropout(p) in training mode on a storage tensor (iresnet.py:231); the mask comes from a
    counter-based hash, a fresh seed per call unless one is given.  Under hipGraph capture a host-side seed would be
    baked into the graph (every replay the same mask): the seed then lives in a device tensor that the captured
    sequence itself advances -- copy it for this call's forward / backward, add 1 for the next call or replay.
    The device sequence is one per device (every captured graph of the process draws from it, so two graphs never
    repeat each other's masks) and is re-derived, in place, when torch.manual_seed() changed the seed since the last
    eager call."""
    if seed is None:
        if torch.cuda.is_current_stream_capturing():
            st = _DROP_STATE.get(x.device)
            if st is None:       # (a tensor created inside the capture would be re-initialised by every replay)
                raise RuntimeError("msml_amd dropout under graph capture: run one eager (warm-up) step first")
            seed = st[1].clone()
            st[1].add_(1)
        else:
            s0 = torch.initial_seed()
            st = _DROP_STATE.get(x.device)
            if st is None:      # the seed tensor captured steps will read and advance
                _DROP_STATE[x.device] = [s0, torch.full((1,), _capture_seed_base(s0), dtype=torch.int64, device=x.device)]
            elif st[0] != s0:   # torch.manual_seed() since: restart the device sequence (same tensor: graphs hold its address)
                st[0] = s0
                st[1].fill_(_capture_seed_base(s0))
            _DROP_COUNTER[0] += 1
            seed = (s0 * 1000003 + _DROP_COUNTER[0]) & _M63
    return _Dropout.apply(x, p, seed)


def relu_res(x, residual=None):
    """relu(x [+ residual]) without autograd (DeepMind decoder, whose output the reference discards)."""
    c = x.shape[-1]
    m = x.numel() // c
    dev = x.device
    one = torch.ones(c, dtype=torch.float32, device=dev)
    zero = torch.zeros(c, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    call("msml_bn_act_fwd", x, one, zero, zero, residual, 1, y, m, c, DTYPE_OF[x.dtype])
    return y
