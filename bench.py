#!/usr/bin/env python
"""Benchmark of the MSML hot path on MI355X (driver contract, see DESIGN.md 'Measurement').

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full training pass over one synthetic batch: MSML (OSB + FRB + FM) forward,
PartialFC/ArcFace head forward+backward (class-parallel over the N ranks), consensus seg loss,
backbone backward, gradient all-reduce (N > 1), fused clip + SGD on backbone and head.
Workload (BASELINE.json configs[2], the one `metric` is quoted on): ires50-MSML + 85 742-id
PartialFC, 112x112, batch 256 per GPU, bf16 operands / f32 accumulation.  Weak scaling.

Rank 0 prints ONE JSON line: images/sec over all ranks, plus
  roofline     -- the implicit-GEMM conv kernel: algorithmic FLOP of its launches in the timed
                  region / their summed duration (HIP events on the launch stream) vs the dense
                  bf16 MFMA peak
  cpu_baseline -- (N == 1 only) the CPU oracle's training step on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # MI355X_MICROARCH.md (dense)
ROUND = "r06"                                     # PMC summaries of OTHER rounds are never quoted (kernels change)
# forward GFLOP per image (hooks on the imported reference); F_train = 3 x F_fwd (section 3)
F_FWD_GF = {"iresnet18": 8.446, "iresnet34": 12.146, "iresnet50": 15.845, "iresnet100": 27.406}
PEER_OFF = {"use_ori": False, "use_conv": False, "mask_trans": "conv", "use_decoder": False}
JSON_OUT = None                                   # the process's real stdout (main() points fd 1 at stderr)

# Box calibration.  The boxes this bench has run on differ by +-2.5 % on an unchanged build (19 bench lines
# of one build, profiles/r05_bench_box_*.json).  `calibration` records, in the SAME run and right before the timed region: (i) a
# register-resident bf16 MFMA loop on random operands (msml_probe_mfma, csrc/probe.hip), (ii) the same with every operand re-read
# from LDS in the halo conv's mix (msml_probe_mfma_lds), (iii) a 2 x 512 MiB device copy, (iv) a fixed 8192 x 8192 x 1024 GEMM on
# the im2col kernel (calibrate_gemm) -- and what they showed is that NONE of them predicts how fast a box runs the conv families:
# boxes with the lowest MFMA probes (1 609-1 620 LDS-fed) ran the step at 28.8-29.1 ms, one with a middle probe (1 645) at 29.9.
# What does predict it is the step's own dominant launch timed alone in the run's one-stream event pass (`roofline.achieved`:
# 839-920 TFLOP/s over those lines; conv + weight-gradient time follows it monotonically).  `value_normalised` therefore rescales
# the MFMA families' share of the kernel time by (REF_DOMINANT / that rate) ^ DOM_EXP (normalise()): raw spread +-2.5 % ->
# +-1.1 %.  The price: a change of THAT launch's own speed cancels out of value_normalised (it shows in `roofline.frac` and in
# `value`); everything else in the step shows.  The reference constant is arbitrary but fixed: only ratios between runs mean
# anything.
REF_DOMINANT_TFLOPS = 900.0
# how strongly the step's MFMA families follow the dominant launch's isolated rate (least squares over the first 16 undisturbed lines:
# the families are a mix of MFMA-, LDS- and byte-bound launches, so less than proportionally)
DOM_EXP = 0.7
MFMA_FAMILIES = ("conv_igemm", "conv_wgrad", "gemm_splitk", "conv_x3", "conv_fused")


def _sclk_mhz():
    """Current shader clock from sysfs (the starred line of pp_dpm_sclk), None when the box does not expose it."""
    import glob
    for path in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(path):
                if "*" in line:
                    return float(line.split(":")[1].split("M")[0])
        except (OSError, ValueError, IndexError):
            continue
    return None


def calibrate():
    """(register-resident MFMA TFLOP/s, copy TB/s, LDS-fed MFMA TFLOP/s) of this box, measured with HIP events on the
    current stream; ~0.6 s."""
    from msml_amd import _lib
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(7)
    seed = torch.randn(4096, generator=g).to(torch.bfloat16).to(dev)
    wgs, iters = 512, 20000
    out = torch.empty(wgs * 256, dtype=torch.float32, device=dev)
    flop = wgs * 4 * iters * 16 * 2.0 * 16 * 16 * 32

    def run(fn, n_warm, n_timed):
        for _ in range(n_warm):
            fn()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_timed + 1)]
        evs[0].record()
        for i in range(n_timed):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_timed))
        return ts[len(ts) // 2] * 1e-3
    t_mfma = run(lambda: _lib.call("msml_probe_mfma", seed, out, wgs, iters), 12, 15)
    wgs2, iters2 = 256, 16000
    flop2 = wgs2 * 8 * iters2 * 14 * 2.0 * 16 * 16 * 32
    t_lds = run(lambda: _lib.call("msml_probe_mfma_lds", seed, out, wgs2, iters2), 12, 15)
    src = torch.empty(128 << 20, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    t_copy = run(lambda: dst.copy_(src), 4, 9)
    nbytes = 2.0 * src.numel() * 4
    del src, dst
    return flop / t_mfma / 1e12, nbytes / t_copy / 1e12, flop2 / t_lds / 1e12


def calibrate_gemm():
    """TFLOP/s of a fixed 8192 x 8192 x 1024 bf16 GEMM on the im2col kernel's split-K entry (msml_gemm_splitk: LDS-DMA
    operand fills, LDS fragment reads, 32x32x16 MFMAs, two f32 slabs + their reduce) -- a probe that loads the chip the
    way the conv families do (global loads + LDS + MFMA + stores at once), which the two MFMA loops do not."""
    from msml_amd import _lib, ops
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(11)
    m, k, n = 8192, 8192, 1024
    a = (torch.randn(m, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, 1, 1, generator=g) * 0.05).to(dev)
    wp = ops.pack_weight(w, False, k, 0, _lib.BF16)
    need = _lib.value("msml_gemm_splitk_workspace", m, n, k)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out = torch.empty(m, n, dtype=torch.float32, device=dev)

    def fn():
        _lib.call("msml_gemm_splitk", a, m, k, wp, wp.shape[0], out, n, ws, need, _lib.BF16)
    for _ in range(6):
        fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(16)]
    evs[0].record()
    for i in range(15):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(15))
    return 2.0 * m * k * n / (ts[7] * 1e-3) / 1e12


def normalise(value, calib, kernels, dominant=None):
    """value x (what this box costs relative to the reference box): the MFMA families' share of the kernel time scales with
    (REF_DOMINANT / isolated rate of the dominant launch) ^ DOM_EXP, everything else (BatchNorm / element-wise / optimizer:
    HBM-bound) is taken as box-independent.  dominant: TFLOP/s of the dominant launch in the one-stream event pass of the same
    run (default: calib["dominant_tflops"]); without it the value is returned as it is."""
    share = 0.76                                       # (round-4 family table; used when the run has no kernel events)
    if kernels:
        tot = sum(v["ms_per_step"] for v in kernels.values())
        if tot > 0:
            share = sum(v["ms_per_step"] for k, v in kernels.items() if k in MFMA_FAMILIES) / tot
    if dominant is None:
        dominant = calib.get("dominant_tflops")
    if not dominant:
        return value, share
    slow = share * (REF_DOMINANT_TFLOPS / dominant) ** DOM_EXP + (1.0 - share)
    return value * slow, share


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frb", default="iresnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--classes", type=int, default=85742)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--precision", default=None, choices=["bf16x3", "bf16"],
                    help="inference precision with --dtype bf16 (default: the model's default, bf16x3)")
    ap.add_argument("--no-extra-modes", action="store_true",
                    help="skip the short extra measurements of the other precision modes (rank 0, N == 1)")
    ap.add_argument("--data", default="resident", choices=["resident", "device-synth"],
                    help="resident: 4 synthetic batches resident in HBM (the contract's timed region); "
                         "device-synth: uint8 faces from pinned host memory through msml_amd.data.DeviceLoaderX "
                         "(H2D on a side stream + occlusion / flip / light / normalise kernels) every step")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="config 4 sizing on ONE GPU: run the head as rank 0 of an E-way class-parallel job "
                         "(classes/E local rows, batch*E gathered feature rows, collectives omitted)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-calibration", action="store_true", help="skip the box calibration probes (MFMA loop + device copy)")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="graph: replay ONE captured hipGraph per step; eager: issue the ~1300 launches "
                         "from Python with weight-gradient kernels on a second stream; auto: time both "
                         "during warm-up and keep the faster")
    ap.add_argument("--no-graph", action="store_true", help="alias of --launch eager")
    ap.add_argument("--cpu-share", type=int, default=0,
                    help="pin this process to the first K CPUs of its affinity mask before anything touches the GPU: what ONE "
                         "rank of an 8-rank job gets on a 16-CPU box share is K = 2 (the eager step needs one Python thread "
                         "at 60-80 %% duty; DESIGN section 6)")
    return ap.parse_args()


class Trainer:
    """The training step of train.py:252-318 (op2 / PartialFC branch) on the HIP path."""

    def __init__(self, args, rank, local_rank, world):
        from msml_amd import functional as Fh
        from msml_amd.backbones import MSML
        from msml_amd.headers import ArcMargin, PartialFC
        from msml_amd.optim import FlatSGD, reference_param_groups
        from msml_amd.tricks.consensus_loss import StructureConsensuLossFunction
        from msml_amd import synthetic
        self.Fh = Fh
        self.world = world
        fp16 = args.dtype == "bf16"
        dev = torch.device("cuda", local_rank)
        torch.manual_seed(1234)                      # same init on every rank (train.py:133-134)
        self.model = MSML(args.frb, "unet", (1, 1, 1, 1), 8, fp16=fp16,
                          fm_params=(3, 2, "sigmoid", "mul"), header_type="AMArcFace",
                          header_params=(64.0, 0.48, 0.0, 0.0), peer_params=dict(PEER_OFF)).to(dev)
        for p in self.model.classification.parameters():   # live full-class head unused here
            p.requires_grad_(False)
        self.model.train()
        self.emu = max(1, args.emulate_world)
        if self.emu > 1:
            assert world == 1
            self.pfc = PartialFC(0, local_rank, 1, args.batch * self.emu, False, ArcMargin(64.0, 0.48, 0.0, 0.0),
                                 args.classes // self.emu, fp16=fp16)
        else:
            self.pfc = PartialFC(rank, local_rank, world, args.batch, False,
                                 ArcMargin(64.0, 0.48, 0.0, 0.0), args.classes, fp16=fp16)
        self.opt = FlatSGD(reference_param_groups(self.model, args.batch, world), 0.9, 5e-4, 5.0)
        if world > 1 or os.environ.get("MSML_FORCE_DIST"):
            self.opt.enable_overlap(world)
        self.opt_pfc = FlatSGD([{"params": [self.pfc.sub_weight], "lr": 0.1 / 512 * args.batch * world}],
                               0.9, 5e-4, None)
        self.pfc.adopt_flat_optimizer(self.opt_pfc)     # weight / momentum / dW live in the head's arenas
        self.seg_crit = StructureConsensuLossFunction(10.0, 5.0, "idx", "idx")
        # a short stream of synthetic batches resident in HBM (cycled), so the head cannot
        # simply memorise one batch during the run
        self.batches = []
        for i in range(4):
            seed = 1 + rank + 100 * i
            x = synthetic.images(args.batch, seed=seed)
            x, msk = synthetic.rect_occlusion(x, seed=seed)
            lab = synthetic.labels(args.batch, args.classes, seed=seed)
            self.batches.append((x.to(dev), msk.to(dev), lab.to(dev)))
        self.it = 0
        self.phases = None
        self.loader = None
        if getattr(args, "data", "resident") == "device-synth":
            from msml_amd import data
            self.loader = iter(data.DeviceLoaderX(data.SynthFaceSource(args.batch, args.classes, steps=None, seed=1 + rank),
                                                  local_rank, seed=1 + rank, mode="train", want_ori=False))

    def next_batch(self):
        if self.loader is not None:
            img, msk, _, lab = next(self.loader)
            return img, msk, lab
        b = self.batches[self.it % len(self.batches)]
        self.it += 1
        return b

    def mark(self, name):
        """MSML_BENCH_PHASES=1: an event on the main stream at every phase boundary of the step (diagnostic)."""
        if self.phases is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.phases.append((name, ev))

    def step(self, batch=None):
        Fh = self.Fh
        x, msk, label = batch if batch is not None else self.next_batch()
        self.mark("start")
        self.opt.zero_grad()
        if self.world > 1:
            self.pfc.prefetch_labels(label)          # label all-gather on the head's side stream
        feature, final_seg, kd = self.model(x)                   # head-less training return
        seg_loss = self.seg_crit(final_seg, msk, msk)
        fn = Fh.normalize(feature)
        self.mark("forward")
        if self.world > 1:
            # the OSB backward depends on final_seg only: issue it first so that it runs (on the OSB
            # stream) underneath the head's collectives; the FRB backward follows once dX is back
            seg_loss.backward()
            x_grad, loss_v = self.pfc.forward_backward(label, fn, self.opt_pfc)
            fn.backward(x_grad)
        elif self.emu > 1:
            # rank 0 of an emu-way job: batch * emu gathered rows against classes / emu local rows (rows
            # whose class lives elsewhere map to -1 exactly as on a real rank)
            xg, loss_v = self.pfc.forward_backward(label.repeat(self.emu), fn.detach().repeat(self.emu, 1), self.opt_pfc)
            x_grad = xg[:fn.shape[0]] * float(self.emu)
            torch.autograd.backward([fn, seg_loss], [x_grad, None])
        else:
            # one engine call: the OSB nodes, created first, run last (as its own call the OSB backward
            # measured 1 % slower at world size 1 -- 4 ms of host time before the FRB backward starts)
            x_grad, loss_v = self.pfc.forward_backward(label, fn, self.opt_pfc)
            self.mark("head")
            torch.autograd.backward([fn, seg_loss], [x_grad, None])
        self.mark("backward")
        self.opt.all_reduce_grads(self.world)
        self.opt.step()
        self.opt_pfc.step()                          # dW was written straight into its gradient arena
        self.mark("optimizer")
        return loss_v, seg_loss


class Inferer:
    """Config 5: embedding extraction, orig + h-flip passes summed (qeval_mxnet.py:326-390)."""

    def __init__(self, args, rank, local_rank, world):
        from msml_amd.backbones import MSML
        from msml_amd import synthetic
        dev = torch.device("cuda", local_rank)
        torch.manual_seed(1234)
        self.model = MSML(args.frb, "unet", (1, 1, 1, 1), 8, fp16=args.dtype == "bf16",
                          fm_params=(3, 2, "sigmoid", "mul"), header_type="AMArcFace",
                          header_params=(64.0, 0.48, 0.0, 0.0), peer_params=dict(PEER_OFF)).to(dev).eval()
        if args.precision:
            self.model.eval_precision = args.precision
        self.precision = self.model.eval_precision if args.dtype == "bf16" else "f32"
        a, b, _ = synthetic.occluded_pairs(args.batch // 2, seed=1 + rank)
        self.x = torch.cat((a, b)).to(dev)

    def next_batch(self):
        return (self.x,)

    @torch.no_grad()
    def step(self, batch=None):
        x = batch[0] if batch is not None else self.x
        f1, _ = self.model(x)
        f2, _ = self.model(x.flip(3))
        return torch.nn.functional.normalize(f1 + f2), None


def cpu_baseline(args):
    """The CPU oracle's training step on a bounded sample (rank 0, N == 1), protocol of SURVEY section 8d: batch 16,
    2 warm-up + 5 timed steps, median; the thread count is the best of {all cores, 64, 32} (one step each)."""
    from msml_amd import synthetic
    from oracle import model as om
    bs = 16
    torch.manual_seed(0)
    m = om.MSML(args.frb, "unet", (1, 1, 1, 1), 8, fm_params=(3, 2, "sigmoid", "mul"),
                header_type="AMArcFace", header_params=(64.0, 0.48, 0.0, 0.0))
    m.train()
    w = torch.randn(args.classes, 512) * 0.01
    mom = torch.zeros_like(w)
    x = synthetic.images(bs, 1)
    x, msk = synthetic.rect_occlusion(x, 1)
    label = synthetic.labels(bs, args.classes, 1)
    params = [p for n, p in m.named_parameters() if "classification" not in n and p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.1 / 512 * bs, momentum=0.9, weight_decay=5e-4)
    margin = lambda lg, lab: om.margin_logits(lg, lab, "arc", 64.0, 0.48, 0.0, 0.0)  # noqa: E731

    def step():
        t0 = time.time()
        opt.zero_grad()
        seg = m.osb(x)
        feat, _ = m.frb(x, [seg[3], seg[2], seg[1], seg[0]], None)
        fn = torch.nn.functional.normalize(feat)
        loss, dx, dw = om.pfc_rank_step(fn.detach(), label, w, 0, margin, lambda t: t, lambda t: t)
        seg_loss = om.consensus_loss(seg[4], msk)
        torch.autograd.backward([fn, seg_loss], [dx, None])
        torch.nn.utils.clip_grad_norm_(params, 5, 2)
        opt.step()
        g = dw + 5e-4 * w
        mom.mul_(0.9).add_(g)
        w.sub_(0.1 / 512 * bs * mom)
        return time.time() - t0
    ncpu = torch.get_num_threads()      # torch's default (physical cores); os.cpu_count() counts SMT siblings, and one
    # step with 256 threads on this pool's boxes took 399 s (8.3 s with 128) -- never go above the default
    budget = float(os.environ.get("MSML_CPU_BASELINE_S", "60"))       # wall-clock bound of this leg
    t_begin = time.time()

    def note(what, dt):                               # progress on stderr: a silent minute looks like a hang
        print("cpu_baseline: %s %.2f s" % (what, dt), file=sys.stderr, flush=True)
    note("first step (%d threads)" % torch.get_num_threads(), step())   # allocator / oneDNN primitive caches
    sweep = {}
    for c in sorted({c for c in (ncpu, 64, 32, 16) if c <= ncpu}, reverse=True):
        if sweep and time.time() - t_begin > budget / 2:
            break
        torch.set_num_threads(c)
        sweep[c] = step()
        note("%d threads" % c, sweep[c])
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    note("warm-up at %d threads" % best, step())      # second warm-up step at the chosen thread count
    times = []
    while len(times) < 5 and (len(times) < 2 or time.time() - t_begin + sweep[best] < budget):
        times.append(step())
        note("timed step %d" % len(times), times[-1])
    times.sort()
    med = times[len(times) // 2]
    torch.set_num_threads(ncpu)
    return {"value": round(bs / med, 3), "unit": "images/sec", "cores": best, "host_cores": os.cpu_count(),
            "cpu_model": cpu_model(), "kind": "port",
            "thread_sweep_s_per_step": {str(c): round(t, 2) for c, t in sweep.items()},
            "sample": "CPU oracle training step (%s-MSML + %d-id head), batch %d, f32: 2 warm-up + %d timed steps "
                      "(5 unless the %.0f s bound of this leg cut them short), median %.2f s (min %.2f, max %.2f), "
                      "torch.set_num_threads(%d)"
                      % (args.frb, args.classes, bs, len(times), budget, med, times[0], times[-1], best)}


def pmc_traffic(label):
    """(HBM bytes per launch, source file) of `label` from THIS round's committed PMC summary
    (profiles/<ROUND>_pmc_traffic.json, written from a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of the kernels as
    they are now); (None, None) when the round has no counters for the label -- a summary of an earlier round is never
    quoted for a kernel that has changed since."""
    path = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % ROUND)
    try:
        pm = json.load(open(path))
    except Exception:
        return None, None
    base = label.split(" [")[0]
    for k, v in pm.items():
        if isinstance(v, dict) and "hbm_bytes" in v and (k == label or k.split(" [")[0] == base):
            return v["hbm_bytes"], os.path.relpath(path, ROOT)
    return None, None


def pmc_family(family):
    """(average HBM bytes per launch, launches per step, source file) of a kernel family ('conv', 'wgrad') from THIS
    round's whole-step PMC summary (profiles/<ROUND>_pmc_step.json, tools/pmc_step.py: FETCH_SIZE / WRITE_SIZE passes over
    `bench.py --steps 2`, summed over every kernel of the family); (None, None, None) without current counters."""
    path = os.path.join(ROOT, "profiles", "%s_pmc_step.json" % ROUND)
    try:
        f = json.load(open(path))["families"][family]
        return f["hbm_bytes_per_launch"], f["launches_per_step"], os.path.relpath(path, ROOT)
    except Exception:
        return None, None, None


def memory_table(runner, args):
    """HBM sizing of the run (288 GB per GPU): torch's allocator counters plus
    the resident state by owner; activations = peak minus resident state."""
    gb = 2.0 ** 30
    t = {"hbm_capacity_gb": round(torch.cuda.get_device_properties(0).total_memory / gb, 1),
         "max_allocated_gb": round(torch.cuda.max_memory_allocated() / gb, 2),
         "reserved_gb": round(torch.cuda.memory_reserved() / gb, 2),
         "allocated_now_gb": round(torch.cuda.memory_allocated() / gb, 2)}
    if args.mode != "train":
        return t
    bb = runner.opt.flat_w.numel() * 4
    hw = runner.opt_pfc.flat_w.numel() * 4
    n_rows = args.batch * max(runner.world, runner.emu)
    ncls = runner.pfc.num_local
    el = 2 if args.dtype == "bf16" else 4
    resident = 3 * bb + 3 * hw
    t.update({
        "backbone_params_grads_momentum_gb": round(3 * bb / gb, 3),
        "head_rows": ncls, "head_weight_grad_momentum_gb": round(3 * hw / gb, 3),
        "head_normalised_copy_and_transpose_gb": round(2 * ncls * 512 * el / gb, 3),
        "head_logits_f32_gb": round(n_rows * ncls * 4 / gb, 3),
        "head_dcos_gb": round(n_rows * ncls * el / gb, 3),
        "activations_and_workspaces_peak_gb": round((torch.cuda.max_memory_allocated() - resident) / gb, 2),
    })
    return t


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def timed_throttled(fn, steps, warm):
    """timed() with at most two steps in flight (as the headline's loop: a host that runs many steps ahead pins every
    step's activations until the caching allocator starts returning memory to the driver)."""
    inflight = []

    def one():
        if len(inflight) >= 2:
            inflight.pop(0).synchronize()
        fn()
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
    for _ in range(warm):
        one()
    torch.cuda.synchronize()
    inflight.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def train_mode(a, rank, local_rank, steps, warm, what):
    """One more BASELINE.json training configuration on the line (`modes`): the SAME step as the headline (Trainer.step)
    on another network / head size, issued eagerly with the side streams AND as one hipGraph replay; `value` is the
    faster of the two (as --launch auto), both times are reported, plus the sum of the step's kernel-event times on one
    stream (what the GPU needs when nothing overlaps and nothing waits for the host) and the whole-step roofline fraction."""
    import gc
    from msml_amd import ops
    r = Trainer(a, rank, local_rank, 1)
    side, osb = torch.cuda.Stream(), torch.cuda.Stream()
    ops.WGRAD_STREAM, ops.OSB_STREAM = side, osb
    t_eager = timed_throttled(r.step, steps, warm)
    t_graph, graph = None, None
    try:
        s0 = torch.cuda.Stream()
        s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            r.step()
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.synchronize()
        static = tuple(t.clone() for t in r.next_batch())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            r.step(static)

        def replay():
            for dst, src in zip(static, r.next_batch()):
                dst.copy_(src)
            graph.replay()
        t_graph = timed_throttled(replay, steps, 3)
    except Exception as e:                                  # pragma: no cover (diagnostic path)
        print("train_mode: graph capture failed: %r" % (e,), file=sys.stderr)
        torch.cuda.synchronize()
    graph = None
    gc.collect()
    # kernel time of one step, one stream, an event pair around every instrumented launch
    ops.WGRAD_STREAM = ops.OSB_STREAM = None
    r.step()
    ops.PROFILE.start()
    for _ in range(2):
        r.step()
    prof = ops.PROFILE.stop()
    k_ms = sum(v["ms"] for v in prof.values()) / 2
    dt = min(t for t in (t_eager, t_graph) if t is not None)
    value = a.batch / dt
    f_img = F_FWD_GF[a.frb] * 3.0
    out = {"value": round(value, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 2), "batch": a.batch,
           "launch": "hipGraph replay" if (t_graph is not None and t_graph <= t_eager) else "eager, side streams",
           "ms_per_step_eager": round(t_eager * 1e3, 2),
           "ms_per_step_graph": None if t_graph is None else round(t_graph * 1e3, 2),
           "kernel_event_ms_per_step": round(k_ms, 2),
           "roofline_step": {"bound": "mfma", "achieved": round(value * f_img / 1e3, 2), "peak": PEAK_TFLOPS["bf16"],
                             "unit": "TFLOP/s", "frac": round(value * f_img / 1e3 / PEAK_TFLOPS["bf16"], 4),
                             "gflop_per_image": round(f_img, 3)},
           "what": what}
    if r.emu > 1:
        out["head"] = "rank 0 of a %d-way class-parallel %d-id head: %d local rows x %d gathered feature rows, collectives omitted" \
            % (r.emu, a.classes, a.classes // r.emu, a.batch * r.emu)
        out["max_allocated_gb"] = round(torch.cuda.max_memory_allocated() / 2.0 ** 30, 2)
    del r
    gc.collect()
    torch.cuda.empty_cache()
    return out


def extra_modes(args, rank, local_rank):
    """Short measurements of the other precision modes on the same workload (rank 0, N == 1), so that
    every mode that meets the reference's tolerances has a number from the same run as the headline:
      train_f32     exact-f32 MFMA training step (the parity mode of the training path)
      infer_bf16x3  embedding extraction, orig + flip, split-bf16 (meets 1e-3 / bit-exact masks; default)
      infer_bf16    the same in plain bf16 (throughput mode, ~8e-3 embedding error)."""
    import copy
    import gc
    out = {}
    a = copy.copy(args)
    a.mode, a.batch = "infer", 1024
    for prec in ("bf16x3", "bf16"):
        a.dtype, a.precision = "bf16", prec
        r = Inferer(a, rank, local_rank, 1)
        dt = timed(r.step, 4, 3)
        out["infer_" + prec] = {"value": round(a.batch / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 2),
                                "batch": a.batch, "what": "%s-MSML embedding extraction, orig + h-flip" % args.frb}
        del r
        gc.collect()
        torch.cuda.empty_cache()
    a = copy.copy(args)
    a.data = "device-synth"
    from msml_amd import ops
    r = Trainer(a, rank, local_rank, 1)
    side, osb = torch.cuda.Stream(), torch.cuda.Stream()
    ops.WGRAD_STREAM, ops.OSB_STREAM = side, osb
    dt = timed(r.step, 12, 6)            # (3 warm-up steps left allocator growth / arena creation inside the timed six: 35.7 ms reported for a 30.6 ms step)
    ops.WGRAD_STREAM = ops.OSB_STREAM = None
    out["train_bf16_device_input"] = {"value": round(a.batch / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 2),
                                      "batch": a.batch, "what": "the headline step fed by msml_amd.data.DeviceLoaderX: uint8 faces "
                                      "from pinned host memory, H2D + occlusion / flip / light / normalise on the GPU every step"}
    del r
    gc.collect()
    torch.cuda.empty_cache()
    # BASELINE.json configs[1] and configs[3]: the same training step on the other two networks
    a = copy.copy(args)
    a.frb, a.classes, a.batch, a.emulate_world, a.data = "iresnet18", 10000, 128, 1, "resident"
    out["train_config2"] = train_mode(a, rank, local_rank, 20, 8, "BASELINE.json configs[1]: ires18-MSML + ArcFace 10 000-id "
                                      "PartialFC, batch 128, bf16, 1 x MI355X -- fwd + bwd + clip + SGD")
    a = copy.copy(args)
    a.frb, a.classes, a.batch, a.emulate_world, a.data = "iresnet100", 2000000, 256, 8, "resident"
    torch.cuda.reset_peak_memory_stats()
    out["train_config4_shard"] = train_mode(a, rank, local_rank, 6, 4, "BASELINE.json configs[3] at its per-GPU size: ires100-"
                                            "variant MSML + one of eight 250 000-row shards of a 2 000 000-id PartialFC, "
                                            "batch 256 per GPU (2048 gathered feature rows), bf16")
    a = copy.copy(args)
    a.dtype, a.mode, a.precision = "f32", "train", None
    from msml_amd import ops
    ops.WGRAD_STREAM = ops.OSB_STREAM = None
    r = Trainer(a, rank, local_rank, 1)
    dt = timed(r.step, 3, 2)
    out["train_f32"] = {"value": round(a.batch / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 2),
                        "batch": a.batch, "what": "the headline training step on the exact-f32 MFMA path"}
    del r
    gc.collect()
    torch.cuda.empty_cache()
    return out


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, cut by a cgroup CPU quota when one is set (the GPU boxes
    expose 256 logical CPUs to os.cpu_count() but grant a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and period > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def rank_env(n, environ, cpus=None):
    """Environment shared by the n rank processes launch_ranks() starts (RANK / LOCAL_RANK are added per rank)."""
    import socket
    env = dict(environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as sk:                       # a free port (closed again before the ranks bind it)
            sk.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(sk.getsockname()[1])
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this driver)
    # a rank is one Python thread issuing launches plus a few tiny CPU ops: its share of the USABLE CPUs, at most four OpenMP
    # threads (cpu_count() // n handed every rank 32 threads on a 16-CPU share of a 256-CPU host)
    cpus = usable_cpus() if cpus is None else cpus
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(4, cpus // n))))
    return env


def launch_ranks(n):
    """Start `n` rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, one GPU
    each), wait for all of them and return the exit code: 0 only if every rank exited 0.  stdout of rank 0 (the JSON
    line) is passed through as the only stdout of this process; the other ranks' stdout goes to stderr.  A rank that
    dies takes the others down (they would otherwise sit in a collective until the watchdog fires)."""
    import subprocess
    env = rank_env(n, os.environ)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else sys.stderr))
    import signal

    def stop(signum, _frame):                             # killed from outside: do not leave ranks behind
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)
    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:                           # exact children of this process, by handle
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    args = parse()
    if args.cpu_share > 0:
        # (sched_setaffinity only: no exec, nothing GPU-side has been initialised yet; children of launch_ranks inherit it)
        cpus = sorted(os.sched_getaffinity(0))[:args.cpu_share]
        os.sched_setaffinity(0, cpus)
        torch.set_num_threads(max(1, len(cpus)))
        args.cpu_share = len(cpus)                      # (a box with fewer CPUs than asked for: report what was granted)
        args.cpu_set = cpus
    if "WORLD_SIZE" in os.environ or args.gpus == 1:
        # stdout carries the JSON line and nothing else: keep the real stdout for it and point fd 1 at stderr, so that
        # whatever a library prints from C (gloo's "[Gloo] Rank 0 is connected to ...", ROCm notices) cannot land there
        global JSON_OUT
        sys.stdout.flush()
        JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as the reference's launch line
        # does (README.md:33-38, `python -m torch.distributed.launch --nproc_per_node=8 ... train.py`).  This parent
        # has not touched the GPU (importing torch does not) and never will: it relays rank 0's JSON line.
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # rehearsal of the multi-rank control flow on a ONE-GPU box (MSML_BENCH_ONE_GPU=1): every rank on device 0, gloo
    # instead of RCCL (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    one_gpu = bool(os.environ.get("MSML_BENCH_ONE_GPU"))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dist = bool(os.environ.get("MSML_FORCE_DIST"))   # exercise the RCCL path at world == 1
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo" if one_gpu else "nccl", rank=rank, world_size=world)
    from msml_amd import blocks, functional, ops
    assert not blocks.FAULT and blocks.TAP is None and not functional.FAULT, "test instrumentation is on (MSML_FAULT?)"
    if args.no_graph or (world > 1 and args.launch == "auto"):
        # multi-rank: eager only -- capturing RCCL collectives into a hipGraph was verified with a
        # one-rank communicator only, and a capture that fails mid-collective cannot be retried
        args.launch = "eager"
    side_stream = torch.cuda.Stream()
    if os.environ.get("MSML_BENCH_MAIN_PRIORITY"):
        # experiment: the step's critical chain (forward, backward-data, BatchNorm) on a HIGH-priority stream, the side
        # streams (weight gradients, OSB) at the default priority
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the heads announce themselves like the reference's do: stdout
        runner = (Trainer if args.mode == "train" else Inferer)(args, rank, local_rank, world)   # carries the JSON line only

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Two ways to issue the step (~1300 launches):
    #  graph -- captured once into a hipGraph (static input buffers refreshed by a device copy),
    #           immune to host speed; the replay executes the captured work in one queue;
    #  eager -- launches issued from Python, weight-gradient kernels on a second HIP stream so
    #           that they fill the tails of the backward-data / BatchNorm chain.
    # `auto` times a few untimed warm-up steps of each and keeps the faster one.
    graph = None
    static = None

    osb_stream = torch.cuda.Stream()

    def eager_mode(on):
        # (MSML_BENCH_NO_SIDE_STREAMS=1: diagnostic -- everything on one stream; against a single-queue graph replay of
        # the same sequence, MSML_GRAPH_SERIAL=1, the difference is the GPU idle time the host's issue rate causes)
        if os.environ.get("MSML_BENCH_NO_SIDE_STREAMS"):
            on = False
        ops.WGRAD_STREAM = side_stream if on else None
        ops.OSB_STREAM = osb_stream if on else None

    for _ in range(max(args.warmup, 2)):
        runner.step()
    torch.cuda.synchronize()
    t_eager = None
    if args.launch in ("auto", "eager"):
        eager_mode(True)
        runner.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            runner.step()
        barrier()
        t_eager = (time.perf_counter() - t0) / 3
        eager_mode(False)
    t_graph = None
    if args.launch in ("auto", "graph"):
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                runner.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            static = tuple(t.clone() for t in runner.next_batch())
            graph = torch.cuda.CUDAGraph()
            # RCCL collectives are captured too (verified with a 1-rank communicator); the
            # watchdog thread of ProcessGroupNCCL must not trip the capture -> thread_local mode
            mode = "thread_local" if dist.is_initialized() else "global"
            # the capture keeps the three-stream fork / join (weight gradients and the OSB on side streams; every
            # side stream is joined by FlatSGD.step before the capture ends); MSML_GRAPH_SERIAL=1 captures one queue
            if not os.environ.get("MSML_GRAPH_SERIAL"):
                eager_mode(True)
            with torch.cuda.graph(graph, capture_error_mode=mode):
                out = runner.step(static)
            eager_mode(False)
            graph.replay()
            barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                graph.replay()
            barrier()
            t_graph = (time.perf_counter() - t0) / 3
        except Exception as e:                              # pragma: no cover (diagnostic path)
            import traceback
            traceback.print_exc()
            print("graph capture failed, running eagerly: %r" % (e,), file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    use_graph = graph is not None and (t_eager is None or t_graph <= t_eager)
    if world > 1 and t_eager is not None and graph is not None:
        flag = torch.tensor([1.0 if use_graph else 0.0], device="cuda")     # same choice on all ranks
        dist.all_reduce(flag, dist.ReduceOp.MIN)
        use_graph = bool(flag.item() > 0.5)
    if not use_graph:
        # drop the probe graph NOW: destroying it (and returning its private memory pool, ~20 GB of
        # hipFree) otherwise happens whenever Python's cycle collector next runs -- measured as one
        # 0.5-0.8 s step inside the timed loop
        graph = None
        static = None
        out = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        eager_mode(True)

    def one_step():
        if graph is None:
            return runner.step()
        for dst, src in zip(static, runner.next_batch()):
            dst.copy_(src)
        graph.replay()
        return out

    # The host issues an eager step faster than the GPU runs it; left alone it runs many steps
    # ahead, and every step in flight pins its ~10 GB of activations (tensors handed to the side
    # streams cannot be recycled before those streams have passed them), until the caching
    # allocator starts freeing / re-allocating device memory (measured: 36 -> 60+ ms/step on hosts
    # that issue a step in 20 ms).  Keep at most two steps in flight, as a training loop that reads
    # its loss or waits for its data loader does.
    inflight = []
    step_events = []

    def throttled_step():
        if len(inflight) >= 2:
            inflight.pop(0).synchronize()
        r = one_step()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        inflight.append(ev)
        step_events.append(ev)
        return r

    # settle the caching allocator in the steady two-steps-in-flight pattern before timing (the probe
    # graph's pool has just gone back to the driver: the next steps re-grow the eager pool)
    for _ in range(3):
        throttled_step()
    barrier()
    inflight.clear()
    import gc
    gc.collect()
    calib = None
    if not args.no_calibration:           # every rank (the two extra steps below hold collectives); rank 0 reports
        mf, cp, ml = calibrate()
        calib = {"mfma_tflops": round(mf, 1), "mfma_lds_tflops": round(ml, 1), "copy_tbs": round(cp, 3),
                 "gemm_tflops": round(calibrate_gemm(), 1)}
        for _ in range(2):        # the probes leave the caches and the clock in their own state: two steps of the workload again
            throttled_step()
        barrier()
        inflight.clear()
    if args.cpu_share > 0 and len(os.sched_getaffinity(0)) != args.cpu_share:
        # something between the start of the process and here widened the main thread's mask again (intermittent: two of fifteen
        # pinned runs of round 6, one row of round 5's table; the culprit -- a runtime thread start? -- was not found): pin again; the check behind the timed
        # region still fails the run should it happen inside it
        print("bench.py: affinity mask was reset to %d CPUs before the timed region; pinning to %d again"
              % (len(os.sched_getaffinity(0)), args.cpu_share), file=sys.stderr)
        os.sched_setaffinity(0, args.cpu_set)
    sclk = []
    gc.disable()                  # no collector pauses inside the timed region (nothing is skipped)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = throttled_step()
        if calib is not None:
            sclk.append(_sclk_mhz())       # one sysfs read per step (~20 us of host time, the host runs ahead of the GPU)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if args.cpu_share > 0 and len(os.sched_getaffinity(0)) != args.cpu_share:
        # (one row of profiles/r05_cpu_share.log reported cpu_share = 256 for a K = 1 run -- an unpinned run must
        # never be reported as a pinned one)
        raise SystemExit("bench.py: --cpu-share %d but the timed region ran with %d CPUs in the affinity mask"
                         % (args.cpu_share, len(os.sched_getaffinity(0))))
    evs = step_events[-(args.steps + 1):]
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None
    if os.environ.get("MSML_BENCH_STEP_TIMES"):
        print("step ms:", " ".join("%.0f" % v for v in step_ms),
              "| reserved GB %.1f" % (torch.cuda.memory_reserved() / 2 ** 30), file=sys.stderr)
    if os.environ.get("MSML_BENCH_PHASES") and args.mode == "train":
        # diagnostic: where the main stream spends the eager multi-stream step (events at the phase boundaries; the
        # side streams' work shows up in the phase during which the main stream waits for it -- the optimizer's join)
        eager_mode(True)
        acc = {}
        for _ in range(6):
            runner.phases = []
            runner.step()
            torch.cuda.synchronize()
            ph = runner.phases
            for (n0, e0), (n1, e1) in zip(ph, ph[1:]):
                acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1) / 6
        runner.phases = None
        print("phases (ms on the main stream): " + "  ".join("%s %.2f" % kv for kv in acc.items()), file=sys.stderr)
    # roofline pass: the same step, eagerly, with a HIP-event pair around every instrumented
    # launch (events cannot be recorded inside a graph replay); kernel durations are unaffected
    prof = {}
    if not args.no_kernel_events:
        eager_mode(False)             # serial stream: event pairs then bracket one kernel each
        runner.step()                 # (one unrecorded step in THIS stream mode: the caching allocator re-homes the blocks the
        # side streams owned, and a hipMalloc inside an event bracket reads as kernel time -- seen as 0.27 ms for a 0.04 ms
        # launch in the batch-128 profile of config 2)
        ops.PROFILE.start()
        for _ in range(min(args.steps, 3)):
            runner.step()
        prof = ops.PROFILE.stop()
        prof_steps = min(args.steps, 3)
    if world > 1:
        tt = torch.tensor([dt], device="cuda")
        dist.all_reduce(tt, dist.ReduceOp.MAX)
        dt = tt.item()
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    imgs = args.batch * world * args.steps
    value = imgs / dt
    rec = {
        "metric": "images/sec (112x112) %s-MSML+PartialFC %s step" % (args.frb.replace("iresnet", "ires"),
                                                                     "training" if args.mode == "train" else "embedding-extraction"),
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "ms_per_step_median": None if median_ms is None else round(median_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": getattr(runner, "precision", args.dtype),
        "data": "synthetic" if args.data == "resident" else "synthetic uint8 faces, device input pipeline (H2D + occlusion synthesis per step)",
        "config": {"workload": "%s-MSML (OSB r18 + FM x4) + %d-id ArcFace PartialFC, 112x112, batch %d/GPU, %s"
                               % (args.frb, args.classes, args.batch,
                                  "fwd+bwd+clip+SGD" if args.mode == "train" else "orig+flip forward"),
                   "global_batch": args.batch * world, "parallelism": "dp%d+class-parallel head" % world,
                   "launch": ("hipGraph replay" if graph is not None else
                              "eager, weight gradients + OSB on side streams" if args.mode == "train" else "eager"),
                   "cpu_share": len(os.sched_getaffinity(0)),
                   "warmup_ms_per_step": {"eager": None if t_eager is None else round(t_eager * 1e3, 2),
                                          "graph": None if t_graph is None else round(t_graph * 1e3, 2)}},
    }
    if args.mode == "train" and out[0] is not None:
        rec["loss"] = round(float(out[0]), 4)
    if args.frb in F_FWD_GF and args.dtype in PEAK_TFLOPS:
        # whole-step fraction of the conv-MFMA roofline: img/s per GPU x F per image / peak
        f_img = F_FWD_GF[args.frb] * (3.0 if args.mode == "train" else 2.0)       # inference: orig + flip passes
        rec["roofline_step"] = {"bound": "mfma", "achieved": round(value / world * f_img / 1e3, 2),
                                "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                "frac": round(value / world * f_img / 1e3 / PEAK_TFLOPS[args.dtype], 4),
                                "gflop_per_image": round(f_img, 3),
                                "what": "images/sec per GPU x algorithmic GFLOP per image (F_train = 3 x F_fwd) / dense peak"}
    rec["memory"] = memory_table(runner, args)
    if args.mode == "train" and runner.emu > 1:
        rec["config"]["head"] = ("rank 0 of a %d-way class-parallel %d-id head emulated on one GPU: %d local rows x %d "
                                 "gathered feature rows, collectives omitted" % (runner.emu, args.classes,
                                 args.classes // runner.emu, args.batch * runner.emu))
    if prof:
        # aggregate the per-shape event records into kernel families
        fam = {}
        for name, v in prof.items():
            key = "conv_igemm" if name.startswith("conv ") else ("conv_wgrad" if name.startswith("wgrad ") else name)
            d = fam.setdefault(key, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
            for q in d:
                d[q] += v[q]
        if os.environ.get("MSML_PROFILE_DETAIL"):
            for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("MSML_PROFILE_DETAIL") or 45) if os.environ.get("MSML_PROFILE_DETAIL", "1") != "1" else 45]:
                print("%-52s n=%3d ms/step=%7.3f TF/s=%6.1f GB/s=%6.0f" % (name, v["n"] // prof_steps, v["ms"] / prof_steps,
                      v["flops"] / max(v["ms"], 1e-9) / 1e9, v["bytes"] / max(v["ms"], 1e-9) / 1e6), file=sys.stderr)
        peak = PEAK_TFLOPS[args.dtype]
        if "conv_igemm" not in fam:            # inference runs: fused / split-precision conv labels
            fam["conv_igemm"] = dict(fam.get("conv_x3") or fam.get("conv_fused") or
                                     {"ms": 1e-9, "n": 1, "flops": 0.0, "bytes": 0.0})
        k = fam["conv_igemm"]
        ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
        ftraffic, _, fsrc = pmc_family("conv") if args.mode == "train" and args.dtype == "bf16" else (None, None, None)
        rec["roofline_family"] = {"bound": "mfma", "kernel": "k_conv_halo / k_conv_fast / k_conv_igemm (conv fwd + dgrad, every shape of the step)",
                           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": ftraffic, "traffic_source": fsrc,
                           "launches": k["n"], "avg_us": round(k["ms"] * 1e3 / k["n"], 2)}
        # the dominant launch of the step over conv forward / backward-data AND weight-gradient labels
        # (one shape, one kernel): achieved from its own events, HBM traffic from the committed rocprofv3
        # PMC run of the same launch (profiles/*_pmc_traffic.json, keyed by the label)
        cands = {n: v for n, v in prof.items() if n.startswith("conv") or n.startswith("wgrad ")}
        top = max(cands, key=lambda n: cands[n]["ms"])
        tv = cands[top]
        tach = tv["flops"] / (tv["ms"] * 1e-3) / 1e12
        traffic, tsrc = pmc_traffic(top)
        rec["roofline"] = {"bound": "mfma", "kernel": top[top.index("[") + 1:-1] if "[" in top else top.split()[0],
                           "launch": top, "achieved": round(tach, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(tach / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                           "algorithmic_flop": tv["flops"] / tv["n"], "launches": tv["n"],
                           "avg_us": round(tv["ms"] * 1e3 / tv["n"], 2)}
        # the two launches that have carried `roofline` in earlier rounds, both on every line so that rounds stay comparable
        # (round <= 3: the 256 -> 256 @ 14x14 backward-data conv with the fused BatchNorm sums, `T+bnb`; round 4: the forward
        # conv of the same shape with BatchNorm + PReLU in its prologue, `N+bn`)
        rec["roofline_labels"] = {}
        for tag in ("conv N+bn c256+0->256 14x14 k3x3 s1", "conv T+bnb c256+0->256 14x14 k3x3 s1", "conv N c256+0->256 14x14 k3x3 s1"):
            for n_, v_ in prof.items():
                if n_.startswith(tag + " "):
                    a_ = v_["flops"] / (v_["ms"] * 1e-3) / 1e12
                    rec["roofline_labels"][tag.split()[1]] = {"launch": n_, "achieved": round(a_, 2), "frac": round(a_ / peak, 4),
                                                              "launches": v_["n"], "avg_us": round(v_["ms"] * 1e3 / v_["n"], 2)}
        # and the heaviest launch of the OTHER family, so that neither hides behind the other
        for famname, pre in (("roofline_conv", "conv "), ("roofline_wgrad", "wgrad ")):
            sub = {n: v for n, v in prof.items() if n.startswith(pre.strip() if pre == "conv " else pre)}
            if not sub:
                continue
            t2 = max(sub, key=lambda n: sub[n]["ms"])
            v2 = sub[t2]
            a2 = v2["flops"] / (v2["ms"] * 1e-3) / 1e12
            rec[famname] = {"launch": t2, "achieved": round(a2, 2), "frac": round(a2 / peak, 4), "unit": "TFLOP/s",
                            "launches": v2["n"], "avg_us": round(v2["ms"] * 1e3 / v2["n"], 2),
                            "ms_per_step": round(v2["ms"] / prof_steps, 3), "traffic": pmc_traffic(t2)[0]}
        if "conv_wgrad" in fam:
            kw = fam["conv_wgrad"]
            aw = kw["flops"] / (kw["ms"] * 1e-3) / 1e12
            wtraffic, _, wsrc = pmc_family("wgrad") if args.mode == "train" and args.dtype == "bf16" else (None, None, None)
            rec["roofline_family_wgrad"] = {"bound": "mfma", "kernel": "k_wgrad_halo / k_wgrad_fast (+ slab reduce)",
                                            "achieved": round(aw, 2), "peak": peak, "unit": "TFLOP/s",
                                            "frac": round(aw / peak, 4), "traffic": wtraffic, "traffic_source": wsrc,
                                            "launches": kw["n"],
                                            "avg_us": round(kw["ms"] * 1e3 / kw["n"], 2)}
        rec["kernels"] = {name: {"ms_per_step": round(v["ms"] / prof_steps, 3), "launches_per_step": v["n"] // prof_steps,
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                                 "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                          for name, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
    if calib is not None:
        dom = (rec.get("roofline") or {}).get("achieved") if args.mode == "train" else None
        vn, share = normalise(value, calib, rec.get("kernels"), dom)
        sc = [v for v in sclk if v is not None]
        calib.update({"sclk_mhz_avg": round(sum(sc) / len(sc), 1) if sc else None,
                      "dominant_tflops": dom, "ref_dominant_tflops": REF_DOMINANT_TFLOPS, "dominant_exponent": DOM_EXP,
                      "mfma_share_of_kernel_time": round(share, 4),
                      "method": "msml_probe_mfma: 512 WGs x 4 waves x 20000 x 16 register-resident v_mfma_f32_16x16x32_bf16 on random "
                                "operands, median of 15 launches after 12; copy: torch copy_ of 512 MiB (read + write), median of 9; "
                                "msml_probe_mfma_lds: 256 WGs x 8 waves, 9 ds_read_b128 per 14 MFMAs; gemm: msml_gemm_splitk 8192 x 8192 x "
                                "1024, median of 15; all on the training stream right before the timed region -- recorded, NOT used (they do "
                                "not predict the conv families' speed on a box); value_normalised = value x (share x (ref_dominant / "
                                "dominant) ^ dominant_exponent + (1 - share)), dominant = roofline.achieved of this run, share = MFMA "
                                "families' part of the kernel-event time"})
        rec["calibration"] = calib
        rec["value_normalised"] = round(vn, 2)       # AUXILIARY diagnostic across boxes; the headline is `value`
        rec["value_normalised_role"] = "auxiliary (cancels the dominant launch's own speed; exponent validated on held-out lines)"
    if world == 1 and not args.no_extra_modes and args.mode == "train" and args.dtype == "bf16":
        runner = None
        graph = static = out = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        try:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):
                rec["modes"] = extra_modes(args, rank, local_rank)
        except Exception as e:                                  # pragma: no cover (diagnostic path)
            rec["modes"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline and args.mode == "train":
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            rec["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(rec), file=JSON_OUT or sys.stdout, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
