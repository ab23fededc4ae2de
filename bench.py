#!/usr/bin/env python3
"""Headline benchmark: samples/s of the 50-step Karras-Heun sampler, PUNetG 64ch, 1x128x128,
batch 64 per GPU (BASELINE.json configs[1]); synthetic random-init weights and Gaussian noise.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            # self-launching: starts N ranks through torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or under a launcher

One "step" = one full sampling run (50 Heun steps = 99 network evaluations) of one batch per
rank, replayed from the captured hipGraph, followed (N>1) by the RCCL all-gather of the samples.
Rank 0 prints ONE JSON line (schema: task contract + `roofline` + `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA, dense
BF16_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: bf16 / fp16 MFMA, dense; split kernels issue 3 or 6 products per fp32 product
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec


def host_cores():
    """CPU share of this process: cgroup quota if set, else the affinity mask, capped at 16
    (the GPU box gives 16 cores per GPU)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def _stats(ms):
    v = sorted(ms)
    return {"min": round(v[0], 3), "median": round(v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]), 3),
            "max": round(v[-1], 3)}


def gpu_state(dev_index=0):
    """Shader clock (the starred pp_dpm_sclk level, MHz) and board power (hwmon power1_average, W) of the device this rank runs on, read
    from sysfs in Python -- no tool is launched.  None where the box does not show them.  (MI355X_MICROARCH.md, DVFS give-back: the
    in-kernel clock under dense MFMA can sit up to 10 % below pp_dpm_sclk, so this explains box-to-box spread, it does not replace stamps.)"""
    import glob
    out = {"sclk_mhz": None, "power_w": None, "power_cap_w": None}
    try:
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            if not os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                continue
            slot = ""
            try:
                for ln in open(os.path.join(d, "uevent")):
                    if ln.startswith("PCI_SLOT_NAME="):
                        slot = ln.strip().split("=", 1)[1].lower()
            except Exception:
                pass
            cards.append((d, slot))
        pick = next((d for d, slot in cards if want and slot == want), None) or (cards[dev_index][0] if dev_index < len(cards) else None)
        if pick is None:
            return out
        for ln in open(os.path.join(pick, "pp_dpm_sclk")):
            if "*" in ln:
                out["sclk_mhz"] = int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        for h in glob.glob(os.path.join(pick, "hwmon", "hwmon*")):
            for key, name in (("power_w", "power1_average"), ("power_w", "power1_input"), ("power_cap_w", "power1_cap")):
                f = os.path.join(h, name)
                if out[key] is None and os.path.exists(f):
                    try:
                        out[key] = round(int(open(f).read().strip()) / 1e6, 1)
                    except Exception:
                        pass
    except Exception:
        pass
    return out


def conv_flops(B, Cin, Cout, H, W, ks):
    return 2.0 * B * Cout * Cin * ks * ks * H * W


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--nsteps", type=int, default=50, help="Heun steps per sample")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="go through the multi-rank code (RCCL process group, gather, barriers, all-reduce of the time) even with one rank "
                         "(also DIFFSCI_BENCH_FORCE_DIST=1): what one GPU can exercise of the N > 1 path")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true", help="skip the bf16x6 / exact-fp32 legs of the line")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config-3 (ADM-128) and config-5-share legs of the line")
    ap.add_argument("--fuse-max-cot", type=int, default=None, help="fuse norms only in layers with Cout/64 <= this")
    ap.add_argument("--no-up-parity", action="store_true", help="UpSampler convolutions through the generic gather loader")
    ap.add_argument("--no-direct-out", action="store_true", help="output layer on the MFMA kernel (Cout padded to 64)")
    ap.add_argument("--no-fuse-norm", action="store_true", help="standalone norm kernels instead of norms folded into the convolutions")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only run the dominant kernel's launch set (for rocprofv3 --pmc passes)")
    ap.add_argument("--precision", default="fp16x3", choices=["fp16x3", "bf16x6", "fp32"],
                    help="3x3 conv arithmetic: fp32 emulated on the 16-bit matrix cores (fp16 hi/lo split, 3 products; "
                         "bf16 3-way split, 6 products) or exact-fp32 MFMA")
    return ap.parse_args()


def build_module(args, dev):
    import diffsci_amd.models as M
    # random-init weights of the architecture: the network's own containers carry the reference's default
    # initialisers (Kaiming-uniform convolutions / linears, unit norms, Fourier W ~ N(0, 30^2)); nothing from oracle/
    torch.manual_seed(0)
    net = M.PUNetG(M.PUNetGConfig(model_channels=args.channels))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}      # CPU copy for the cpu_baseline leg
    cfg = dict(model_channels=args.channels)
    net.conv_precision = args.precision
    net.fuse_norm = not args.no_fuse_norm
    net.direct_out = not args.no_direct_out
    net.upsample_parity = not args.no_up_parity
    if args.fuse_max_cot is not None:
        net.fuse_max_cot = args.fuse_max_cot
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
    module.use_graph = not args.no_graph
    return module, sd, cfg


def dominant_kernel_roofline(module, args, dev, reps=40):
    """Average launch duration of the dominant kernel (k_conv3h<PLAIN, *>: the 3x3 convolutions of
    the residual blocks and convin; convout too when it is not on the direct kernel) over one network evaluation's worth of its launches,
    timed with events on the launch stream, against its algorithmic FLOPs.  The launches carry what
    they carry in the network: conv1 = fused norm+SiLU loader + time shift + tile statistics,
    conv2 = fused norm+SiLU loader + residual + tile statistics (the fused parts only with
    fuse_norm, i.e. the default fp16x3 path)."""
    from diffsci_amd import ops
    net = module.model
    pk = net.packed_weights()
    B, S = args.batch, args.size
    mods = [net.convin] + [c for blk in net._resblocks() for c in (blk.conv1, blk.conv2)]
    if not (net.convout.out_channels <= 4 and getattr(net, "direct_out", True)):
        mods.append(net.convout)              # otherwise the output layer runs on ds_conv2d_direct, not on this kernel
    # spatial size of every launch, in forward order
    mult = net.config.extended_channel_expansion
    def side(c):
        return S // (c // args.channels) if c >= args.channels else S
    launches = []
    for m in mods:
        cin, cout = m.in_channels, m.out_channels
        s = side(max(cin, cout)) if m not in (net.convin, net.convout) else S
        launches.append((m, cin, cout, s))
    del mult
    bufs = {}
    def buf(c, s, tag="in"):
        k = (c, s, tag)
        if k not in bufs:
            bufs[k] = torch.randn(B, c, s, s, device=dev)
        return bufs[k]
    outs = {(m.out_channels, s): torch.empty(B, m.out_channels, s, s, device=dev) for m, _, _, s in launches}
    flops = sum(conv_flops(B, cin, cout, s, s, 3) for _, cin, cout, s in launches)
    conv1s = {id(blk.conv1) for blk in net._resblocks()}
    conv2s = {id(blk.conv2) for blk in net._resblocks()}
    shift = {c: torch.randn(1, c, device=dev) for c in {m.out_channels for m in mods}}
    fused = net._fused()
    blk_of = {id(c): blk for blk in net._resblocks() for c in (blk.conv1, blk.conv2)}
    k1, k2 = net.norm_kinds
    imgs = {}
    tabs, stats = {}, {}
    if fused:
        for m, cin, cout, s in launches:
            if (cin, s) not in tabs:
                t = torch.zeros(B, ops.table_channels(cin), 4, device=dev)
                t[:, :cin, 1] = 1.0                       # (M, A, C) = (0, 1, 0): the loader computes SiLU(x) ...
                t[:, :, 3] = 2.0 ** -10                   # ... times 2^10, the exponent a bound of 8 on |x| asks for
                tabs[(cin, s)] = t
            stats[(cout, s)] = torch.empty(B, cout, ops.conv_tile_count(s, s), 4, device=dev)

    def use_images(m, cin, s):      # the standalone-norm blocks hand their convolutions pre-split images (punetg._res)
        blk = blk_of.get(id(m))
        folded = fused and (cin + 63) // 64 <= net.fuse_max_cot
        return blk is not None and not folded and net._norm_images_ok(blk, cin, s, s, k1, k2)

    for m, cin, cout, s in launches:
        if use_images(m, cin, s) and (cin, s) not in imgs:
            ones = torch.ones(cin, device=dev)
            imgs[(cin, s)] = ops.inorm_silu_images(buf(cin, s), ones, torch.zeros(cin, device=dev), 0)

    # activation exponents as in the network (fp16x3): the input layer reads the row a reduction over c_in * x left, the folded
    # launches the bound their table call left, and the blocks whose result feeds a Down / UpSampler or the attention (the last
    # block of a group: 5 of the 28 block launches) record max |out| in their epilogue
    h3 = net.conv_precision == "fp16x3"
    amax_kw = {}
    groups_all = [list(g) for g in net.downward_blocks] + [list(net.attn_resnet_block), list(net.after_block)] + [list(g) for g in net.upward_blocks]
    with_out_stats_off = {id(g[-1].conv2) for g in groups_all if len(g)}
    if h3:
        row_in = ops.absmax_rows(buf(mods[0].in_channels, S))
        row_out = ops.amax_new(B, dev)
        groups = [list(g) for g in net.downward_blocks] + [list(net.attn_resnet_block), list(net.after_block)] + [list(g) for g in net.upward_blocks][:-1]
        with_out = {id(g[-1].conv2) for g in groups if len(g)}
        for m, cin, cout, s in launches:
            block = (id(m) in conv1s or id(m) in conv2s) and (cin + 63) // 64 <= net.fuse_max_cot
            kw = {}
            if m is net.convin:
                kw["in_amax"] = row_in
            elif fused and block:
                pass                                                         # the table's fourth column carries the exponent
            elif m is not net.convout:
                kw["in_amax"] = ops.NORMALISED                               # a standalone norm's output
            if id(m) in with_out:
                kw["out_amax"] = row_out
            amax_kw[id(m)] = kw

    def run():      # the same loaders / epilogues as in the network
        for m, cin, cout, s in launches:
            block = (id(m) in conv1s or id(m) in conv2s) and (cin + 63) // 64 <= net.fuse_max_cot
            akw = amax_kw.get(id(m), {})
            if use_images(m, cin, s):
                ops.conv_img(imgs[(cin, s)], pk[id(m)], B, cin, s, s, bias=m.bias,
                             shift=shift[cout] if id(m) in conv1s else None,
                             res1=buf(cout, s, "res") if id(m) in conv2s else None,
                             out=outs[(cout, s)],                   # no tile statistics: the image route's norms compute their own
                             **({"out_amax": akw["out_amax"]} if "out_amax" in akw else {}))
                continue
            ops.conv(buf(cin, s), pk[id(m)], bias=m.bias,
                     shift=shift[cout] if id(m) in conv1s else None,
                     res1=buf(cout, s, "res") if id(m) in conv2s else None,
                     prenorm=tabs[(cin, s)] if (fused and block) else None,
                     # statistics where the network asks for them: conv1 of a folded block (conv2's table reads them), conv2 unless its
                     # result feeds a Down / UpSampler or the attention, the input layer
                     tile_stats=stats[(cout, s)] if (fused and m is not net.convout and (m is net.convin or block)
                                                     and id(m) not in with_out_stats_off) else None,
                     out=outs[(cout, s)], **akw)
    run()
    torch.cuda.synchronize()
    # default reps: ~0.3-1 s of sustained launches (short bursts run at a higher clock than the real loop)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        run()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n = len(launches)
    kname, peak = {
        "fp16x3": ("k_conv3p / k_conv3h<PLAIN> (ds_conv2d_h3 / ds_conv2d_h3_img, 3x3, fp32 via 3 fp16 MFMA products; all launches but the input layer as persistent producer / consumer workgroups"
                   + (", norm+SiLU in the loader or pre-split image input, tile statistics in the epilogue)" if fused else ")"), BF16_PEAK_TFLOPS / 3.0),
        "bf16x6": ("k_conv6<PLAIN> (ds_conv2d_x6, 3x3, fp32 via 6 bf16 MFMA products)", BF16_PEAK_TFLOPS / 6.0),
        "fp32": ("k_conv<3,PLAIN> (ds_conv2d 3x3, exact-fp32 MFMA)", MFMA_F32_PEAK_TFLOPS)}[net.conv_precision]
    achieved = flops / (ms * 1e-3) / 1e12
    traffic = None
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"{r}_dominant_kernel_traffic.json") for r in ("r04", "r03", "r02", "r01"))
                  if os.path.exists(q)), "")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("precision") == net.conv_precision:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2),
            "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": traffic, "launches_per_eval": n, "avg_launch_ms": round(ms / n, 4),
            "flop_per_launch_avg": flops / n}


def hbm_class(module, args, dev):
    """Achieved HBM GB/s of the two HBM-bound kernel classes at this workload's shapes."""
    from diffsci_amd import ops
    from diffsci_amd._native import EvalCoef
    out = {}
    B, S, C = args.batch, args.size, args.channels
    x = torch.randn(B, C, S, S, device=dev)
    y = torch.empty_like(x)
    w = torch.ones(C, device=dev)
    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = timed(lambda: ops.inorm_silu(x, w, w, 0, out=y))
    gbs = x.numel() * 8 / (ms * 1e-3) / 1e9
    out["inorm_silu_L0"] = {"bytes_per_elt": 8, "GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                            "ms": round(ms, 4)}
    k = EvalCoef(c_out=0.4, c_skip=0.1, sigma_sq=2.0, neg_mult=-1.4, neg_lang=0.0, guidance=1.0,
                 one_minus_guidance=0.0, input_kind=0, stochastic=0)

    def timed_in_graph(fn, launches=200, reps=5):
        """The sampler replays its step kernels from a hipGraph; a 4 MiB step is a ~5 us kernel, shorter than one
        eager launch through ctypes takes to issue.  Capture `launches` back-to-back calls of fn and time the replay;
        returns ms per call of fn."""
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn(); side.synchronize()
            with ops.Graph() as g:
                for _ in range(launches):
                    fn()
            g.launch(); side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                g.launch()
            e1.record(side); side.synchronize()
        torch.cuda.current_stream(dev).wait_stream(side)
        return e0.elapsed_time(e1) / (reps * launches)

    # The fused Karras-Heun step = Euler predictor (R x, R F1, W c_in*x_e: 12 B/elt) + Heun corrector (R x, R F1, R F2,
    # W x', W c_in*x': 20 B/elt) = 32 algorithmic bytes per element per step.  Timed as the sampler runs
    # it: replayed from a hipGraph, back to back.  At config 2 the tensors are 4 MiB each: the whole working set
    # (24 MiB) sits in L2 / Infinity Cache, so that figure is a CACHE rate (the kernel pair is launch-latency bound:
    # ~1.5 us of dependent-launch boundary per kernel) and is labelled so; the HBM figure is the 256 MiB case, whose
    # working set (1.5 GiB) cannot be cache resident.
    for label, n, resident in (("heun_step_cfg2_4MiB", B * S * S, True), ("heun_step_256MiB", B * C * S * S, False)):
        xs, f1, f2 = (torch.randn(n, device=dev) for _ in range(3))
        xo, xi = torch.empty(n, device=dev), torch.empty(n, device=dev)

        def step():
            ops.euler(xs, f1, k, -0.5, x_out=None, xin_out=xi, c_in_next=0.3)              # predictor: only c_in*x_e is stored
            ops.heun(xs, f1, k, f2, k, -0.5, x_out=xo, xin_out=xi, c_in_next=0.3)          # corrector

        def corrector():
            ops.heun(xs, f1, k, f2, k, -0.5, x_out=xo, xin_out=xi, c_in_next=0.3)
        launches = 100 if resident else 10
        ms = timed_in_graph(step, launches=launches) * 1.0          # per (predictor + corrector) pair
        ms_c = timed_in_graph(corrector, launches=launches)
        gbs = n * 32 / (ms * 1e-3) / 1e9
        out[label] = {"bytes_per_elt": 32, "GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                      "ms": round(ms, 5), "cache_resident": resident,
                      "timed": f"hipGraph replay of {launches} (Euler predictor + Heun corrector) pairs",
                      "corrector_only": {"bytes_per_elt": 20, "ms": round(ms_c, 5),
                                         "GB/s": round(n * 20 / (ms_c * 1e-3) / 1e9, 1)}}
        del xs, f1, f2, xo, xi
    # ADM's per-sample GroupNorm(1, C) pair at config 3's level-0 size ([32, 128, 256, 256] = 1 GiB): statistics pass
    # (4 B/elt) + apply pass (8 B/elt) = 12 B/elt
    try:
        xa = torch.randn(32, 128, 256, 256, device=dev)
        ya = torch.empty_like(xa)
        wa = torch.ones(128, device=dev)
        stats = torch.empty(32, 2, device=dev)
        wsb = torch.empty(N_gnorm_ws(32) // 4, device=dev)
        ms_s = timed(lambda: ops.gnorm1_stats(xa, 0, stats=stats, workspace=wsb), reps=10)
        ms_a = timed(lambda: ops.gnorm1_apply(xa, stats, wa, wa, 0, out=ya), reps=10)
        ne = xa.numel()
        out["adm_group1_norm_cfg3_1GiB"] = {
            "stats": {"bytes_per_elt": 4, "ms": round(ms_s, 4), "GB/s": round(ne * 4 / (ms_s * 1e-3) / 1e9, 1)},
            "apply": {"bytes_per_elt": 8, "ms": round(ms_a, 4), "GB/s": round(ne * 8 / (ms_a * 1e-3) / 1e9, 1)},
            "bytes_per_elt": 12, "GB/s": round(ne * 12 / ((ms_s + ms_a) * 1e-3) / 1e9, 1),
            "frac_hbm_peak": round(ne * 12 / ((ms_s + ms_a) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "cache_resident": False}
        del xa, ya
    except Exception as e:                                   # reporting only: never fail the bench line on it
        out["adm_group1_norm_cfg3_1GiB"] = {"error": str(e)[:200]}
    return out


def N_gnorm_ws(B):
    from diffsci_amd import _native as N
    return N.lib().ds_gnorm1_workspace_bytes(B)


def other_precisions(args, dev, wn):
    """The same workload on the two other convolution arithmetics (one timed batch each): the exact 3-way bf16 split
    (6 MFMA products, no range limit -- what the range guard falls back to) and the exact-fp32 MFMA."""
    import copy
    out = {}
    for prec in ("bf16x6", "fp32"):
        a = copy.copy(args)
        a.precision = prec
        module, _, _ = build_module(a, dev)
        module.propagate_white_noise(wn, nsteps=a.nsteps)            # eager pass + capture + first replay
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        module.propagate_white_noise(wn, nsteps=a.nsteps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[prec] = {"samples/s": round(wn.shape[0] / dt, 3), "ms_per_step": round(dt * 1e3, 1)}
        del module
        torch.cuda.empty_cache()
    return out


class _FlopCounter:
    """Algorithmic FLOPs of one network evaluation: hooks on the convolution / attention entry points of diffsci_amd.ops during
    ONE eager forward (true channel counts, no padding; the parity kernels are counted as the 3x3 convolution they replace)."""

    def __init__(self, ops):
        self.ops, self.flops, self.by = ops, 0.0, {}
        self.saved = {}

    def _add(self, kind, f):
        self.flops += f
        self.by[kind] = self.by.get(kind, 0.0) + f

    def __enter__(self):
        ops, me = self.ops, self

        def conv2d(x, w_packed, Cout, ks, *a, **k):
            out = me.saved["conv2d"](x, w_packed, Cout, ks, *a, **k)
            me._add("conv3x3" if ks == 3 else "conv1x1", 2.0 * out.shape[0] * Cout * x.shape[1] * ks * ks * out.shape[2] * out.shape[3])
            return out

        def conv_img(images, pw, B, Cin, H, W, *a, **k):
            me._add("conv3x3", 2.0 * B * pw.Cout * Cin * 9 * H * W)
            return me.saved["conv_img"](images, pw, B, Cin, H, W, *a, **k)

        def conv_up_img(images, pw, B, Cin, Hl, Wl, *a, **k):
            me._add("conv3x3", 2.0 * B * pw.Cout * Cin * 9 * 4 * Hl * Wl)
            return me.saved["conv_up_img"](images, pw, B, Cin, Hl, Wl, *a, **k)

        def conv_direct(x, w, *a, **k):
            me._add("conv3x3", 2.0 * x.shape[0] * w.shape[0] * x.shape[1] * 9 * x.shape[2] * x.shape[3])
            return me.saved["conv_direct"](x, w, *a, **k)

        def attention(qkv, E, *a, **k):
            me._add("attention", 4.0 * qkv.shape[0] * qkv.shape[2] * qkv.shape[2] * E)
            return me.saved["attention"](qkv, E, *a, **k)

        for name, fn in (("conv2d", conv2d), ("conv_img", conv_img), ("conv_up_img", conv_up_img), ("conv_direct", conv_direct),
                         ("attention", attention)):
            self.saved[name] = getattr(ops, name)
            setattr(ops, name, fn)
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.ops, name, fn)
        return False


def adm_conv_roofline(net, B, S, dev, reps=6):
    """Event-timed launch set of config 3's dominant kernel family (the fp16x3 3x3 convolutions): conv2 of every residual block --
    Cout -> Cout at the block's output resolution, with the loader the network uses at that width (norm + SiLU folded into the
    loader up to fuse_max_cot channel tiles, plain above), tile statistics where folded -- against its algorithmic FLOPs."""
    from diffsci_amd import ops
    pk = net.packed_weights()
    launches, side = [], S
    mult = net.config.extended_channel_expansion
    nb_d, nb_u = net.config.number_resnet_downward_block, net.config.number_resnet_upward_block
    sides = []
    for i in range(len(mult) - 1):                               # encoder layer i: nb - 1 blocks at `side`, the last one halves it
        sides += [side] * (nb_d - 1) + [side // 2]
        side //= 2
    sides += [side] * net.config.num_blocks_middle_block
    for i in range(len(mult) - 1):                               # decoder layer: the last block doubles
        sides += [side] * (nb_u - 1) + [side * 2]
        side *= 2
    blocks = list(net._blocks())
    assert len(blocks) == len(sides)
    bufs, tabs, stats, outs = {}, {}, {}, {}
    for blk, sd_ in zip(blocks, sides):
        co = blk.conv2.out_channels
        fold = net._fused() and (co + 63) // 64 <= net.fuse_max_cot
        key = (co, sd_)
        if key not in bufs:
            bufs[key] = torch.randn(B, co, sd_, sd_, device=dev)
            outs[key] = torch.empty(B, co, sd_, sd_, device=dev)
            t = torch.zeros(B, ops.table_channels(co), 4, device=dev)
            t[:, :co, 1] = 1.0
            t[:, :, 3] = 2.0 ** -10
            tabs[key] = t
            stats[key] = torch.empty(B, co, ops.conv_tile_count(sd_, sd_), 4, device=dev)
        launches.append((blk.conv2, co, sd_, fold))
    flops = sum(conv_flops(B, co, co, sd_, sd_, 3) for _, co, sd_, _ in launches)

    def run():
        for m, co, sd_, fold in launches:
            k = (co, sd_)
            ops.conv(bufs[k], pk[id(m)], bias=m.bias, prenorm=tabs[k] if fold else None, tile_stats=stats[k] if fold else None,
                     in_amax=None if fold else ops.NORMALISED, out=outs[k])
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ach, peak = flops / (ms * 1e-3) / 1e12, BF16_PEAK_TFLOPS / 3.0
    return {"bound": "mfma", "kernel": "k_conv3p / k_conv3h<PLAIN> (ds_conv2d_h3: conv2 of every ADM residual block, folded norm + SiLU loader where the network folds; the persistent form wherever the shape is its own)",
            "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
            "launches": len(launches), "avg_launch_ms": round(ms / len(launches), 4)}


def other_configs(dev):
    """BASELINE.json configs[2] and configs[4] (one GPU's share) on this box, after the timed region: one capture run and ONE
    timed run each (the driver sees them next to the headline; tools/bench_adm.py / bench_cfg5.py are the stand-alone forms).
    `eval_equiv` prices the WHOLE evaluation -- norms, attention, resampling, step kernels included -- against the fp16x3
    convolution peak with the network's algorithmic FLOPs: a lower bound on its dominant kernel's own fraction."""
    import diffsci_amd.models as M
    from diffsci_amd import ops
    out = {}

    REPLAYS = 3

    def timed(module, wn, evals, **kw):
        t0 = time.perf_counter()
        o = module.propagate_white_noise(wn, **kw)                  # eager pass + capture + first replay
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        dts = []
        for _ in range(REPLAYS):
            t0 = time.perf_counter()
            o = module.propagate_white_noise(wn, **kw)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        assert bool(torch.isfinite(o).all())
        dts.sort()
        return dts[len(dts) // 2], first, dts

    def price(net, x, t, y, dt, evals, per_eval_batch):
        with torch.inference_mode(), _FlopCounter(ops) as fc:
            net(x, t, y) if y is not None else net(x, t)
        tf = fc.flops * (per_eval_batch / x.shape[0]) / 1e12          # TFLOP per evaluation of the sampler's batch
        ach = tf / (dt / evals)
        return {"algorithmic_TFLOP_per_eval": round(tf, 3), "achieved_TFLOPs_equiv": round(ach, 1), "peak": round(BF16_PEAK_TFLOPS / 3.0, 1),
                "frac": round(ach / (BF16_PEAK_TFLOPS / 3.0), 4), "flop_share": {k: round(v / fc.flops, 3) for k, v in fc.by.items()}}

    try:
        torch.manual_seed(0)
        c, B, S, N = 128, 32, 256, 50
        net = M.ADM(M.ADMConfig(input_channels=3, output_channels=3, model_channels=c, time_embed_dim=c, output_embed_dim=4 * c,
                                channel_expansion=[1, 2, 4, 4], skip_integration_type="concat"))
        module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
        wn = torch.randn(B, 3, S, S, device=dev)
        evals = 2 * N - 1
        dt, first, dts = timed(module, wn, evals, nsteps=N, integrator="karras")
        out["config3_adm128"] = {
            "workload": f"ADM-{c} (concat skips, attention at 16^2), [{B},3,{S},{S}], {N}-step sigma-churn (KarrasIntegrator, noise generated in the kernels)",
            "samples/s": round(B / dt, 3), "ms_per_eval": round(1e3 * dt / evals, 2), "first_call_s": round(first, 2),
            "timed_replays": REPLAYS, "ms_per_run": _stats([1e3 * d for d in dts]), "roofline": adm_conv_roofline(net, B, S, dev),
            "dominant_kernel": "k_conv3h / k_convup (fp16x3 3x3 convolutions: folded norm loader, image input, parity upsampling)",
            "eval_equiv": price(net, wn[:2], torch.tensor([0.5, 1.0], device=dev), None, dt, evals, B)}
        del module, net, wn
        torch.cuda.empty_cache()
    except Exception as e:                                   # reporting only: never fail the bench line on it
        out["config3_adm128"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    try:
        torch.manual_seed(0)
        B, S, N, g = 16, 256, 100, 2.0
        net = M.PUNetG(M.PUNetGConfig(input_channels=4, output_channels=4), conditional_embedding=M.nets.PorosityEmbedder(dembed=64))
        module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev).eval()
        wn = torch.randn(B, 4, S, S, device=dev)
        y = {"porosity": torch.tensor([0.2], device=dev)}
        evals = 2 * N - 1                                                # guided evaluations: each is one launch set on batch 2B
        dt, first, dts = timed(module, wn, evals, y=y, guidance=g, nsteps=N)
        a5 = argparse.Namespace(batch=2 * B, size=S, channels=64)
        out["config5_share_cond_punetg64"] = {
            "timed_replays": REPLAYS, "ms_per_run": _stats([1e3 * d for d in dts]),
            "roofline": dominant_kernel_roofline(module, a5, dev, reps=6),
            "workload": f"conditional PUNetG-64 (PorosityEmbedder), [{B},4,{S},{S}] per GPU, classifier-free guidance {g} "
                        f"(conditional + unconditional evaluation as one of batch {2 * B}), {N}-step Heun",
            "samples/s": round(B / dt, 3), "ms_per_eval": round(1e3 * dt / evals, 2), "network_calls": 2 * evals,
            "first_call_s": round(first, 2),
            "dominant_kernel": "k_conv3h (fp16x3 3x3 convolutions) -- attention at L = 4096 (k_attn3h) is ~10 %",
            "eval_equiv": price(net, wn[:2], torch.tensor([0.5, 1.0], device=dev), {"porosity": torch.tensor([0.2], device=dev)},
                                dt, evals, 2 * B)}
        del module, net, wn
        torch.cuda.empty_cache()
    except Exception as e:
        out["config5_share_cond_punetg64"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def cpu_baseline(sd, cfg, args):
    """The CPU oracle (a port: the reference itself cannot travel) on this host's cores, on a
    bounded sample of the same workload."""
    from oracle import karras_ref as K
    from oracle import punetg_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline on {cores} host threads ...", file=sys.stderr, flush=True)
    B, N = 8, 8                                   # 8-step Heun = 15 network evaluations (~10-20 s)
    g = torch.Generator().manual_seed(1)
    wn = torch.randn(B, 1, args.size, args.size, generator=g)
    net = punetg_ref.make_net(sd, punetg_ref.default_config(**cfg))
    with torch.inference_mode():
        K.propagate_white_noise(net, wn[:1], 2)    # warm-up (3 evaluations of one sample)
        t0 = time.time()
        K.propagate_white_noise(net, wn, N)
        dt = time.time() - t0
    evals = 2 * N - 1
    full = 2 * args.nsteps - 1
    sps = B / (dt * full / evals)
    return {"value": round(sps, 5), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU restatement), batch {B}, {N}-step Heun = {evals} evaluations in {dt:.1f} s, "
                      f"scaled by {full}/{evals} evaluations to the {args.nsteps}-step workload"}


def launcher_command(argv, n, port, script=None):
    """The command `python bench.py --gpus N` runs for itself when no launcher started it: one rank per GPU through
    torch.distributed.run on this node (the reference's multi-GPU sampler spawns its own workers too,
    stochasticity_paper/scripts/test-diffusion-cifar10karras-colormap-parallel.py:191-291)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv=None, script=None):
    """Parent of a multi-rank run started without a launcher.  Runs BEFORE anything touches the GPU in this process
    (a process that has initialised HIP must not be replaced or forked into ranks); relays the children's output
    (rank 0 prints the JSON line) and exits with their status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // max(1, args.gpus))))
    cmd = launcher_command(sys.argv[1:] if argv is None else argv, args.gpus, port, script)
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    # DIFFSCI_BENCH_SHARE_GPU=1 (rehearsals on a one-GPU box): the ranks share the visible devices round-robin and rendezvous /
    # gather over gloo -- RCCL refuses two ranks on one device.  Never set by the driver; the line then reports the backend.
    share = os.environ.get("DIFFSCI_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force = args.force_dist or os.environ.get("DIFFSCI_BENCH_FORCE_DIST") == "1"
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "WORLD_SIZE" not in os.environ:                 # one rank, no launcher: a rendezvous of our own
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    from diffsci_amd.parallel import gather_samples, global_white_noise, shard_rows
    module, sd, cfg = build_module(args, dev)
    if args.roofline_only:
        print(json.dumps(dominant_kernel_roofline(module, args, dev, reps=2)), flush=True)
        return
    shape = [1, args.size, args.size]
    B = args.batch
    lo, hi = shard_rows(B * world, world, rank)
    # inputs resident in HBM before the timed region: this rank's rows of the global noise tensor
    t_noise = time.perf_counter()
    noise = [global_white_noise(B * world, shape, seed=s, rows=(lo, hi)).to(dev)
             for s in range(args.steps + args.warmup)]
    torch.cuda.synchronize()
    noise_draw_ms = (time.perf_counter() - t_noise) * 1e3 / (args.steps + args.warmup)    # host draw of this rank's rows + upload, per step
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]   # step start, gather start, step end

    def one_step(i, timed=False):
        if timed:
            ev[i][0].record()
        out = module.propagate_white_noise(noise[i], nsteps=args.nsteps)
        if timed:
            ev[i][1].record()
        out = gather_samples(out) if dist is not None else out
        if timed:
            ev[i][2].record()
        return out

    for i in range(args.warmup):
        one_step(args.steps + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    state0 = gpu_state(local_rank)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step(i, timed=True)
    torch.cuda.synchronize()
    state1 = gpu_state(local_rank)                      # right behind the last kernel: the clock / power the loop ran at
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_rank = dt
    step_ms = [ev[i][0].elapsed_time(ev[i][2]) for i in range(args.steps)]        # device time per step, events on the launch stream
    gather_ms = [ev[i][1].elapsed_time(ev[i][2]) for i in range(args.steps)]
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([dt_rank * 1e3 / args.steps, sum(gather_ms) / len(gather_ms), noise_draw_ms], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(v), 3) for v in r.tolist()] for r in allr]
    assert out.shape[0] == B * world and bool(torch.isfinite(out).all())
    if rank == 0:
        value = B * world * args.steps / dt
        line = {
            "metric": "samples/sec (50-step Karras Heun), PUNetG 64ch 128x128",
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            # device time of each timed step (events on the launch stream): the spread inside one run, next to the clock / power the
            # box held -- what separates a slower build from a slower box
            "ms_per_step_min": _stats(step_ms)["min"], "ms_per_step_median": _stats(step_ms)["median"], "ms_per_step_max": _stats(step_ms)["max"],
            "gpu_state": {"before": state0, "after": state1},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x6": "f32 (3x3 convs: exact 3-way bf16 split, 6 bf16 MFMA products, fp32 accumulate)",
                      "fp16x3": "f32 (3x3 convs: fp16 hi+lo split, 3 fp16 MFMA products, fp32 accumulate)"}[args.precision],
            "data": "synthetic (random-init weights, Gaussian noise)",
            "config": {"workload": f"PUNetG {args.channels}-base-ch, 1x{args.size}x{args.size} fields, batch {B} per GPU, "
                                   f"{args.nsteps}-step Heun deterministic sampler ({2*args.nsteps-1} network evaluations)",
                       "global_batch": B * world, "parallelism": f"dp{world} (batch shards, all-gather of samples)",
                       "world_size": dist.get_world_size() if dist is not None else 1,
                       "backend": ("gloo (rehearsal)" if share else "nccl (RCCL)") if dist is not None else "none (one process)",
                       **({"rehearsal": "ranks share one GPU, gloo instead of RCCL"} if share else {}),
                       "hipgraph": not args.no_graph},
        }
        if dist is not None:
            # what a scaling curve will be read from: per rank [ms per step, all-gather ms per step (events), host noise draw + upload ms per step]
            line["multi_gpu"] = {"per_rank_ms_per_step": [r[0] for r in per_rank], "gather_ms": [r[1] for r in per_rank],
                                 "noise_draw_ms": [r[2] for r in per_rank],
                                 "gather_bytes_per_rank": int(B * args.size * args.size * 4)}
        print(f"[bench] {value:.3f} samples/s, {dt / args.steps * 1e3:.1f} ms per {B}-sample batch", file=sys.stderr, flush=True)
        line["roofline"] = dominant_kernel_roofline(module, args, dev)
        print(f"[bench] roofline {line['roofline']}", file=sys.stderr, flush=True)
        line["roofline_hbm_class"] = hbm_class(module, args, dev)
        print(f"[bench] hbm class {line['roofline_hbm_class']}", file=sys.stderr, flush=True)
        if world == 1 and not args.no_other_precisions and args.precision == "fp16x3":
            line["other_precisions"] = other_precisions(args, dev, noise[0])
            print(f"[bench] other precisions {line['other_precisions']}", file=sys.stderr, flush=True)
        if world == 1 and not args.no_other_configs and args.precision == "fp16x3":
            del module
            torch.cuda.empty_cache()
            line["other_configs"] = other_configs(dev)
            print(f"[bench] other configs {line['other_configs']}", file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(sd, cfg, args)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
