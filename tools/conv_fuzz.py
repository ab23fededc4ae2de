"""Randomised cross-check of the fp16x3 convolution family (3x3 with every load mode, padding mode, fused
loader and tile statistics; 1x1 with its load modes; direct output layer) against fp64 torch on random shapes.
Raw-input launches draw a whole-problem scale 2^-k (input, bias, shift and residual alike: nothing of order one to hide
behind), k in {0, 0, 8, 16, 24, 40, -20}: the per-sample activation exponents keep the relative error where it is at k = 0.

    python tools/conv_fuzz.py [--n 200] [--seed 0]
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from diffsci_amd import ops


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


SCALES = [0, 0, 8, 16, 24, 40, -20]


def fuzz_up(ri, g, dev):
    """ds_conv2d_h3_up: whole-tile low-resolution shapes, every epilogue / loader option."""
    B, Cin, Cout = ri(1, 3), ri(1, 70), ri(1, 140)
    if ri(0, 1):
        Hl, Wl = 8 * ri(1, 4), 32 * ri(1, 2)
    else:
        Hl, Wl = 16 * ri(1, 2), 16 * ri(1, 3)
    assert ops.N.lib().ds_conv2d_h3_up_supported(Hl, Wl)
    H, W = 2 * Hl, 2 * Wl
    circ, pre, res = ri(0, 2) == 0, ri(0, 1) == 1, ri(0, 2)
    sc = 1.0 if pre else 2.0 ** -SCALES[ri(0, len(SCALES) - 1)]
    x = (torch.randn(B, Cin, Hl, Wl, generator=g) * 2 + 0.3) * sc
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias, shift = torch.randn(Cout, generator=g) * sc, torch.randn(B, Cout, generator=g) * sc
    r1 = None if res == 0 else torch.randn(B, Cout, *((H, W) if res == 1 else (Hl, Wl)), generator=g) * sc
    xin = x.double()
    tab = None
    if pre:
        M, A, C = torch.randn(B, Cin, generator=g) * 0.3, torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.3
        tab = torch.zeros(B, ops.table_channels(Cin), 4)
        tab[:, :Cin, 0], tab[:, :Cin, 1], tab[:, :Cin, 2] = M, A, C
        if ri(0, 1):
            tab[:, :, 3] = 2.0 ** -ri(-4, 11)                                     # an activation exponent: exact, the result does not change
        xin = F.silu((xin - M.double()[..., None, None]) * A.double()[..., None, None] + C.double()[..., None, None])
    up = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if circ:
        up = F.pad(F.pad(up, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1), mode="circular")
        want = F.conv2d(up, w.double(), bias.double())
    else:
        want = F.conv2d(up, w.double(), bias.double(), padding=1)
    want = want + shift.double()[..., None, None]
    if r1 is not None:
        want = want + (r1.double() if res == 1 else F.interpolate(r1.double(), scale_factor=2.0, mode="nearest"))
    ts = torch.full((B, Cout, ops.conv_tile_count(H, W), 4), float("nan"), device=dev)
    pw = ops.pack_conv(w.to(dev), "fp16x3", upsampled=True)
    got = ops.conv(x.to(dev), pw, bias=bias.to(dev), shift=shift.to(dev), res1=None if r1 is None else r1.to(dev),
                   res1_upsampled=res == 2, load_mode=2, circular=circ, prenorm=None if tab is None else tab.to(dev),
                   tile_stats=ts).cpu()
    e = rel(got, want)
    assert torch.isfinite(ts).all(), "tile statistics not fully written"
    K, S, Q, n = ts.cpu().double().unbind(-1)
    assert torch.equal(n.sum(-1), torch.full_like(n.sum(-1), H * W)), "tile pixel counts"
    sxx = (Q + 2 * K * S + n * K * K).sum(-1)
    return max(e, float(((sxx - (want * want).sum(dim=(2, 3))).abs() / (want * want).sum(dim=(2, 3)).clamp_min(1e-300)).max()))


def fuzz_images(ri, g, dev):
    """The image route: ds_gnorm1_apply_images / ds_inorm_silu_images, then ds_conv2d_h3_img or ds_conv2d_h3_up_img -- bit-identical
    to the fp32 route (norm kernel, then the convolution splitting in its loader) and close to fp64."""
    B, Cout = ri(1, 3), ri(1, 140)
    up = ri(0, 2) == 0
    Cin = ri(1, 70) if up else 32 * ri(1, 3) - ri(0, 15)            # plain kernel: an even number of 16-channel chunks
    if up:
        H, W = (8 * ri(1, 3), 32 * ri(1, 2)) if ri(0, 1) else (16 * ri(1, 2), 16 * ri(1, 2))
    else:
        H, W = ri(1, 24), ri(1, 40)
    adm = ri(0, 1) == 1 or (H * W) % 4 != 0 or H * W > 4096          # group-1 norm (any shape) or per-channel norm
    kind = ri(0, 1)
    pool = adm and not up and ri(0, 2) == 0
    Hx, Wx = (2 * H, 2 * W) if pool else (H, W)
    x = (torch.randn(B, Cin, Hx, Wx, generator=g) * 1.5 + 0.2).to(dev)
    wn, bn = torch.randn(Cin, generator=g).to(dev), torch.randn(Cin, generator=g).to(dev)
    film = (torch.randn(B, 2 * Cin, generator=g) * 0.3).to(dev) if adm and ri(0, 1) else None
    if adm:
        st = ops.gnorm1_stats(x, kind, eps=1e-5)
        act = ops.gnorm1_apply(x, st, wn, bn, kind, pool=pool, film=film)
        img = ops.gnorm1_apply_images(x, st, wn, bn, kind, pool=pool, film=film)
    else:
        act = ops.inorm_silu(x, wn, bn, kind=kind)
        img = ops.inorm_silu_images(x, wn, bn, kind)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias, shift = torch.randn(Cout, generator=g).to(dev), torch.randn(B, Cout, generator=g).to(dev)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    res_up = not up and Ho % 2 == 0 and Wo % 2 == 0 and ri(0, 2) == 0
    r1 = torch.randn(B, Cout, *((Ho // 2, Wo // 2) if res_up else (Ho, Wo)), generator=g).to(dev)
    pw = ops.pack_conv(w.to(dev), "fp16x3", upsampled=up)
    ts_a = torch.full((B, Cout, ops.conv_tile_count(Ho, Wo), 4), float("nan"), device=dev)
    ts_b = ts_a.clone()
    if up:
        want = ops.conv(act, pw, bias=bias, shift=shift, res1=r1, load_mode=2, tile_stats=ts_a, in_amax=ops.NORMALISED)
        got = ops.conv_up_img(img, pw, B, Cin, H, W, bias=bias, shift=shift, res1=r1, tile_stats=ts_b)
    else:
        want = ops.conv(act, pw, bias=bias, shift=shift, res1=r1, res1_upsampled=res_up, tile_stats=ts_a, in_amax=ops.NORMALISED)
        got = ops.conv_img(img, pw, B, Cin, H, W, bias=bias, shift=shift, res1=r1, res1_upsampled=res_up, tile_stats=ts_b)
    exact = adm or H * W <= 1024                                     # larger planes: ds_inorm_silu_images sums its statistics in another order
    exact = exact and os.environ.get("DS_CONV_SHAPE") != "32"        # the image-input kernels exist in the 16x16x32 form only
    if not exact:
        assert rel(got, want) < 2e-6
    else:
        assert torch.equal(got, want) and torch.equal(ts_a, ts_b), f"image route differs: up={up} adm={adm} Cin={Cin} Cout={Cout} {H}x{W} pool={pool}"
    a64 = act.double().cpu()
    src = F.interpolate(a64, scale_factor=2.0, mode="nearest") if up else a64
    ref = F.conv2d(src, w.double(), bias.double().cpu(), padding=1) + shift.double().cpu()[..., None, None]
    ref = ref + (F.interpolate(r1.double().cpu(), scale_factor=2.0, mode="nearest") if res_up else r1.double().cpu())
    return rel(got.cpu(), ref)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(a.seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))       # noqa: E731
    worst = 0.0
    for it in range(a.n):
        kind = ["3x3", "3x3", "3x3", "1x1", "direct", "up", "images", "images"][ri(0, 7)]
        if kind in ("up", "images"):
            e = fuzz_up(ri, g, dev) if kind == "up" else fuzz_images(ri, g, dev)
            worst = max(worst, e)
            if e > 3e-6:
                print(f"FAIL it={it} kind={kind} err={e:.3e}")
                sys.exit(1)
            if it % 25 == 0:
                print(f"it {it}: ok (worst so far {worst:.2e})", flush=True)
            continue
        B, Cin, Cout = ri(1, 3), ri(1, 80), ri(1, 140)
        H, W = ri(1, 12) * 2, ri(1, 20) * 2
        if ri(0, 3) == 0:
            W += 1 if kind != "1x1" else 0                                       # odd width: scalar epilogue path
        circ = kind != "1x1" and ri(0, 2) == 0
        mode = ri(0, 2) if kind == "3x3" else (ri(0, 2) * 0 if kind == "direct" else [0, 2, 3][ri(0, 2)])
        if mode in (1, 3):
            Hin, Win = 2 * H, 2 * W
        elif mode == 2:
            if W % 2:
                W += 1
            Hin, Win = H // 2, W // 2
        else:
            Hin, Win = H, W
        sc = 2.0 ** -SCALES[ri(0, len(SCALES) - 1)]
        x = (torch.randn(B, Cin, Hin, Win, generator=g) * 2 + 0.3) * sc
        src = x
        if mode == 1:
            src = F.max_pool2d(x, 2)
        elif mode == 3:
            src = F.avg_pool2d(x, 2)
        elif mode == 2:
            src = F.interpolate(x, scale_factor=2.0, mode="nearest")
        bias = torch.randn(Cout, generator=g) * sc
        pad = (lambda t: F.pad(F.pad(t, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1), mode="circular")) if circ else None
        if kind == "direct":
            Cout = ri(1, 4)
            bias = torch.randn(Cout, generator=g) * sc
            w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
            want = F.conv2d(pad(src.double()), w.double(), bias.double()) if circ else F.conv2d(src.double(), w.double(), bias.double(), padding="same")
            got = ops.conv_direct(x.to(dev), w.to(dev), bias.to(dev), circular=circ).cpu()
            e = rel(got, want)
        elif kind == "1x1":
            w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
            r1 = torch.randn(B, Cout, H, W, generator=g) * sc
            want = F.conv2d(src.double(), w.double(), bias.double()) + r1.double()
            nt = ops.conv_tile_count(H, W)
            ts = torch.zeros(B, Cout, nt, 4, device=dev) if W % 4 == 0 or True else None
            # the output's per-sample maxima (out_amax), optionally split at a 64-channel boundary (the attention's q, k | v)
            split = 64 * ri(1, (Cout - 1) // 64) if Cout > 64 and ri(0, 1) else 0
            am = torch.zeros((2, B) if split else (B,), dtype=torch.int32, device=dev)
            got = ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp16x3"), bias=bias.to(dev), res1=r1.to(dev), load_mode=mode,
                           tile_stats=ts, out_amax=am.view(-1), amax_split=split).cpu()
            e = rel(got, want)
            parts = [got[:, :split], got[:, split:]] if split else [got]
            bits = torch.stack([p_.reshape(B, -1).abs().amax(dim=1).contiguous().view(torch.int32) for p_ in parts]).view(am.shape)
            assert torch.equal(am.cpu(), bits), f"out_amax differs: 1x1 Cout={Cout} split={split}"
            K, S, Q, n = ts.cpu().double().unbind(-1)
            sx = (n * K + S).sum(-1)
            e = max(e, float((sx - want.sum(dim=(2, 3))).abs().max() / (want.abs().sum(dim=(2, 3)).max() + 1e-300)))
        else:
            w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
            pre = mode != 1 and ri(0, 1) == 1
            if pre and sc != 1.0:                                                # the table below normalises unit-scale data
                x, src, bias, sc = x / sc, src / sc, bias / sc, 1.0
            shift = torch.randn(B, Cout, generator=g) * sc
            r1 = torch.randn(B, Cout, H, W, generator=g) * sc
            s64 = src.double()
            tab = None
            if pre:
                M, A, C = torch.randn(B, Cin, generator=g) * 0.3, torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.3
                tab = torch.zeros(B, ops.table_channels(Cin), 4)
                tab[:, :Cin, 0], tab[:, :Cin, 1], tab[:, :Cin, 2] = M, A, C
                if ri(0, 1):
                    tab[:, :, 3] = 2.0 ** -ri(-4, 11)
                xin = x.double()
                act = F.silu((xin - M.double()[..., None, None]) * A.double()[..., None, None] + C.double()[..., None, None])
                s64 = F.interpolate(act, scale_factor=2.0, mode="nearest") if mode == 2 else act
            want = (F.conv2d(pad(s64), w.double(), bias.double()) if circ else F.conv2d(s64, w.double(), bias.double(), padding="same"))
            want = want + shift.double()[..., None, None] + r1.double()
            nt = ops.conv_tile_count(H, W)
            ts = torch.full((B, Cout, nt, 4), float("nan"), device=dev)
            am = torch.zeros(B, dtype=torch.int32, device=dev)
            got = ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp16x3"), bias=bias.to(dev), shift=shift.to(dev), res1=r1.to(dev),
                           load_mode=mode, circular=circ, prenorm=None if tab is None else tab.to(dev), tile_stats=ts, out_amax=am).cpu()
            e = rel(got, want)
            assert torch.equal(am.cpu(), got.reshape(B, -1).abs().amax(dim=1).contiguous().view(torch.int32)), "out_amax differs: 3x3"
            K, S, Q, n = ts.cpu().double().unbind(-1)
            assert torch.isfinite(ts).all(), "tile statistics not fully written"
            assert torch.equal(n.sum(-1), torch.full_like(n.sum(-1), H * W)), "tile pixel counts"
            sxx = (Q + 2 * K * S + n * K * K).sum(-1)
            e = max(e, float(((sxx - (want * want).sum(dim=(2, 3))).abs() / (want * want).sum(dim=(2, 3)).clamp_min(1e-300)).max()))
        worst = max(worst, e)
        if e > 3e-6:
            print(f"FAIL it={it} kind={kind} B={B} Cin={Cin} Cout={Cout} H={H} W={W} mode={mode} circ={circ} scale={sc:.3g} err={e:.3e}")
            sys.exit(1)
        if it % 25 == 0:
            print(f"it {it}: ok (worst so far {worst:.2e})", flush=True)
    print(f"all {a.n} cases passed; worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
