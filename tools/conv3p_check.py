"""The persistent producer / consumer convolution (ds_conv3p.hip) against the one-tile-per-workgroup kernel (ds_conv3h.hip) and fp64.

The two kernels accumulate in the same order (same tap pairs, same tile statistics reductions), so outputs, tile statistics and
output maxima must agree BIT FOR BIT; the library reads DS_CONV_PC once per process, so the reference arm runs in a child
process with DS_CONV_PC=0 and hands its results over in a file.

    python tools/conv3p_check.py            # parent: DS_CONV_PC=3, DS_CONV_PC_MIN=1 unless set
"""
import math
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (B, Cin, Cout, H, W, pre, res1, res2, res1_up, stats, amax, circular, tap)
CASES = [
    (8, 64, 64, 128, 128, 1, 1, 0, 0, 1, 1, 0, None),
    (8, 64, 64, 128, 128, 1, 0, 0, 0, 1, 0, 0, None),
    (10, 32, 64, 72, 96, 1, 1, 1, 0, 1, 1, 0, None),          # 270 items: uneven item counts per workgroup
    (16, 128, 128, 64, 64, 1, 1, 0, 0, 1, 1, 0, None),        # two channel tiles
    (9, 64, 128, 48, 64, 1, 0, 0, 0, 0, 0, 1, None),          # periodic padding, nothing optional
    (8, 64, 64, 64, 128, 1, 1, 0, 1, 1, 0, 0, None),          # low-resolution residual: stays on the one-tile kernel
    (8, 32, 64, 128, 128, 0, 1, 0, 0, 1, 1, 0, None),         # raw input (activation exponent from in_amax)
    (8, 64, 64, 64, 128, 0, 0, 0, 0, 0, 0, 0, (1, -1)),       # tap offset
    (64, 64, 64, 128, 128, 1, 1, 0, 0, 1, 1, 0, None),        # config 2, level 0
    (64, 128, 128, 64, 64, 1, 1, 0, 0, 1, 1, 0, None),        # config 2, level 1
]


# --var DS_CONV_VEC: the 16-byte patch loads of ds_conv3h.hip (VEC) against its one-pixel staging items, same kernel otherwise
# (both arms with DS_CONV_PC=0): more shapes whose staging differs -- ragged heights, a row tap offset, periodic padding with one tile
# column (the halo columns wrap into the tile's own columns), a single tile, 96 and 32 input channels, raw inputs
VEC_CASES = [
    (3, 64, 64, 20, 32, 1, 1, 0, 0, 1, 1, 0, None),
    (2, 32, 64, 8, 64, 1, 0, 0, 0, 1, 0, 1, None),
    (2, 96, 128, 36, 96, 1, 1, 1, 0, 1, 1, 1, None),
    (2, 64, 64, 24, 64, 0, 0, 0, 0, 0, 0, 0, (1, 0)),
    (2, 64, 64, 24, 64, 0, 0, 0, 0, 0, 0, 0, (-1, 0)),
    (1, 32, 64, 8, 32, 0, 1, 0, 0, 1, 1, 1, None),
    (5, 128, 128, 16, 128, 1, 0, 0, 0, 1, 1, 0, None),
    (2, 160, 64, 40, 32, 1, 0, 1, 0, 0, 1, 0, None),
]


def random_cases(n, seed):
    """n more launch shapes the persistent kernel takes (>= 256 items, full tiles, channel counts multiples of 64), every option drawn."""
    import random
    rnd = random.Random(seed)
    out = []
    while len(out) < n:
        Cin, Cout = 64 * rnd.randint(1, 3), 64 * rnd.randint(1, 3)
        H, W = 8 * rnd.randint(1, 12), 32 * rnd.randint(1, 4)
        items_per_sample = (Cout // 64) * (H // 8) * (W // 32)
        B = max(1, -(-rnd.choice([256, 300, 520, 1100]) // items_per_sample))
        if B * Cin * H * W > 3e7 or B * Cout * H * W > 3e7:
            continue
        pre = rnd.randint(0, 3) > 0
        tap = None if pre or rnd.randint(0, 2) else (rnd.randint(-1, 1), rnd.randint(-1, 1))
        circ = 0 if tap else int(rnd.randint(0, 3) == 0)
        out.append((B, Cin, Cout, H, W, int(pre), rnd.randint(0, 1), int(rnd.randint(0, 3) == 0), 0, rnd.randint(0, 1), rnd.randint(0, 1), circ, tap))
    return out


def run_cases(dev):
    import torch
    from diffsci_amd import ops
    outs = []
    for ci, (B, Cin, Cout, H, W, pre, r1, r2, r1up, st, am, circ, tap) in enumerate(CASES):
        g = torch.Generator().manual_seed(100 + ci)
        x = (torch.randn(B, Cin, H, W, generator=g) * 1.7 + 0.2).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(dev)
        bias = torch.randn(Cout, generator=g).to(dev)
        shift = torch.randn(B, Cout, generator=g).to(dev)
        res1 = torch.randn(B, Cout, *((H // 2, W // 2) if r1up else (H, W)), generator=g).to(dev) if r1 else None
        res2 = torch.randn(B, Cout, H, W, generator=g).to(dev) if r2 else None
        tab = None
        if pre:
            tab = torch.zeros(B, ops.table_channels(Cin), 4)
            tab[:, :Cin, 0] = torch.randn(B, Cin, generator=g) * 0.3
            tab[:, :Cin, 1] = torch.rand(B, Cin, generator=g) + 0.5
            tab[:, :Cin, 2] = torch.randn(B, Cin, generator=g) * 0.3
            tab[:, :, 3] = 2.0 ** -3
            tab = tab.to(dev)
        ts = torch.full((B, Cout, ops.conv_tile_count(H, W), 4), float("nan"), device=dev) if st else None
        oa = torch.zeros(B, dtype=torch.int32, device=dev) if am else None
        pw = ops.pack_conv(w, "fp16x3")
        got = ops.conv(x, pw, bias=bias, shift=shift, res1=res1, res2=res2, res1_upsampled=bool(r1up), prenorm=tab,
                       tile_stats=ts, out_amax=oa, circular=bool(circ), tap_offset=tap)
        torch.cuda.synchronize()
        outs.append(dict(out=got.cpu(), ts=None if ts is None else ts.cpu(), oa=None if oa is None else oa.cpu(),
                         x=x.cpu(), w=w.cpu(), bias=bias.cpu(), shift=shift.cpu(), res1=None if res1 is None else res1.cpu(),
                         res2=None if res2 is None else res2.cpu(), tab=None if tab is None else tab.cpu()))
    return outs


def main():
    import torch
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    if "--random" in sys.argv:                         # python tools/conv3p_check.py --random N [seed]: N random shapes instead of the fixed ten
        i = sys.argv.index("--random")
        n, seed = int(sys.argv[i + 1]), int(sys.argv[i + 2]) if len(sys.argv) > i + 2 and sys.argv[i + 2].isdigit() else 0
        CASES[:] = random_cases(n, seed)
        globals().update(_RANDOM=True, _SEED=seed)
        del sys.argv[i:i + (3 if len(sys.argv) > i + 2 and sys.argv[i + 2].isdigit() else 2)]
    var = "DS_CONV_PC"
    if "--var" in sys.argv:                            # python tools/conv3p_check.py --var DS_CONV_VEC: that switch on (parent) against off (child)
        i = sys.argv.index("--var")
        var = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        if not globals().get("_RANDOM"):
            CASES.extend(VEC_CASES)
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        outs = run_cases(dev)
        torch.save([dict(out=o["out"], ts=o["ts"], oa=o["oa"]) for o in outs], sys.argv[2])
        return
    if var == "DS_CONV_PC":
        os.environ.setdefault("DS_CONV_PC", "3")
        os.environ.setdefault("DS_CONV_PC_MIN", "1")
        env = dict(os.environ, DS_CONV_PC="0")
    else:
        os.environ["DS_CONV_PC"] = "0"
        os.environ[var] = "1"
        env = dict(os.environ, **{var: "0"})
    with tempfile.TemporaryDirectory() as td:
        ref_path = os.path.join(td, "ref.pt")
        extra = ["--random", str(len(CASES)), str(globals().get("_SEED", 0))] if globals().get("_RANDOM") else []
        if var != "DS_CONV_PC":
            extra += ["--var", var]
        subprocess.check_call([sys.executable, os.path.abspath(__file__)] + extra + ["--child", ref_path], env=env)
        ref = torch.load(ref_path)
    outs = run_cases(dev)
    bad = 0
    for ci, (case, o, r) in enumerate(zip(CASES, outs, ref)):
        B, Cin, Cout, H, W, pre, r1, r2, r1up, st, am, circ, tap = case
        same_out = torch.equal(o["out"], r["out"])
        same_ts = o["ts"] is None or torch.equal(o["ts"], r["ts"])
        same_oa = o["oa"] is None or torch.equal(o["oa"], r["oa"])
        # fp64 reference on (up to) two samples
        nb = min(B, 2)
        xin = o["x"][:nb].double()
        if pre:
            t = o["tab"][:nb, :Cin].double()
            xin = F.silu((xin - t[..., 0, None, None]) * t[..., 1, None, None] + t[..., 2, None, None])
        oy, ox = tap if tap else (0, 0)
        if circ:
            xp = F.pad(xin, (1, 1, 1, 1), mode="circular")
            want = F.conv2d(xp, o["w"].double())
        else:
            xp = F.pad(xin, (1 + abs(ox), 1 + abs(ox), 1 + abs(oy), 1 + abs(oy)))
            full = F.conv2d(xp, o["w"].double())
            want = full[:, :, abs(oy) + oy: abs(oy) + oy + H, abs(ox) + ox: abs(ox) + ox + W]
        want = want + o["bias"].double()[None, :, None, None] + o["shift"][:nb].double()[:, :, None, None]
        if o["res1"] is not None:
            rr = o["res1"][:nb].double()
            want = want + (F.interpolate(rr, scale_factor=2.0, mode="nearest") if r1up else rr)
        if o["res2"] is not None:
            want = want + o["res2"][:nb].double()
        rel = float((o["out"][:nb].double() - want).norm() / want.norm())
        rel_ref = float((r["out"][:nb].double() - want).norm() / want.norm())
        ok = same_out and same_ts and same_oa and rel < 2e-6
        bad += 0 if ok else 1
        print(f"case {ci} {case}: out {'==' if same_out else '!='}  stats {'==' if same_ts else '!='}  amax {'==' if same_oa else '!='}"
              f"  rel-L2 vs fp64 {rel:.2e} (one-tile kernel {rel_ref:.2e})  {'ok' if ok else 'FAIL'}", flush=True)
        if not same_out:
            d = (o["out"] != r["out"])
            idx = d.nonzero()
            print("   first mismatches:", idx[:5].tolist(), " count", int(d.sum()), "of", d.numel(),
                  " max abs diff", float((o["out"] - r["out"]).abs().max()), flush=True)
    print("conv3p_check:", "ALL OK" if bad == 0 else f"{bad} FAILED")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
