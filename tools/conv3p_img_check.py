"""The persistent convolution's image-input form (ds_conv3p.hip, IMG) against the one-shot image-input kernel (ds_conv3h.hip, IMGIN):
outputs, tile statistics and output maxima bit for bit (the reference arm runs in a child process with DS_CONV_PC_IMG=0), and
against fp64 on two samples.      python tools/conv3p_img_check.py"""
import math
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (B, Cin, Cout, H, W, res1, res2, stats, amax)
CASES = [
    (64, 256, 256, 32, 32, 1, 0, 1, 1),        # config 2, level 2
    (64, 256, 256, 32, 32, 0, 0, 1, 0),
    (16, 128, 128, 64, 64, 1, 1, 1, 1),
    (8, 64, 64, 128, 128, 1, 0, 0, 1),
    (5, 128, 256, 40, 64, 0, 1, 1, 0),         # 5 x 4 x 10 = 400 items: uneven counts per workgroup
    (3, 192, 64, 88, 96, 1, 0, 1, 1),          # 99 items: below the persistent kernel's minimum -> both arms on the one-shot kernel
]


def build_images(a):
    import torch
    B, C, H, W = a.shape
    nch = (C + 15) // 16
    ap = torch.zeros(B, nch * 16, H + 2, W + 2, device=a.device)
    ap[:, :C, 1:-1, 1:-1] = a
    hi = ap.half()
    lo = (ap - hi.float()).half()
    v = torch.stack([hi, lo], dim=1)
    v = v.view(B, 2, nch, 2, 8, H + 2, W + 2).permute(0, 2, 1, 3, 5, 6, 4).contiguous()
    return v.view(torch.float32).reshape(-1)


def run_cases(dev):
    import torch
    from diffsci_amd import ops
    outs = []
    for ci, (B, Cin, Cout, H, W, r1, r2, st, am) in enumerate(CASES):
        g = torch.Generator().manual_seed(300 + ci)
        x = (torch.randn(B, Cin, H, W, generator=g) * 1.3).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(dev)
        bias = torch.randn(Cout, generator=g).to(dev)
        shift = torch.randn(B, Cout, generator=g).to(dev)
        res1 = torch.randn(B, Cout, H, W, generator=g).to(dev) if r1 else None
        res2 = torch.randn(B, Cout, H, W, generator=g).to(dev) if r2 else None
        ts = torch.full((B, Cout, ops.conv_tile_count(H, W), 4), float("nan"), device=dev) if st else None
        oa = torch.zeros(B, dtype=torch.int32, device=dev) if am else None
        pw = ops.pack_conv(w, "fp16x3")
        got = ops.conv_img(build_images(x), pw, B, Cin, H, W, bias=bias, shift=shift, res1=res1, res2=res2, tile_stats=ts, out_amax=oa)
        torch.cuda.synchronize()
        outs.append(dict(out=got.cpu(), ts=None if ts is None else ts.cpu(), oa=None if oa is None else oa.cpu(), x=x.cpu(), w=w.cpu(),
                         bias=bias.cpu(), shift=shift.cpu(), res1=None if res1 is None else res1.cpu(), res2=None if res2 is None else res2.cpu()))
    return outs


def main():
    import torch
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        torch.save([dict(out=o["out"], ts=o["ts"], oa=o["oa"]) for o in run_cases(dev)], sys.argv[2])
        return
    with tempfile.TemporaryDirectory() as td:
        ref_path = os.path.join(td, "ref.pt")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", ref_path], env=dict(os.environ, DS_CONV_PC_IMG="0"))
        ref = torch.load(ref_path)
    os.environ["DS_CONV_PC_IMG"] = "1"
    outs = run_cases(dev)
    bad = 0
    for ci, (case, o, r) in enumerate(zip(CASES, outs, ref)):
        same = torch.equal(o["out"], r["out"]) and (o["ts"] is None or torch.equal(o["ts"], r["ts"])) and (o["oa"] is None or torch.equal(o["oa"], r["oa"]))
        nb = min(case[0], 2)
        want = F.conv2d(F.pad(o["x"][:nb].double(), (1, 1, 1, 1)), o["w"].double()) + o["bias"].double()[None, :, None, None] + o["shift"][:nb].double()[:, :, None, None]
        for rr in (o["res1"], o["res2"]):
            if rr is not None:
                want = want + rr[:nb].double()
        rel = float((o["out"][:nb].double() - want).norm() / want.norm())
        ok = same and rel < 2e-6
        bad += 0 if ok else 1
        print(f"case {ci} {case}: outputs / statistics / maxima {'==' if same else '!='}  rel-L2 vs fp64 {rel:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
    print("conv3p_img_check:", "ALL OK" if bad == 0 else f"{bad} FAILED")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
