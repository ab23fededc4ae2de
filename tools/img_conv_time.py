"""ds_conv2d_h3_img (input as pre-split fp16 hi / lo images, staged by LDS-DMA) against ds_conv2d_h3 (fp32 input, split in the
kernel) on the same activation: equality of the results and launch times at config 2's layer shapes."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")


def build_images(a):
    """The layout of ds_inorm_silu_images, from an fp32 tensor, with torch ops (test helper)."""
    B, C, H, W = a.shape
    nch = (C + 15) // 16
    ap = torch.zeros(B, nch * 16, H + 2, W + 2, device=a.device)
    ap[:, :C, 1:-1, 1:-1] = a
    hi = ap.half()
    lo = (ap - hi.float()).half()
    v = torch.stack([hi, lo], dim=1)                                   # [B, piece, C, Hp, Wp]
    v = v.view(B, 2, nch, 2, 8, H + 2, W + 2).permute(0, 2, 1, 3, 5, 6, 4).contiguous()   # [B, chunk, piece, h, Hp, Wp, 8]
    return v.view(torch.float32).reshape(-1)


torch.manual_seed(0)
for (B, C, S) in [(2, 32, 32), (3, 64, 24), (64, 256, 32), (64, 128, 64), (64, 64, 128)]:
    x = torch.randn(B, C, S, S, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5)
    bias = torch.randn(C, device=dev)
    res = torch.randn(B, C, S, S, device=dev)
    pw = ops.pack_conv(w, "fp16x3")
    img = build_images(x)
    ts_a = torch.zeros(B, C, ops.conv_tile_count(S, S), 4, device=dev)
    ts_b = torch.zeros_like(ts_a)
    ya = ops.conv(x, pw, bias=bias, res1=res, tile_stats=ts_a)
    yb = ops.conv_img(img, pw, B, C, S, S, bias=bias, res1=res, tile_stats=ts_b)
    torch.cuda.synchronize()
    same = bool(torch.equal(ya, yb)) and bool(torch.equal(ts_a, ts_b))
    t = {}
    for name, f in (("fp32 input", lambda: ops.conv(x, pw, bias=bias, out=ya)), ("images", lambda: ops.conv_img(img, pw, B, C, S, S, bias=bias, out=yb))):
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record(); torch.cuda.synchronize()
        t[name] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"B={B} C={C} {S}x{S}: identical {same}   fp32 input {t['fp32 input']:.1f} us   images {t['images']:.1f} us   ratio {t['fp32 input'] / t['images']:.2f}", flush=True)
