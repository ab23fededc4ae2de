"""Smaller batches: is the persistent kernel still the better choice with 1 or 2 items per workgroup?  Launch times of the level-1
fused-loader shape and the level-2 image-input shape at batch 16 and 32 under the current DS_CONV_PC_MIN (run once per value)."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
print("DS_CONV_PC_MIN =", os.environ.get("DS_CONV_PC_MIN", "(default 4)"))


def build_images(a):
    B, C, H, W = a.shape
    nch = (C + 15) // 16
    ap = torch.zeros(B, nch * 16, H + 2, W + 2, device=a.device)
    ap[:, :C, 1:-1, 1:-1] = a
    hi = ap.half()
    lo = (ap - hi.float()).half()
    v = torch.stack([hi, lo], dim=1).view(B, 2, nch, 2, 8, H + 2, W + 2).permute(0, 2, 1, 3, 5, 6, 4).contiguous()
    return v.view(torch.float32).reshape(-1)


def timed(f):
    for _ in range(10):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3


for B in (8, 16, 32):
    g = torch.Generator().manual_seed(B)
    for (C, S, img) in ((64, 128, False), (128, 64, False), (256, 32, True)):
        x = torch.randn(B, C, S, S, generator=g).to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
        res = torch.randn(B, C, S, S, generator=g).to(dev)
        pw = ops.pack_conv(w, "fp16x3")
        ts = torch.empty(B, C, ops.conv_tile_count(S, S), 4, device=dev)
        out = torch.empty(B, C, S, S, device=dev)
        items = (C // 64) * (S // 8) * (S // 32) * B
        if img:
            im = build_images(x)
            us = timed(lambda: ops.conv_img(im, pw, B, C, S, S, res1=res, tile_stats=ts, out=out))
        else:
            tab = torch.zeros(B, ops.table_channels(C), 4, device=dev)
            tab[:, :, 1] = 1.0
            tab[:, :, 3] = 0.125
            us = timed(lambda: ops.conv(x, pw, res1=res, prenorm=tab, tile_stats=ts, out=out))
        print(f"B={B} [{C},{S},{S}] {'image input' if img else 'fused loader'}: {items} items = {items / 256:.1f} per workgroup: {us:.1f} us", flush=True)
