"""ds_inorm_silu_images launch times (norm + SiLU written as the convolution's pre-split images) at the plane sizes of configs 2 / 5,
against the HBM time of its algorithmic traffic (4 B read + 4 B written per element, plus the zero border)."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, C, S) in [(64, 256, 32), (64, 256, 16), (64, 128, 64), (32, 256, 64), (16, 256, 64)]:
    x = torch.randn(B, C, S, S, device=dev)
    w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    out = torch.empty(ops.conv_images_floats(B, C, S, S), device=dev)
    ref = ops.inorm_silu(x, w, b, kind=0)
    for kind in (0, 1):
        for _ in range(5):
            ops.inorm_silu_images(x, w, b, kind, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.inorm_silu_images(x, w, b, kind, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        byts = x.numel() * 4 + out.numel() * 4
        print(f"B={B} C={C} {S}x{S} kind {kind}: {us:.1f} us   {byts / us / 1e3:.0f} GB/s", flush=True)
