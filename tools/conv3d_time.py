"""Timing of ds_conv3d_direct (the 3-D path's convolution) on a few PUNetG-like layer shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diffsci_amd import ops

dev = torch.device("cuda:0")
for (B, Cin, Cout, S, mode) in [(8, 64, 64, 32, 0), (8, 128, 128, 16, 0), (8, 64, 128, 16, 1), (8, 128, 64, 32, 2), (8, 1, 64, 32, 0)]:
    shp = {0: S, 1: 2 * S, 2: S // 2}[mode]
    x = torch.randn(B, Cin, shp, shp, shp, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
    out = torch.empty(B, Cout, S, S, S, device=dev)
    ops.conv3d(x, w, load_mode=mode, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3d(x, w, load_mode=mode, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * Cout * Cin * 27 * S ** 3
    line = f"B={B} {Cin}->{Cout} @{S}^3 mode={mode}: direct {ms:.3f} ms = {fl / ms / 1e9:.1f} TFLOP/s"
    if Cout > 4:
        packs = ops.pack_conv3d(w, upsampled=mode == 2)
        ops.conv3d_mfma(x, packs, load_mode=mode, out=out)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.conv3d_mfma(x, packs, load_mode=mode, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / 10
        line += f";  matrix cores (3 x 2-D fp16x3 + slice copies) {ms2:.3f} ms = {fl / ms2 / 1e9:.1f} TFLOP/s-equivalent"
    print(line)
