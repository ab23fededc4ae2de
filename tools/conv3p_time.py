"""Launch times of the fused-loader convolutions at config 2's level-0 / level-1 shapes under the library's current DS_CONV_PC
setting (run once per setting: the switch is read once per process).   python tools/conv3p_time.py [reps]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
print("DS_CONV_PC =", os.environ.get("DS_CONV_PC", "(default)"), " DS_CONV_TWO =", os.environ.get("DS_CONV_TWO", "(default)"))
for name, (B, C, S), res in (("level0 conv1 (shift)", (64, 64, 128), False), ("level0 conv2 (residual)", (64, 64, 128), True),
                             ("level1 conv1 (shift)", (64, 128, 64), False), ("level1 conv2 (residual)", (64, 128, 64), True)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, S, S, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    shift = torch.randn(B, C, generator=g).to(dev)
    res1 = torch.randn(B, C, S, S, generator=g).to(dev) if res else None
    tab = torch.zeros(B, ops.table_channels(C), 4)
    tab[:, :C, 0] = torch.randn(B, C, generator=g) * 0.3
    tab[:, :C, 1] = torch.rand(B, C, generator=g) + 0.5
    tab[:, :C, 2] = torch.randn(B, C, generator=g) * 0.3
    tab[:, :, 3] = 2.0 ** -3
    tab = tab.to(dev)
    ts = torch.empty(B, C, ops.conv_tile_count(S, S), 4, device=dev)
    oa = torch.zeros(B, dtype=torch.int32, device=dev)
    pw = ops.pack_conv(w, "fp16x3")
    out = torch.empty(B, C, S, S, device=dev)
    kw = dict(bias=bias, shift=shift, res1=res1, prenorm=tab, tile_stats=ts, out_amax=oa if res else None, out=out)
    for _ in range(20):
        ops.conv(x, pw, **kw)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        ops.conv(x, pw, **kw)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / reps * 1e3
    gf = 2.0 * B * C * C * 9 * S * S / 1e9
    print(f"{name:26s} [{B},{C},{S},{S}]: {us:7.1f} us   {gf / us * 1e3:6.1f} TF/s-equiv   frac of 833 = {gf / us * 1e3 / 833.3:.3f}", flush=True)
