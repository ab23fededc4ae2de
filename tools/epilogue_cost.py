"""What the epilogue's optional parts cost in the fused-loader convolution (ds_conv2d_h3 with prenorm) at config 2's level-0 / level-1
shapes: launch time with / without the tile statistics and the residual."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, C, S) in [(64, 64, 128), (64, 128, 64)]:
    x = torch.randn(B, C, S, S, device=dev)
    res = torch.randn(B, C, S, S, device=dev)
    pw = ops.pack_conv(torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5), "fp16x3")
    bias, shift = torch.randn(C, device=dev), torch.randn(1, C, device=dev)
    tab = torch.zeros(B, ops.table_channels(C), 4, device=dev)
    tab[:, :C, 1] = 1.0
    ts = torch.zeros(B, C, ops.conv_tile_count(S, S), 4, device=dev)
    out = torch.empty_like(x)
    for name, kw in (("stats + residual", dict(tile_stats=ts, res1=res)), ("stats", dict(tile_stats=ts)),
                     ("residual", dict(res1=res)), ("neither", dict()), ("stats + residual", dict(tile_stats=ts, res1=res))):
        f = lambda: ops.conv(x, pw, bias=bias, shift=shift, prenorm=tab, out=out, **kw)   # noqa: E731
        for _ in range(10):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} C={C} {S}x{S} fused loader, {name:17s}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us", flush=True)
