"""The two routes of a standalone norm + SiLU written as pre-split images at config 2's 256-channel level: ds_inorm_silu_images (statistics
recomputed from the tensor, precise activation) against ds_inorm_table + ds_table_apply_images (statistics from the producer's tile
statistics, the fused loader's arithmetic).   python tools/table_images_time.py"""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, C, S) in [(64, 256, 32), (64, 128, 64)]:
    x = torch.randn(B, C, S, S, device=dev)
    w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    out = torch.empty(ops.conv_images_floats(B, C, S, S), device=dev)
    nt = ops.conv_tile_count(S, S)
    ts = torch.zeros(B, C, nt, 4, device=dev)
    ts[..., 3] = S * S / nt
    ts[..., 2] = S * S / nt
    tab = torch.empty(B, ops.table_channels(C), 4, device=dev)

    def route_a():
        ops.inorm_silu_images(x, w, b, 0, out=out)

    def route_b():
        ops.inorm_table(ts, w, b, 0, S * S, out=tab)
        ops.table_apply_images(x, tab, out=out)

    def route_c():
        ops.table_apply_images(x, tab, out=out)

    for name, f in (("inorm_silu_images", route_a), ("table + table_apply_images", route_b), ("table_apply_images alone", route_c)):
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} C={C} {S}x{S} {name}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us", flush=True)
