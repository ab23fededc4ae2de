"""Does a level-0 residual block run faster per sample when the batch is cut so that its tensors stay in the Infinity Cache?
A block = conv1 (fused norm + SiLU loader, shift) -> table -> conv2 (fused loader, residual); config 2's level 0 is [64,64,128,128]:
268 MB per tensor, three tensors per block -- past the 256 MiB cache.  Runs the chain over the whole batch in sub-batches of
B / parts samples, depth first (both convolutions of a sub-batch before the next sub-batch), and prints the time per whole batch.
    python tools/subbatch_time.py [C S [reps]]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
B = 64
print("DS_CONV_PC =", os.environ.get("DS_CONV_PC", "(default)"), f" C = {C}  S = {S}")
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, S, S, generator=g).to(dev)
h = torch.empty_like(x)
out = torch.empty_like(x)
w1 = ops.pack_conv((torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev), "fp16x3")
w2 = ops.pack_conv((torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev), "fp16x3")
bias = torch.randn(C, generator=g).to(dev)
shift = torch.randn(B, C, generator=g).to(dev)
tab = torch.zeros(B, ops.table_channels(C), 4)
tab[:, :C, 0] = torch.randn(B, C, generator=g) * 0.3
tab[:, :C, 1] = torch.rand(B, C, generator=g) + 0.5
tab[:, :C, 2] = torch.randn(B, C, generator=g) * 0.3
tab[:, :, 3] = 2.0 ** -3
tab = tab.to(dev)
ts = torch.empty(B, C, ops.conv_tile_count(S, S), 4, device=dev)
oa = torch.zeros(B, dtype=torch.int32, device=dev)


def block(parts, depth_first=True):
    n = B // parts
    order = [(q, k) for q in range(parts) for k in range(2)] if depth_first else [(q, k) for k in range(2) for q in range(parts)]
    for q, k in order:
        sl = slice(q * n, (q + 1) * n)
        if k == 0:
            ops.conv(x[sl], w1, bias=bias, shift=shift[sl], prenorm=tab[sl], tile_stats=ts[sl], out=h[sl])
        else:
            ops.conv(h[sl], w2, bias=bias, res1=x[sl], prenorm=tab[sl], tile_stats=ts[sl], out_amax=oa[sl], out=out[sl])


for parts, df in ((1, True), (2, True), (4, True), (8, True), (4, False), (1, True)):
    for _ in range(3):
        block(parts, df)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        block(parts, df)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / reps * 1e3
    print(f"parts = {parts}  {'depth first' if df else 'layer by layer'}: {us:8.1f} us per block of {B} samples "
          f"({B // parts} samples, {x[0].numel() * 4 * (B // parts) / 2**20:.0f} MiB per tensor and launch)", flush=True)
