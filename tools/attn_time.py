"""Time the fp16x3 attention kernel at the bottleneck shapes of configs 2 and 5."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from diffsci_amd import ops
dev = torch.device("cuda:0")
for (B, E, L) in [(64, 256, 1024), (16, 256, 4096), (64, 128, 1024), (64, 64, 1024)]:
    qkv = torch.randn(B, 3 * E, L, device=dev)
    out = torch.empty(B, E, L, device=dev)
    f = lambda: ops.attention(qkv, E, out=out, precision="fp16x3")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B{B} E{E} L{L}: {us:8.1f} us  {4.0*L*L*E*B/us/1e6:6.1f} TF-eq")
