"""ds_conv1x1_h3 (fp16x3 1x1 convolution: attention projections, ADM's convresidual) at ADM-128 / PUNetG shapes: launch time and
algorithmic HBM rate (input read once + output written once, fp32)."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(32, 256, 128, 256), (32, 384, 128, 256), (32, 512, 256, 128), (32, 768, 384, 64), (32, 1024, 512, 32),
          (32, 512, 1536, 16), (64, 256, 768, 32), (16, 256, 768, 64)]
for (B, Cin, Cout, S) in SHAPES:
    x = torch.randn(B, Cin, S, S, device=dev)
    pw = ops.pack_conv(torch.randn(Cout, Cin, 1, 1, device=dev) / Cin ** 0.5, "fp16x3")
    bias = torch.randn(Cout, device=dev)
    out = torch.empty(B, Cout, S, S, device=dev)
    am = ops.absmax_rows(x)
    f = lambda: ops.conv(x, pw, bias=bias, out=out, in_amax=am)   # noqa: E731
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    gb = (x.numel() + out.numel()) * 4 / 1e9
    tf = 2 * B * Cout * Cin * S * S / 1e12
    print(f"B={B} {Cin}->{Cout} {S}x{S}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s algorithmic  {tf / us * 1e6:6.1f} TF/s-equiv", flush=True)
