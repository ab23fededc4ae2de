"""Time ds_conv2d_direct against the MFMA path for the three output-layer shapes."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (B, Cin, Cout, S) in [(64, 64, 1, 128), (16, 64, 4, 256), (32, 128, 3, 256)]:
    x = torch.randn(B, Cin, S, S, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, device=dev)
    pw = ops.pack_conv(w, "fp16x3")
    out = torch.empty(B, Cout, S, S, device=dev)
    td = timed(lambda: ops.conv_direct(x, w, b, out=out))
    tm = timed(lambda: ops.conv(x, pw, bias=b, out=out))
    gb = x.numel() * 4 / 1e9
    print(f"B{B} {Cin}->{Cout} @{S}: direct {td:7.1f} us ({gb/td*1e6:6.0f} GB/s)   mfma {tm:7.1f} us")
