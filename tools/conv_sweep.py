import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops
from tools.conv_diag import timeit
dev = torch.device("cuda:0")
B, S, Cout = 64, 128, 64
for Cin in (16, 32, 64, 128, 256):
    x = torch.randn(B, Cin, S, S, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    wp = ops.pack_conv(w, os.environ.get("PREC","fp16x3")); out = torch.empty(B, Cout, S, S, device=dev)
    ms = timeit(lambda: ops.conv(x, wp, out=out), 20)
    print(f"Cin={Cin:4d}: {ms*1e3:8.1f} us   per-chunk {ms*1e3/(Cin/16):7.1f} us", flush=True)
# small grids: 1 round (512 WGs) and 2 rounds
for Bq in (8, 16):
    x = torch.randn(Bq, 64, S, S, device=dev); w = torch.randn(Cout, 64, 3, 3, device=dev) * 0.05
    wp = ops.pack_conv(w, os.environ.get("PREC","fp16x3")); out = torch.empty(Bq, Cout, S, S, device=dev)
    ms = timeit(lambda: ops.conv(x, wp, out=out), 50)
    print(f"B={Bq} (WGs={Bq*64}): {ms*1e3:8.1f} us", flush=True)
