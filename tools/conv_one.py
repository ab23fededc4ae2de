"""One conv shape, a few launches -- for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops
dev = torch.device("cuda:0")
B, Cin, Cout, S = [int(v) for v in sys.argv[1:5]]
prec = sys.argv[5] if len(sys.argv) > 5 else "bf16x6"
x = torch.randn(B, Cin, S, S, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
wp = ops.pack_conv(w, prec)
out = torch.empty(B, Cout, S, S, device=dev)
for _ in range(6):
    ops.conv(x, wp, out=out)
torch.cuda.synchronize()
