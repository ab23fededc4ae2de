import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops
dev = torch.device("cuda:0")
B, E, L = 1, 32, 256
g = torch.Generator().manual_seed(6)
qkv = torch.randn(B, 3*E, L, generator=g)
qkv[:, E:, :] = 0
for key in range(L):
    qkv[0, E + (key % E), key] = 1.0        # K: key -> e_{key%E}
    qkv[0, 2*E + (key % E), key] = 1.0      # V: same
q = qkv[0, :E, :].double() * math.sqrt(1.0 / E)      # [E, L]
want = torch.softmax(q, dim=0)                        # out[d][query] = softmax_d(q_d)
for prec in ("fp32", "fp16x3"):
    got = ops.attention(qkv.to(dev), E, precision=prec).cpu().double()[0]
    rel = ((got - want).abs() / want)
    print(prec, "max rel err of softmax weights:", float(rel.max()), "at (d,q)", [int(x) for x in (rel == rel.max()).nonzero()[0]])
    bad = (rel > 1e-5).nonzero()
    print("   #elements with rel err > 1e-5:", len(bad), bad[:10].tolist())
    if len(bad):
        d, qq = [int(x) for x in bad[0]]
        print("   q value there:", float(qkv[0, d, qq]), "scaled", float(q[d, qq]), "got", float(got[d, qq]), "want", float(want[d, qq]))
