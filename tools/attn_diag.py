import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops
dev = torch.device("cuda:0")
def ref(qkv, E):
    q, k, v = (t.transpose(1, 2).double() for t in qkv.split(E, dim=1))
    att = torch.softmax((q * math.sqrt(1.0 / E)) @ k.transpose(1, 2), dim=-1) @ v
    return att.transpose(1, 2)
for (B, E, L, seed) in [(2,128,256,384),(2,128,256,1),(1,128,128,2),(1,128,1024,3),(1,64,256,4),(2,256,256,5),(1,32,256,6)]:
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(B, 3*E, L, generator=g)
    want = ref(qkv, E)
    for prec in ("fp32", "fp16x3"):
        got = ops.attention(qkv.to(dev), E, precision=prec).cpu().double()
        err = (got - want)
        rel = (err.norm()/want.norm()).item()
        # per-query error
        pq = err.norm(dim=1) / want.norm(dim=1)
        print(f"B={B} E={E} L={L} {prec}: rel {rel:.2e} max-abs {err.abs().max():.2e}  worst query rel {pq.max():.2e} at {int(pq.argmax())%L}; frac queries >1e-5: {(pq>1e-5).float().mean():.3f}")
