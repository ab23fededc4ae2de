import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops
dev = torch.device("cuda:0")
def ref(qkv, E):
    q, k, v = (t.transpose(1, 2).double() for t in qkv.split(E, dim=1))
    s = (q * math.sqrt(1.0 / E)) @ k.transpose(1, 2)
    att = torch.softmax(s, dim=-1)
    return (att @ v).transpose(1, 2), s, att
B, E, L = 1, 32, 256
g = torch.Generator().manual_seed(6)
qkv = torch.randn(B, 3*E, L, generator=g)
want, s, att = ref(qkv, E)
got = ops.attention(qkv.to(dev), E, precision="fp16x3").cpu().double()
err = (got - want)[0]          # [E, L]
pq = err.norm(dim=0) / want[0].norm(dim=0)
bad = torch.nonzero(pq > 1e-5).flatten().tolist()
print("bad queries", bad)
for q in bad[:4]:
    print("query", q, "rel", float(pq[q]), "max att", float(att[0, q].max()), "argmax key", int(att[0, q].argmax()), "smax", float(s[0,q].max()), "smin", float(s[0,q].min()))
    print("   err per d (first 8):", [f"{float(x):.1e}" for x in err[:8, q]])
# V = ones -> output must be 1
qkv1 = qkv.clone(); qkv1[:, 2*E:, :] = 1.0
got1 = ops.attention(qkv1.to(dev), E, precision="fp16x3").cpu()
print("V=1: max |out-1| =", float((got1 - 1).abs().max()), "at query", int((got1-1).abs().max(dim=1)[0].argmax()))
# V = one-hot per key index (d = key % E)
qkv2 = qkv.clone(); qkv2[:, 2*E:, :] = 0
for key in range(L): qkv2[0, 2*E + (key % E), key] = 1.0
want2, _, att2 = ref(qkv2, E)
got2 = ops.attention(qkv2.to(dev), E, precision="fp16x3").cpu().double()
e2 = (got2 - want2)[0]
print("one-hot V: max abs err", float(e2.abs().max()), "rel", float(e2.norm()/want2.norm()))
q = bad[0] if bad else 0
# which key-groups (mod 32) are wrong for worst query: sum over keys with key%E==d of att = want2[d,q]
print("   per-d err for query", q, [f"{float(x):.1e}" for x in e2[:, q]])
