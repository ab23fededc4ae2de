"""Forward time of PUNetG(dimension=3) with and without the norms folded into the volume path (round 2).
    python tools/vol_time.py [B] [side] [channels]"""
import sys
import time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsci_amd.models as M  # noqa: E402

B, S, C = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 4), (2, 32), (3, 64)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = M.PUNetG(M.PUNetGConfig(model_channels=C, dimension=3)).to(dev).eval()
x = torch.randn(B, 1, S, S, S, device=dev)
t = torch.full((B,), 0.3, device=dev)
outs = {}
with torch.inference_mode():
    for fuse in (True, False, True, False):
        net.fuse_norm = fuse
        for _ in range(3):
            y = net(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = net(x, t)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        outs[fuse] = y
        print(f"PUNetG-3D {C} ch, [{B},1,{S},{S},{S}], norms {'folded' if fuse else 'standalone'}: {ms:.2f} ms per forward")
d = (outs[True] - outs[False]).norm() / outs[False].norm()
print(f"folded vs standalone rel-L2 {float(d):.2e}")
