"""PUNetG 64-ch on 512 x 512 fields (the 256-channel levels have 128^2- and 64^2-pixel planes): one evaluation with the standalone
norms writing images (the table route for planes beyond 4096 floats) against DIFFSCI_NORM_IMAGES=0."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
import diffsci_amd.models as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = M.PUNetG(M.PUNetGConfig(model_channels=64)).to(dev).eval()
x, t = torch.randn(4, 1, 512, 512, device=dev), torch.full((4,), 0.3, device=dev)
outs = {}
for images in (True, False, True, False):
    net.norm_images = images
    for _ in range(2):
        y = net(x, t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = net(x, t)
    e1.record(); torch.cuda.synchronize()
    outs[images] = y.clone()
    print(f"norm_images={images}: {e0.elapsed_time(e1) / 5:.2f} ms per evaluation", flush=True)
print("relative difference of the two routes:", float((outs[True] - outs[False]).norm() / outs[False].norm()))
