"""cProfile of the host side of one captured replay per forecast (tools/forecast_time.py's workload)."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.getcwd())
import torch
import diffsci_amd.models as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = M.PUNetG(M.PUNetGConfig(model_channels=32, input_channels=4, output_channels=4))
module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
x = torch.randn(4, 4, 32, 32, device=dev)
with torch.inference_mode():
    for _ in range(3):
        module.propagate_white_noise(x, nsteps=10)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        y = module.propagate_white_noise(x, nsteps=10)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
