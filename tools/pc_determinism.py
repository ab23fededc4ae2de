"""Replay determinism of the headline workload on the persistent kernels: PUNetG-64 on [64,1,128,128], 4-step Heun (7 evaluations),
the same noise six times through the captured plan and twice eagerly -- every result must be bit-identical (an LDS race in the
producer / consumer hand-offs would show here as run-to-run differences)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffsci_amd.models as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = M.PUNetG(M.PUNetGConfig(model_channels=64))
module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
wn = torch.randn(64, 1, 128, 128, device=dev)
outs = [module.propagate_white_noise(wn, nsteps=4).clone() for _ in range(6)]
x, t = torch.randn(64, 1, 128, 128, device=dev), torch.rand(64, device=dev) + 0.5
e = [net(x, t).clone() for _ in range(4)]
torch.cuda.synchronize()
ok = all(torch.equal(outs[0], o) for o in outs[1:]) and all(torch.equal(e[0], v) for v in e[1:]) and bool(torch.isfinite(outs[0]).all())
print("sampler replays identical:", all(torch.equal(outs[0], o) for o in outs[1:]), " eager evaluations identical:", all(torch.equal(e[0], v) for v in e[1:]),
      " finite:", bool(torch.isfinite(outs[0]).all()))
# across builds / switches: `--save f` writes the two results, `--compare f` demands equality with them (the persistent and the one-shot
# kernels are bit-identical per launch, so the whole run must be)
if "--save" in sys.argv:
    torch.save((outs[0].cpu(), e[0].cpu()), sys.argv[sys.argv.index("--save") + 1])
if "--compare" in sys.argv:
    a, b = torch.load(sys.argv[sys.argv.index("--compare") + 1])
    same = torch.equal(a, outs[0].cpu()) and torch.equal(b, e[0].cpu())
    print("identical to the saved run:", same)
    ok = ok and same
sys.exit(0 if ok else 1)
