"""Soak: many sampling calls with changing batch shapes / step counts / integrators through the plan cache; device
memory must plateau (plans are evicted, workspaces reused) and results stay finite and reproducible."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import diffsci_amd.models as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
nets = {"punetg": M.PUNetG(M.PUNetGConfig(model_channels=16)),
        # 32-channel ADM: its standalone-norm blocks take the image routes (pre-split images, tile-statistics norms, parity kernels)
        "adm": M.ADM(M.ADMConfig(model_channels=32, time_embed_dim=16, output_embed_dim=32))}
modules = {k: M.KarrasModule(n, M.KarrasModuleConfig.from_edm()).to(dev).eval() for k, n in nets.items()}
shapes = [(4, 32, 32), (2, 64, 32), (8, 32, 32), (3, 48, 16), (1, 64, 64), (6, 16, 16)]
first = {}
peak = []
for it in range(120):
    family = "punetg" if it % 2 == 0 else "adm"
    module = modules[family]
    B, H, W = shapes[(it // 2) % len(shapes)]
    n = [6, 9, 12][(it // 2) % 3]
    integ = ["heun", "euler", "karras"][(it // 4) % 3]
    torch.manual_seed(100 + (it // 2) % len(shapes))
    wn = torch.randn(B, 1, H, W, device=dev)
    eps = torch.randn(n, B, 1, H, W, device=dev) if integ == "karras" else None
    out = module.propagate_white_noise(wn, nsteps=n, integrator=integ, eps=eps)
    assert torch.isfinite(out).all()
    key = (family, B, H, W, n, integ)
    if integ != "karras":
        if key in first:
            assert torch.equal(first[key], out), f"iteration {it}: result changed for {key}"
        else:
            first[key] = out.clone()
    torch.cuda.synchronize()
    peak.append(torch.cuda.memory_allocated() / 2 ** 20)
    if it % 10 == 9:
        print(f"it {it}: {peak[-1]:.1f} MiB allocated, {sum(len(m._plans.plans) for m in modules.values())} plans", flush=True)
assert max(peak[80:]) <= max(peak[:80]) * 1.05 + 1, (max(peak[:80]), max(peak[80:]))
print("soak ok: memory plateaued at", round(max(peak), 1), "MiB")
