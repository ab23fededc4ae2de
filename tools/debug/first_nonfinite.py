"""Debug: which op produces the first non-finite value in the extra_res variant's sampling run (fp16x3, auto_precision off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffsci_amd.models as M
from diffsci_amd import ops
from tests.golden_util import load

dev = torch.device("cuda:0")
v, sd = load("punetg8_extra_res")
net = M.PUNetG(M.PUNetGConfig(model_channels=8), extra_residual=torch.nn.AvgPool2d(3, stride=1, padding=1))
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
net.auto_precision = False
found = [False]


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if not found[0] and torch.is_tensor(out) and out.is_floating_point() and not bool(torch.isfinite(out).all()):
            found[0] = True
            ins = [t for t in a if torch.is_tensor(t) and t.is_floating_point()]
            print("FIRST non-finite from", name, "out", tuple(out.shape), "inputs:",
                  [(tuple(t.shape), float(t.abs().max()), bool(torch.isfinite(t).all())) for t in ins],
                  {kk: (vv if not torch.is_tensor(vv) else (str(vv.dtype), tuple(vv.shape), (vv.view(torch.float32).tolist() if vv.dtype == torch.int32 and vv.numel() <= 8 else float(vv.abs().max())))) for kk, vv in k.items() if kk not in ("out",)}, flush=True)
        return out
    setattr(ops, name, f)


for n in ("conv", "conv2d", "conv_img", "conv_up_img", "attention", "inorm_silu", "inorm_silu_images", "add", "conv_direct", "table_apply_images"):
    wrap(n)
module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm())
grids = load("schedule")[0]
sch = module.config.noisescheduler
o = sch.create_steps
sch.create_steps = lambda n: grids[f"steps_{n - 1}"].clone() if f"steps_{n - 1}" in grids else o(n)
hist = module.propagate_white_noise(v["white_noise"].to(dev), nsteps=6, record_history=True).cpu()
print("finite per step", [bool(torch.isfinite(h).all()) for h in hist], "max", [float(h.abs().max()) for h in hist])
