import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffsci_amd.models as M
from diffsci_amd import ops
from diffsci_amd.models.nets import precision
from tests.golden_util import load, rel_l2

dev = torch.device("cuda:0")
v, sd = load("punetg8_3d")
net = M.PUNetG(M.PUNetGConfig(model_channels=8, dimension=3))
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
net.auto_precision = False
module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm())
grids = load("schedule")[0]
sch = module.config.noisescheduler
orig = sch.create_steps
sch.create_steps = lambda n: grids[f"steps_{n - 1}"].clone() if f"steps_{n - 1}" in grids else orig(n)
wn = v["white_noise"].to(dev)
for use_graph in (False, True, True):
    module.use_graph = use_graph
    hist = module.propagate_white_noise(wn, nsteps=4, record_history=True).cpu()
    fin = [bool(torch.isfinite(hist[i]).all()) for i in range(hist.shape[0])]
    print("graph" if use_graph else "eager", fin, [float(rel_l2(hist[i], v["hist_heun_N4_f32"][i])) for i in range(hist.shape[0])], flush=True)

print("--- conv3d_mfma with in_amax=NORMALISED")
orig_mfma = ops.conv3d_mfma
def f(*a, **k):
    k["in_amax"] = ops.NORMALISED
    return orig_mfma(*a, **k)
ops.conv3d_mfma = f
module._plans.clear()
for use_graph in (False, True, True, True):
    module.use_graph = use_graph
    hist = module.propagate_white_noise(wn, nsteps=4, record_history=True).cpu()
    print("graph" if use_graph else "eager", [float(rel_l2(hist[i], v["hist_heun_N4_f32"][i])) for i in range(hist.shape[0])], flush=True)
ops.conv3d_mfma = orig_mfma
print("--- attention arena: skip (monkeypatch _AmaxArena.of to reduce into fresh rows is not possible under capture); instead disable attention amax")
import diffsci_amd.models.nets.punetg as P
oa = net._amax_kw
net._amax_kw = lambda **kw: {"in_amax": ops.NORMALISED} if "in_amax" in kw else {}
module._plans.clear()
for use_graph in (False, True, True, True):
    module.use_graph = use_graph
    hist = module.propagate_white_noise(wn, nsteps=4, record_history=True).cpu()
    print("graph" if use_graph else "eager", [float(rel_l2(hist[i], v["hist_heun_N4_f32"][i])) for i in range(hist.shape[0])], flush=True)
