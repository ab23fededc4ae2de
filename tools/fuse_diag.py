"""Parity of fused vs standalone norms against the reference goldens (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffsci_amd.models as M
from tests.golden_util import load, rel_l2
dev = torch.device("cuda:0")

def run(name, make, fn):
    for fuse in (False, True):
        net, module = make()
        net.fuse_norm = fuse
        print(f"{name:28s} fuse={fuse}:", "  ".join(f"{k}={e:.2e}" for k, e in fn(net, module)))

v, sd = load("punetg8_porosity")
def mk():
    net = M.PUNetG(M.PUNetGConfig(model_channels=8, input_channels=4, output_channels=4),
                   conditional_embedding=M.nets.PorosityEmbedder(dembed=8))
    net.load_state_dict(sd)
    return net, M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev)
def f(net, module):
    y = {"porosity": v["porosity"].to(dev)}
    wn = v["white_noise"].to(dev)
    h = module.propagate_white_noise(wn, y=y, guidance=2.0, nsteps=4, record_history=True).cpu()
    o = module.propagate_white_noise(wn, y=y, guidance=1.0, nsteps=4).cpu()
    return [("g2_hist", rel_l2(h, v["hist_cfg_g2_N4_f32"])), ("g1", rel_l2(o, v["out_cond_g1_N4_f32"]))]
run("porosity", mk, f)

vt, _ = load("punetg8_traj")
vf, sdf = load("punetg8_forward")
def mk2():
    net = M.PUNetG(M.PUNetGConfig(model_channels=8)); net.load_state_dict(sdf)
    return net, M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev)
def f2(net, module):
    wn = vt["white_noise"].to(dev)
    o = module.propagate_white_noise(wn, nsteps=18).cpu()
    fw = net(vf["x"].to(dev), vf["t"].to(dev)).cpu()
    return [("fwd", rel_l2(fw, vf["out_f32"])), ("fwd64", rel_l2(fw, vf["out_f64"])), ("ref32v64", rel_l2(vf["out_f32"], vf["out_f64"])),
            ("N18", rel_l2(o, vt["out_heun_N18_f32"])), ("N18_64", rel_l2(o, vt["out_heun_N18_f64"])),
            ("ref", rel_l2(vt["out_heun_N18_f32"], vt["out_heun_N18_f64"]))]
run("punetg8", mk2, f2)

for skip in ("concat", "add"):
    va, sda = load(f"adm8_{skip}")
    def mk3():
        net = M.ADM(M.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16, skip_integration_type=skip))
        net.load_state_dict(sda)
        return net, M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev)
    def f3(net, module):
        fw = net(va["x"].to(dev), va["t"].to(dev)).cpu()
        o = module.propagate_white_noise(va["white_noise"].to(dev), nsteps=6).cpu()
        return [("fwd", rel_l2(fw, va["out_f32"])), ("fwd64", rel_l2(fw, va["out_f64"])), ("ref32v64", rel_l2(va["out_f32"], va["out_f64"])),
                ("N6", rel_l2(o, va["out_heun_N6_f32"]))]
    run("adm8_" + skip, mk3, f3)
