"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Container-only (the reference does not exist on the GPU box).  Run:
    python oracle/tools/make_golden.py
The fixtures are data only: inputs, state_dicts and the reference's outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import diffsci.models as M  # noqa: E402
import diffsci.data  # noqa: E402

OUT = os.path.join(HERE, "..", "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def npz(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def sd_arrays(sd, prefix="sd/"):
    return {prefix + k: v for k, v in sd.items()}


class RandnRecorder:
    """Record every torch.randn_like draw (the integrators' only RNG use)."""
    def __init__(self):
        self.draws = []
        self._orig = torch.randn_like

    def __enter__(self):
        def rec(x, *a, **k):
            e = self._orig(x, *a, **k)
            self.draws.append(e.clone())
            return e
        torch.randn_like = rec
        return self

    def __exit__(self, *a):
        torch.randn_like = self._orig


# ---------------------------------------------------------------- 1. schedule tables
def schedule():
    s = M.EDMScheduler()
    arrs = {"cpu_capability": np.array(torch.backends.cpu.get_cpu_capability())}
    for n in (2, 5, 10, 18, 50, 100, 256):
        arrs[f"steps_{n}"] = s.create_steps(n + 1)
    tq = torch.tensor([80.0, 57.5, 10.0, 1.0, 0.3, 0.01, 0.002])
    for n in (19, 51):
        arrs[f"step_from_time_{n}"] = s.step_from_time(tq, n)
    arrs["step_from_time_t"] = tq
    p = M.EDMPreconditioner()
    for B in (1, 4, 64):
        for n in (18, 50):
            t = s.create_steps(n + 1)[:-1]
            rows = []
            for ti in t:
                sig = ti * torch.ones(B)
                rows.append(torch.stack([p.skip_scaling(sig)[0], p.output_scaling(sig)[0],
                                         p.input_scaling(sig)[0], p.noise_conditioner(sig)[0],
                                         p.skip_scaling(sig)[-1], p.output_scaling(sig)[-1],
                                         p.input_scaling(sig)[-1], p.noise_conditioner(sig)[-1]]))
            arrs[f"precond_B{B}_N{n}"] = torch.stack(rows)
    npz("schedule", **arrs)


# ---------------------------------------------------------------- 2. toy / analytic + MLP (config 1)
def toy():
    torch.manual_seed(0)
    ds = diffsci.data.ZeroDataset(num_samples=8, shape=[2])
    gs = diffsci.data.ZeroMeanGaussianDataset(num_samples=8, shape=[2], scale=0.7)
    sch = M.EDMScheduler()
    x = torch.randn(8, 2)
    arrs = {"x": x}
    for nm, d in (("zero", ds), ("gauss", gs)):
        for integ in ("heun", "euler"):
            sch.set_temporary_integrator(integ)
            arrs[f"{nm}_{integ}_N18"] = sch.propagate_backward(x * 80.0, d.gradlogprob, 18, record_history=True)
            sch.unset_temporary_integrator()
    npz("toy_analytic", **arrs)

    torch.manual_seed(0)
    model = M.MLPUncond(2, [20])
    config = M.KarrasModuleConfig.from_edm()
    module = M.KarrasModule(model, config)
    torch.manual_seed(1)
    wn = torch.randn(8, 2)
    arrs = dict(sd_arrays(model.state_dict()), white_noise=wn)
    for integ in ("heun", "euler"):
        arrs[f"hist_{integ}_N18_f32"] = module.propagate_white_noise(wn, nsteps=18, record_history=True, integrator=integ)
    with RandnRecorder() as rec:
        arrs["hist_karras_N18_f32"] = module.propagate_white_noise(wn, nsteps=18, record_history=True, integrator="karras")
    arrs["eps_karras_N18"] = torch.stack(rec.draws)
    config.noisescheduler.langevin_const = 0.7
    with RandnRecorder() as rec:
        # EM lives in scheduler.stochastic_integrator; route through propagate_backward(stochastic=True)
        y = None

        def rhs(xx, sigma):
            return module.get_score(xx, sigma, y, 1.0)
        with torch.inference_mode():
            arrs["hist_em_N18_f32"] = config.noisescheduler.propagate_backward(
                wn * 80.0, rhs, 18, record_history=True, stochastic=True)
    arrs["eps_em_N18"] = torch.stack(rec.draws)
    arrs["em_langevin_const"] = np.array(0.7)
    config.noisescheduler.langevin_const = 1.0
    module.double()
    arrs["hist_heun_N18_f64"] = module.propagate_white_noise(wn.double(), nsteps=18, record_history=True, integrator="heun")
    npz("mlp_cfg1", **arrs)


# ---------------------------------------------------------------- 3. PUNetG layers + forward + trajectories
def punetg():
    torch.manual_seed(0)
    cfg = M.nets.PUNetGConfig(model_channels=8)
    net = M.nets.PUNetG(cfg).eval()
    # make norm affines non-trivial so the fixtures exercise them
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k:
                v.add_(0.25 * torch.randn_like(v))
            if k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
                v.add_(0.1 * torch.randn_like(v))
    sd = net.state_dict()
    torch.manual_seed(2)
    x = torch.randn(2, 1, 32, 32)
    t = torch.tensor([0.3, -1.7])
    arrs = dict(sd_arrays(sd), x=x, t=t)
    with torch.inference_mode():
        arrs["out_f32"] = net(x, t)
        # layer-level goldens
        h = net.convin(x)
        te = net.time_projection(t)
        arrs["convin"] = h
        arrs["te"] = te
        blk = net.downward_blocks[0][0]
        arrs["gn1_silu"] = blk.act(blk.gnorm1(h))
        arrs["timeshift"] = blk.timeblock(te)
        y1 = blk.conv1(blk.act(blk.gnorm1(h))) + blk.timeblock(te)
        arrs["conv1_shift"] = y1
        arrs["rms_silu"] = blk.act(blk.gnorm2(y1))
        r = blk(h, te)
        arrs["resblock"] = r
        arrs["down"] = net.downsamplers[0](r)
        hb = torch.randn(2, 32, 8, 8)
        arrs["attn_in"] = hb
        arrs["attn_out"] = net.attn_block[0](hb)
        arrs["up"] = net.upsamplers[0](hb)
    net64 = M.nets.PUNetG(cfg).double()
    net64.load_state_dict({k: v.double() for k, v in sd.items()})
    net64.eval()
    with torch.inference_mode():
        arrs["out_f64"] = net64(x.double(), t.double())
    npz("punetg8_forward", **arrs)

    # trajectories through KarrasModule
    config = M.KarrasModuleConfig.from_edm()
    module = M.KarrasModule(net, config).eval()
    torch.manual_seed(3)
    wn = torch.randn(2, 1, 32, 32)
    arrs = dict(white_noise=wn)   # state_dict shared with punetg8_forward
    arrs["hist_heun_N6_f32"] = module.propagate_white_noise(wn, nsteps=6, record_history=True)
    arrs["out_heun_N18_f32"] = module.propagate_white_noise(wn, nsteps=18)
    arrs["hist_euler_N6_f32"] = module.propagate_white_noise(wn, nsteps=6, record_history=True, integrator="euler")
    with RandnRecorder() as rec:
        arrs["hist_karras_N6_f32"] = module.propagate_white_noise(wn, nsteps=6, record_history=True, integrator="karras")
    arrs["eps_karras_N6"] = torch.stack(rec.draws)
    with RandnRecorder() as rec:
        def rhs(xx, sigma):
            return module.get_score(xx, sigma, None, 1.0)
        with torch.inference_mode():
            arrs["hist_em_N6_f32"] = config.noisescheduler.propagate_backward(
                wn * 80.0, rhs, 6, record_history=True, stochastic=True)
    arrs["eps_em_N6"] = torch.stack(rec.draws)
    # sample(): CPU-generator noise + minibatching (karrasmodule.py:817-838)
    torch.manual_seed(5)
    arrs["sample_seed5_n3_mb2_N4"] = module.sample(3, [1, 32, 32], nsteps=4, maximum_batch_size=2)
    mod64 = M.KarrasModule(net64, M.KarrasModuleConfig.from_edm()).eval()
    arrs["hist_heun_N6_f64"] = mod64.propagate_white_noise(wn.double(), nsteps=6, record_history=True)
    arrs["out_heun_N18_f64"] = mod64.propagate_white_noise(wn.double(), nsteps=18)
    torch.manual_seed(5)          # the same draws sample(3, maximum_batch_size=2) makes: 2 then 1
    wn5 = torch.cat([torch.randn(2, 1, 32, 32), torch.randn(1, 1, 32, 32)])
    arrs["sample_seed5_white_noise"] = wn5
    arrs["sample_seed5_n3_N4_f64"] = mod64.propagate_white_noise(wn5.double(), nsteps=4)
    npz("punetg8_traj", **arrs)

    # classifier-free guidance with a conditional embedding (config 5 shape of the path)
    torch.manual_seed(4)
    emb = torch.nn.Embedding(4, 8)
    cnet = M.nets.PUNetG(cfg, conditional_embedding=emb).eval()
    cnet.load_state_dict({**sd, "conditional_embedding.weight": emb.weight.detach()})
    cmod = M.KarrasModule(cnet, M.KarrasModuleConfig.from_edm(), conditional=True).eval()
    y = torch.tensor(2)
    arrs = {"emb_weight": emb.weight.detach(), "y": y, "white_noise": wn}
    arrs["hist_cfg_g2_N4_f32"] = cmod.propagate_white_noise(wn, y=y, guidance=2.0, nsteps=4, record_history=True)
    arrs["out_cond_g1_N4_f32"] = cmod.propagate_white_noise(wn, y=y, guidance=1.0, nsteps=4)
    arrs["out_cond_g0_N4_f32"] = cmod.propagate_white_noise(wn, y=y, guidance=0.0, nsteps=4)
    npz("punetg8_cfg", **arrs)


# ---------------------------------------------------------------- 4. ADM (config 3 family, tiny)
def langevin():
    """Euler-Maruyama with the runtime knobs langevin_const / langevin_interval (schedulers.py:219-245)."""
    import diffsci.data
    torch.manual_seed(80)
    gs = diffsci.data.ZeroMeanGaussianDataset(num_samples=8, shape=[2], scale=0.7)
    sch = M.EDMScheduler()
    sch.langevin_const = 0.6
    sch.langevin_interval = (0.3, 25.0)
    x = torch.randn(8, 2) * 80.0
    with RandnRecorder() as rec:
        h = sch.propagate_backward(x, gs.gradlogprob, 12, record_history=True, stochastic=True)
    npz("em_interval", x=x, hist=h, eps=torch.stack(rec.draws), langevin_const=np.array(0.6),
        langevin_interval=np.array([0.3, 25.0]))


def punetgcond():
    """PUNetGCond: channel-concatenated field conditioning (punetg.py:706-735), guidance 1."""
    torch.manual_seed(70)
    cfg = M.nets.PUNetGConfig(model_channels=8, input_channels=3, output_channels=1)
    net = M.nets.PUNetGCond(cfg, channel_conditional_items=["field"]).eval()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k:
                v.add_(0.25 * torch.randn_like(v))
    torch.manual_seed(71)
    x, t = torch.randn(2, 1, 32, 32), torch.tensor([0.2, -0.7])
    field = torch.randn(1, 2, 32, 32)
    arrs = dict(sd_arrays(net.state_dict()), x=x, t=t, field=field)
    with torch.inference_mode():
        arrs["out_f32"] = net(x, t, {"field": field})
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).eval()
    wn = torch.randn(2, 1, 32, 32)
    arrs["white_noise"] = wn
    # un-batched y (the module unsqueezes it): [2, 32, 32]
    arrs["hist_heun_N4_f32"] = module.propagate_white_noise(wn, y={"field": field[0]}, nsteps=4, record_history=True)
    npz("punetg8_cond", **arrs)


def inpaint():
    """inpaint / repaint / forward propagation / image interpolation."""
    import diffsci.data
    torch.manual_seed(30)
    gs = diffsci.data.ZeroMeanGaussianDataset(num_samples=8, shape=[1, 8, 8], scale=0.7)
    sch = M.EDMScheduler()
    N = 6
    x = torch.randn(3, 1, 8, 8) * 80.0
    yh = torch.randn(N + 1, 3, 1, 8, 8) * torch.linspace(0.5, 80.0, N + 1).view(-1, 1, 1, 1, 1)
    mask = (torch.rand(1, 8, 8) < 0.4).float()
    arrs = dict(x=x, y_hist=yh, mask=mask)
    arrs["sched_inpaint_hist"] = sch.inpaint(x, yh, mask, gs.gradlogprob, N, record_history=True)
    arrs["sched_inpaint_out"] = sch.inpaint(x, yh, mask, gs.gradlogprob, N)
    with RandnRecorder() as rec:
        arrs["sched_repaint_hist"] = sch.repaint(x, yh, mask, gs.gradlogprob, N, rsteps=2, nresamples=2, record_history=True)
    arrs["sched_repaint_eps"] = torch.stack(rec.draws)
    arrs["sched_forward_heun_hist"] = sch.propagate_forward(x / 80.0, gs.gradlogprob, N, record_history=True)
    # module level, PUNetG-8 (weights of the punetg8_forward fixture)
    z = np.load(os.path.join(OUT, "punetg8_forward.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    net = M.nets.PUNetG(M.nets.PUNetGConfig(model_channels=8)).eval()
    net.load_state_dict(sd)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).eval()
    torch.manual_seed(31)
    x0 = torch.randn(2, 1, 32, 32) * 0.5
    m2 = (torch.rand(1, 32, 32) < 0.5).float()
    arrs.update(x0=x0, mask2=m2)
    arrs["toward_noise_heun_N4"] = module.propagate_toward_noise(x0, nsteps=4, record_history=True)
    with RandnRecorder() as rec:
        fh = module.propagate_toward_noise(x0, nsteps=4, record_history=True, stochastic_integration=True)
    arrs["toward_noise_em_N4"] = fh
    arrs["toward_noise_em_eps"] = torch.stack(rec.draws)
    noise = torch.randn(2, 1, 32, 32) * 80.0
    arrs["inpaint_noise"] = noise
    arrs["module_inpaint_hist"] = module.propagate_inpaint_toward_sample(noise, fh, m2, record_history=True)
    arrs["interp_N4_n3"] = module.interpolate_images(x0[0], x0[1], 3, jitter=None, nsteps=4)
    npz("inpaint8", **arrs)


def vpve():
    """VP / VE parameterisations (schedulers, preconditioners, the non-constant-scaling rhs)."""
    import diffsci.data
    torch.manual_seed(40)
    gs = diffsci.data.ZeroMeanGaussianDataset(num_samples=8, shape=[2], scale=0.7)
    z = np.load(os.path.join(OUT, "punetg8_forward.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    x = torch.randn(8, 2)
    wn = torch.randn(2, 1, 32, 32)
    arrs = dict(x=x, white_noise=wn)
    # VP with M = 2: c_noise = (M-1)*t stays O(1); the default M = 1000 feeds the random-init Fourier
    # features arguments of ~1e5 rad, where one ulp of c_noise changes the trajectory (ill-conditioned
    # for the reference's own fp32 as well)
    for tag, mk in (("vp", lambda: M.KarrasModuleConfig.from_vp(M=2)), ("ve", M.KarrasModuleConfig.from_ve)):
        cfg = mk()
        sch = cfg.noisescheduler
        for n in (4, 6, 18):
            arrs[f"{tag}_steps_{n}"] = sch.create_steps(n + 1)
        arrs[f"{tag}_maximum_scale"] = np.array(sch.maximum_scale, dtype=np.float64)
        sig = torch.tensor([0.05, 0.7, 3.0, 40.0])
        pc = cfg.preconditioner
        arrs[f"{tag}_precond"] = torch.stack([pc.skip_scaling(sig), pc.output_scaling(sig), pc.input_scaling(sig),
                                              pc.noise_conditioner(sig)])
        for integ in ("heun", "euler"):
            sch.set_temporary_integrator(integ)
            arrs[f"{tag}_toy_{integ}_N18"] = sch.propagate_backward(x * sch.maximum_scale, gs.gradlogprob, 18,
                                                                    record_history=True)
            sch.unset_temporary_integrator()
        with RandnRecorder() as rec:
            arrs[f"{tag}_toy_em_N6"] = sch.propagate_backward(x * sch.maximum_scale, gs.gradlogprob, 6,
                                                              record_history=True, stochastic=True)
        arrs[f"{tag}_toy_em_eps"] = torch.stack(rec.draws)
        arrs[f"{tag}_toy_forward_N6"] = sch.propagate_forward(x * 0.3, gs.gradlogprob, 6, record_history=True)
        net = M.nets.PUNetG(M.nets.PUNetGConfig(model_channels=8)).eval()
        net.load_state_dict(sd)
        module = M.KarrasModule(net, cfg).eval()
        arrs[f"{tag}_punetg_heun_N6"] = module.propagate_white_noise(wn, nsteps=6, record_history=True)
        net64 = M.nets.PUNetG(M.nets.PUNetGConfig(model_channels=8)).double().eval()
        net64.load_state_dict({k: w.double() for k, w in sd.items()})
        arrs[f"{tag}_punetg_heun_N6_f64"] = M.KarrasModule(net64, mk()).eval().double().propagate_white_noise(
            wn.double(), nsteps=6, record_history=True)
        arrs[f"{tag}_punetg_euler_N6"] = module.propagate_white_noise(wn, nsteps=6, integrator="euler")
        if tag == "ve":
            with RandnRecorder() as rec:
                arrs["ve_punetg_karras_N4"] = module.propagate_white_noise(wn, nsteps=4, record_history=True,
                                                                           integrator="karras")
            arrs["ve_punetg_karras_eps"] = torch.stack(rec.draws)
        xs = torch.randn(2, 1, 32, 32) * 2.0
        sg = torch.tensor([0.3, 5.0])
        with torch.inference_mode():
            arrs[f"{tag}_xs"] = xs
            arrs[f"{tag}_score"] = module.get_score(xs, sg)
    npz("vpve8", **arrs)


def vp_karras():
    """The sigma-churn integrator on a NON-constant scaling (VP): x_hat = (s(t_hat)/s(t))*x + std*S_noise*eps
    (integrators.py:94-113), toy Gaussian score, recorded draws."""
    import diffsci.data
    torch.manual_seed(45)
    gs = diffsci.data.ZeroMeanGaussianDataset(num_samples=8, shape=[2], scale=0.7)
    sch = M.KarrasModuleConfig.from_vp(M=2).noisescheduler
    x = torch.randn(8, 2)
    arrs = dict(x=x, steps_6=sch.create_steps(7))
    sch.set_temporary_integrator("karras")
    with RandnRecorder() as rec:
        arrs["hist_N6"] = sch.propagate_backward(x * sch.maximum_scale, gs.gradlogprob, 6, record_history=True)
    arrs["eps"] = torch.stack(rec.draws)
    npz("vp_karras", **arrs)


def circular():
    """PUNetG with convolution_type='circular' (periodic padding in every 3x3 convolution)."""
    torch.manual_seed(50)
    cfg = M.nets.PUNetGConfig(model_channels=8, convolution_type="circular")
    net = M.nets.PUNetG(cfg).eval()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k or k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
                v.add_(0.25 * torch.randn_like(v))
    sd = net.state_dict()
    torch.manual_seed(51)
    x = torch.randn(2, 1, 32, 32)
    t = torch.tensor([0.3, -0.9])
    arrs = dict(sd_arrays(sd), x=x, t=t)
    with torch.inference_mode():
        arrs["out_f32"] = net(x, t)
        arrs["convin"] = net.convin(x)
        arrs["down0"] = net.downsamplers[0](arrs["convin"])
        arrs["up1"] = net.upsamplers[1](net.downsamplers[0](arrs["convin"]))
    net64 = M.nets.PUNetG(cfg).double().eval()
    net64.load_state_dict({k: v.double() for k, v in sd.items()})
    with torch.inference_mode():
        arrs["out_f64"] = net64(x.double(), t.double())
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).eval()
    wn = torch.randn(2, 1, 32, 32)
    arrs["white_noise"] = wn
    arrs["hist_heun_N6_f32"] = module.propagate_white_noise(wn, nsteps=6, record_history=True)
    npz("punetg8_circular", **arrs)
    # bias=False: no convolution biases, a constant-one channel appended to the input (punetg.py:390-394)
    torch.manual_seed(53)
    bnet = M.nets.PUNetG(M.nets.PUNetGConfig(model_channels=8, bias=False)).eval()
    with torch.no_grad():
        for k, v in bnet.state_dict().items():
            if "gnorm" in k:
                v.add_(0.25 * torch.randn_like(v))
    with torch.inference_mode():
        npz("punetg8_nobias", **dict(sd_arrays(bnet.state_dict()), x=x, t=t, out_f32=bnet(x, t)))
    # ADM: the blocks' convolutions become circular, input / output layers stay zero padded
    torch.manual_seed(52)
    acfg = M.nets.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16, convolution_type="circular")
    anet = M.nets.ADM(acfg).eval()
    with torch.no_grad():
        for k, v in anet.state_dict().items():
            if "norm" in k:
                v.add_(0.25 * torch.randn_like(v))
    with torch.inference_mode():
        npz("adm8_circular", **dict(sd_arrays(anet.state_dict()), x=x, t=t, out_f32=anet(x, t)))


def si():
    """SIModule (stochastic interpolants / flow matching) sampler."""
    z = np.load(os.path.join(OUT, "punetg8_forward.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    zc = np.load(os.path.join(OUT, "punetg8_cfg.npz"))
    torch.manual_seed(60)
    noise = torch.randn(2, 1, 32, 32)
    arrs = dict(noise=noise)

    def make(cond=False):
        emb = None
        if cond:
            emb = torch.nn.Embedding(4, 8)
            emb.weight.data.copy_(torch.from_numpy(zc["emb_weight"]))
        net = M.nets.PUNetG(M.nets.PUNetGConfig(model_channels=8), conditional_embedding=emb).eval()
        net.load_state_dict(sd, strict=False)
        return net
    ts = torch.linspace(1, 0, 6)
    for tag, kw in (("linear_identity", dict(scheduler="linear")),
                    ("edm_edm", dict(scheduler="edm", precondition_fn="edm")),
                    ("cosine_edm_norm2", dict(scheduler="cosine", precondition_fn="edm", initial_norm=2.0))):
        mod = M.SIModule(M.SIModuleConfig(**kw), make()).eval()
        with torch.inference_mode():
            arrs[f"{tag}_sample_N6"] = mod.sample(2, [1, 32, 32], nsteps=6, orig_noise=noise)
            h = mod.integrate_flow_field(noise * mod.config.sigma_fn(ts[0]), ts, return_history=True)
            arrs[f"{tag}_hist_N6"] = torch.stack([x for _, x in h])
            tt = torch.tensor([0.4, 0.4])
            arrs[f"{tag}_flow"] = mod.get_flow_field(noise, tt)
            arrs[f"{tag}_score"] = mod.get_score_field(noise, tt)
            # (integrate_on_sigma=True raises in the reference for image batches: flow_field / sigma_dot, flowfield.py:455-457,
            #  divides [B,C,H,W] by [B] without broadcasting)
    cmod = M.SIModule(M.SIModuleConfig(scheduler="linear"), make(cond=True)).eval()
    y = torch.tensor(2)
    import warnings
    with torch.inference_mode(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        arrs["cfg_y"] = y
        arrs["cfg_g2_sample_N6"] = cmod.sample(2, [1, 32, 32], y=y, guidance=2.0, nsteps=6, orig_noise=noise)
        arrs["cfg_g1_sample_N6"] = cmod.sample(2, [1, 32, 32], y=y, guidance=1.0, nsteps=6, orig_noise=noise)
    cmod2 = M.SIModule(M.SIModuleConfig(scheduler="edm", precondition_fn="edm"), make(cond=True)).eval()
    with torch.inference_mode(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        arrs["cfg_edm_g2_sample_N6"] = cmod2.sample(2, [1, 32, 32], y=y, guidance=2.0, nsteps=6, orig_noise=noise)
    npz("si8", **arrs)


def si_latent():
    """SIModule with a latent autoencoder and the batch-norm initial_norm (flowfield.py:300-345, 503-544, 742-747), plus
    its single-step entry point integration_step (:749-795).  Network: the latent8 fixture's (4 -> 4 channels)."""
    z = np.load(os.path.join(OUT, "latent8.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    net = M.nets.PUNetG(M.nets.PUNetGConfig(input_channels=4, output_channels=4, model_channels=8)).eval()
    net.load_state_dict(sd)
    torch.manual_seed(130)
    noise = torch.randn(2, 4, 16, 16)
    arrs = dict(noise=noise)
    mod = M.SIModule(M.SIModuleConfig(scheduler="linear", initial_norm=True, num_channels=4), net,
                     autoencoder=ToyAutoencoder()).eval()
    mod.initial_norm.running_mean = torch.tensor([0.3, -0.2, 0.05, 1.1])
    mod.initial_norm.running_var = torch.tensor([2.5, 0.4, 1.0, 0.09])
    arrs["sd_keys"] = np.array(sorted(k for k in mod.state_dict() if not k.startswith("model.")))
    ts = torch.linspace(1, 0, 5)
    with torch.inference_mode():
        arrs["sample_N5"] = mod.sample(2, [4, 16, 16], nsteps=5, orig_noise=noise, is_latent_shape=True)
        arrs["latents_N5"] = mod.sample(2, [4, 16, 16], nsteps=5, orig_noise=noise, is_latent_shape=True, return_latents=True)
        h = mod.integrate_flow_field(noise * mod.config.sigma_fn(ts[0]), ts, return_history=True)
        arrs["hist_N5"] = torch.stack([x for _, x in h])
        t0, t1 = torch.full((2,), 0.7), torch.full((2,), 0.45)
        arrs["step_euler"] = mod.integration_step(noise, t0, t1, method="euler")
        arrs["step_heun"] = mod.integration_step(noise, t0, t1, method="heun")
    npz("si8_latent", **arrs)


def si_inpaint():
    """SIModule.inpaint (flowfield.py:546-702): Euler-Maruyama steps with the known region re-imposed at every noise
    level, RePaint-style jumps, soft mask.  Every randn_like draw is recorded in order.  Network: punetg8_forward's."""
    z = np.load(os.path.join(OUT, "punetg8_forward.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    net = M.nets.PUNetG(M.nets.PUNetGConfig(model_channels=8)).eval()
    net.load_state_dict(sd)
    torch.manual_seed(140)
    x_orig = torch.randn(1, 32, 32) * 0.5
    mask = torch.zeros(1, 32, 32)
    mask[:, 8:24, 4:20] = 1.0
    noise0 = torch.randn(2, 1, 32, 32)
    arrs = dict(x_orig=x_orig, mask=mask, orig_noise=noise0)
    import warnings
    for tag, cfgkw, kw in (("hard", dict(scheduler="linear"), dict(nsteps=5)),
                           ("soft_jump", dict(scheduler="cosine", precondition_fn="edm", initial_norm=2.0),
                            dict(nsteps=5, mask_falloff=2, resample_steps=1, mask_start_t=0.8))):
        mod = M.SIModule(M.SIModuleConfig(**cfgkw), net).eval()
        with RandnRecorder() as rec, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = mod.inpaint(x_orig, mask, nsamples=2, orig_noise=noise0, **kw)
        arrs[tag + "_out"] = out
        arrs[tag + "_ndraws"] = np.array(len(rec.draws))
        for i, d in enumerate(rec.draws):
            arrs[f"{tag}_eps{i:02d}"] = d
        if kw.get("mask_falloff"):
            arrs[tag + "_soft_mask"] = mod._create_soft_mask(mask, kw["mask_falloff"])
    npz("si8_inpaint", **arrs)


def si_custom_precondition(model, x, t, y=None):
    """A user precondition callable (ours; tests define the same function)."""
    return 0.5 * model(x, t, y=y) - 0.1 * x


def si_generic():
    """SIModule with preconditioners that cannot be tabulated: autonomous flows (model(x, y=y), flowfield.py:147-165) and a
    user callable; per-sample times in get_flow_field / get_score_field.  Network: punetg8_forward's."""
    z = np.load(os.path.join(OUT, "punetg8_forward.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    net = M.nets.PUNetG(M.nets.PUNetGConfig(model_channels=8)).eval()
    net.load_state_dict(sd)
    torch.manual_seed(170)
    noise = torch.randn(2, 1, 32, 32)
    arrs = dict(noise=noise)
    ts = torch.linspace(1, 0, 5)
    cases = (("auto_identity", dict(scheduler="linear", autonomous_flow=True)),
             ("auto_edm", dict(scheduler="cosine", autonomous_flow=True, precondition_fn="edm")),
             ("callable", dict(scheduler="linear", precondition_fn=si_custom_precondition)))
    for tag, kw in cases:
        mod = M.SIModule(M.SIModuleConfig(**kw), net).eval()
        with torch.inference_mode():
            arrs[tag + "_sample_N5"] = mod.sample(2, [1, 32, 32], nsteps=5, orig_noise=noise)
            h = mod.integrate_flow_field(noise * mod.config.sigma_fn(ts[0]), ts, return_history=True)
            arrs[tag + "_hist_N5"] = torch.stack([x for _, x in h])
    mod = M.SIModule(M.SIModuleConfig(scheduler="cosine", precondition_fn="edm"), net).eval()
    tt = torch.tensor([0.4, 0.7])
    with torch.inference_mode():
        arrs["persample_t"] = tt
        arrs["persample_flow"] = mod.get_flow_field(noise, tt)
        arrs["persample_score"] = mod.get_score_field(noise, tt)
    npz("si8_generic", **arrs)


def porosity():
    """BASELINE config 5's shape of the path: 4-channel conditional PUNetG with the in-repo dict-style
    PorosityEmbedder (nets/embedder.py:198-229), classifier-free guidance, un-batched dict y."""
    torch.manual_seed(20)
    cfg = M.nets.PUNetGConfig(model_channels=8, input_channels=4, output_channels=4)
    from diffsci.models.nets.embedder import PorosityEmbedder
    emb = PorosityEmbedder(dembed=8)
    net = M.nets.PUNetG(cfg, conditional_embedding=emb).eval()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k or k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
                v.add_(0.25 * torch.randn_like(v))
    sd = net.state_dict()
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).eval()
    torch.manual_seed(21)
    wn = torch.randn(2, 4, 32, 32)
    y = {"porosity": torch.tensor([0.37])}
    arrs = dict(sd_arrays(sd), white_noise=wn, porosity=y["porosity"])
    with torch.inference_mode():
        arrs["ye"] = emb({"porosity": y["porosity"].unsqueeze(0)})
        yb = {"porosity": torch.tensor([[0.1], [0.9]])}
        arrs["ye_batch"] = emb(yb)
        arrs["porosity_batch"] = yb["porosity"]
    arrs["hist_cfg_g2_N4_f32"] = module.propagate_white_noise(wn, y=y, guidance=2.0, nsteps=4, record_history=True)
    arrs["out_cond_g1_N4_f32"] = module.propagate_white_noise(wn, y=y, guidance=1.0, nsteps=4)
    npz("punetg8_porosity", **arrs)


def adm():
    for skip in ("concat", "add"):
        torch.manual_seed(10)
        cfg = M.nets.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16,
                               skip_integration_type=skip)
        net = M.nets.ADM(cfg).eval()
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if "norm" in k or k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
                    v.add_(0.25 * torch.randn_like(v))
        sd = net.state_dict()
        torch.manual_seed(11)
        x = torch.randn(2, 1, 32, 32)
        t = torch.tensor([0.4, -1.1])
        arrs = dict(sd_arrays(sd), x=x, t=t)
        with torch.inference_mode():
            arrs["out_f32"] = net(x, t)
            te = net.time_embedding(t, None)
            arrs["te"] = te
            h = net.input_layer(x)
            arrs["stem"] = h
            blk = net.encoder.layers[0].input_blocks[0]
            arrs["enc00"] = blk(h, te)
            blk = net.encoder.layers[0].input_blocks[1]
            arrs["enc01_down"] = blk(arrs["enc00"], te)
        net64 = M.nets.ADM(cfg).double()
        net64.load_state_dict({k: v.double() for k, v in sd.items()})
        with torch.inference_mode():
            arrs["out_f64"] = net64.eval()(x.double(), t.double())
        module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).eval()
        torch.manual_seed(12)
        wn = torch.randn(2, 1, 32, 32)
        arrs["white_noise"] = wn
        arrs["out_heun_N6_f32"] = module.propagate_white_noise(wn, nsteps=6)
        with RandnRecorder() as rec:
            arrs["hist_karras_N4_f32"] = module.propagate_white_noise(wn, nsteps=4, record_history=True,
                                                                      integrator="karras")
        arrs["eps_karras_N4"] = torch.stack(rec.draws)
        npz(f"adm8_{skip}", **arrs)


def adm_blocks():
    """ADM residual blocks driven on their own, on fields and on volumes, with the shapes of the reference's own test
    (tests/test_adm.py:7-70: cin 16, cout 32, cembed 24, 14^2 / 14^3 inputs)."""
    A = M.nets.adm
    cases = {
        "enc2d": (A.ADMEncoderBlock, dict(), (1, 16, 14, 14), None),
        "enc2d_down": (A.ADMEncoderBlock, dict(has_downsample=True), (1, 16, 14, 14), None),
        "enc3d": (A.ADMEncoderBlock, dict(dimension=3), (1, 16, 14, 14, 14), None),
        "enc3d_full": (A.ADMEncoderBlock, dict(has_residual=True, has_attn=True, has_downsample=True, attn_residual=True,
                                               dimension=3), (2, 16, 14, 14, 14), None),
        "dec2d_skip": (A.ADMDecoderBlock, dict(channels_skip=12, has_residual=True, has_attn=True, has_upsample=True),
                       (1, 16, 14, 14), (1, 12, 14, 14)),
        "dec3d_skip_add": (A.ADMDecoderBlock, dict(channels_skip=16, has_residual=True, has_upsample=True, dimension=3,
                                                   skip_integration_type="add", first_norm="GroupRMS", second_norm="GroupLN"),
                           (2, 16, 6, 8, 10), (2, 16, 6, 8, 10)),
        "enc3d_circ": (A.ADMEncoderBlock, dict(has_residual=True, has_downsample=True, dimension=3, conv_type="circular"),
                       (1, 16, 8, 8, 8), None),
    }
    arrs = {}
    for i, (tag, (cls, kw, xs, ss)) in enumerate(cases.items()):
        torch.manual_seed(300 + i)
        blk = cls(16, 32, 24, **kw).eval()
        with torch.no_grad():
            for k, v in blk.state_dict().items():
                if "norm" in k or k.endswith("bias"):
                    v.add_(0.25 * torch.randn_like(v))
        x, te = torch.randn(*xs), torch.randn(xs[0], 24)
        skip = None if ss is None else torch.randn(*ss)
        with torch.inference_mode():
            out = blk(x, te, skip) if skip is not None else blk(x, te)
            b64 = cls(16, 32, 24, **kw).double().eval()
            b64.load_state_dict({k: v.double() for k, v in blk.state_dict().items()})
            out64 = b64(x.double(), te.double(), skip.double()) if skip is not None else b64(x.double(), te.double())
        for k, v in blk.state_dict().items():
            arrs[f"sd/{tag}/{k}"] = v.detach().numpy()
        arrs[f"{tag}/x"], arrs[f"{tag}/te"], arrs[f"{tag}/out_f32"], arrs[f"{tag}/out_f64"] = x, te, out, out64
        if skip is not None:
            arrs[f"{tag}/skip"] = skip
    npz("adm_blocks", **arrs)


def variants():
    """PUNetG with magnitude-preserving layers (convolution_type='mp': normedlayers.py, the in-house
    attention of attention.py:110-247) and with the other norm choices of ResnetBlockC (commonlayers.py:882-899)."""
    import warnings
    cases = {
        "mp": dict(convolution_type="mp"),
        "pix_ln": dict(first_resblock_norm="GroupPix", second_resblock_norm="GroupLN"),
        "none_rms_noaffine": dict(first_resblock_norm="none", second_resblock_norm="GroupRMS", affine_norm=False),
        "cosine": dict(attn_type="cosine"),
        # round 2: the fixed Fourier input embedding (only runnable with bias=False in the reference) and a shared
        # parameter-free extra_residual module
        "fourier_in": dict(in_embedding=True, bias=False),
        "extra_res": dict(),
        # kernel sizes other than 3 (punetg_config.py:19-25)
        "k5": dict(kernel_size=5, in_out_kernel_size=1, transition_kernel_size=5),
        "k7": dict(kernel_size=1, in_out_kernel_size=7, transition_kernel_size=7),
        # round 3: kernel sizes other than 3 WITH periodic padding (CircularConv2d pads k//2 circularly, commonlayers.py:918-971)
        "k5_circular": dict(kernel_size=5, in_out_kernel_size=5, transition_kernel_size=7, convolution_type="circular"),
    }
    only = os.environ.get("VARIANTS_ONLY")
    for i, (tag, over) in enumerate(cases.items()):
        if only and tag not in only.split(","):
            continue
        torch.manual_seed(70 + i)
        cfg = M.nets.PUNetGConfig(model_channels=8, **over)
        extra = dict(extra_residual=torch.nn.AvgPool2d(3, stride=1, padding=1)) if tag == "extra_res" else {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = M.nets.PUNetG(cfg, **extra).eval()
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if "gnorm" in k or k.endswith(".bias"):
                    v.add_(0.25 * torch.randn_like(v))
        sd = net.state_dict()
        torch.manual_seed(80 + i)
        x = torch.randn(2, 1, 32, 32)
        t = torch.tensor([0.3, -1.1])
        arrs = dict(sd_arrays(sd), x=x, t=t)
        with torch.inference_mode():
            arrs["out_f32"] = net(x, t)
            h = net.convin(x)
            arrs["convin"] = h
            blk = net.downward_blocks[0][0]
            arrs["resblock"] = blk(h, net.time_projection(t))
            hb = torch.randn(2, 32, 8, 8)
            arrs["attn_in"] = hb
            arrs["attn_out"] = net.attn_block[0](hb)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net64 = M.nets.PUNetG(cfg, **extra).double().eval()
        net64.load_state_dict({k: v.double() for k, v in sd.items()})
        with torch.inference_mode():
            arrs["out_f64"] = net64(x.double(), t.double())
        module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).eval()
        wn = torch.randn(2, 1, 32, 32)
        arrs["white_noise"] = wn
        arrs["hist_heun_N6_f32"] = module.propagate_white_noise(wn, nsteps=6, record_history=True)
        npz(f"punetg8_{tag}", **arrs)


def spatial_cond():
    """punetg.py:405-407 + commonlayers.py:537-546,838-869: a conditional embedding that is a FIELD ([B, C, H, W], here a
    user 1x1 convolution of a two-channel condition): the time embedding becomes a field, every block runs its time MLP per
    pixel and CornerPools the result to its own resolution."""
    import warnings
    torch.manual_seed(170)
    cfg = M.nets.PUNetGConfig(model_channels=8)
    emb = torch.nn.Conv2d(2, 8, kernel_size=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = M.nets.PUNetG(cfg, conditional_embedding=emb).eval()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k or k.endswith(".bias"):
                v.add_(0.25 * torch.randn_like(v))
    sd = net.state_dict()
    torch.manual_seed(171)
    x = torch.randn(2, 1, 32, 32)
    y = torch.randn(2, 2, 32, 32)
    t = torch.tensor([0.3, -1.1])
    arrs = dict(sd_arrays(sd), x=x, y=y, t=t)
    with torch.inference_mode():
        arrs["out_f32"] = net(x, t, y)
        arrs["out_uncond_f32"] = net(x, t)
        te = net.time_projection(t).reshape(2, 8, 1, 1) + emb(y)
        blk = net.downward_blocks[1][0]
        h = torch.randn(2, 16, 16, 16)
        arrs["resblock_in"] = h
        arrs["resblock_te"] = te
        arrs["resblock_l1"] = blk(h, te)                      # CornerPool2d(2) of the per-pixel shift
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net64 = M.nets.PUNetG(cfg, conditional_embedding=torch.nn.Conv2d(2, 8, kernel_size=1)).double().eval()
    net64.load_state_dict({k: v.double() for k, v in sd.items()})
    with torch.inference_mode():
        arrs["out_f64"] = net64(x.double(), t.double(), y.double())
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).eval()
    wn = torch.randn(2, 1, 32, 32)
    arrs["white_noise"] = wn
    # the sampler's condition is un-batched (karrasmodule.py:916-917 unsqueezes it): one field shared by the batch
    arrs["hist_heun_N4_g1_f32"] = module.propagate_white_noise(wn, y=y[0], guidance=1.0, nsteps=4, record_history=True)
    arrs["hist_heun_N4_g2_f32"] = module.propagate_white_noise(wn, y=y[0], guidance=2.0, nsteps=4, record_history=True)
    npz("punetg8_spatial_cond", **arrs)


def adm_norms():
    """ADM with the other norm choices of make_norm_layers (adm.py:385-406): RMS first / LN second (+FiLM), and LN / LN
    with affine_norm=False -- which the reference's ADM silently ignores (the flag is not forwarded to the blocks)."""
    for i, (tag, over) in enumerate((("rms_ln", dict(first_resblock_norm="GroupRMS", second_resblock_norm="GroupLN")),
                                     ("ln_ln_noaffine", dict(first_resblock_norm="GroupLN", second_resblock_norm="GroupLN",
                                                             affine_norm=False)),
                                     ("dec2", dict(decoder_type=2)))):
        if os.environ.get("ADM_ONLY") and tag not in os.environ["ADM_ONLY"].split(","):
            continue
        torch.manual_seed(100 + i)
        cfg = M.nets.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16, **over)
        net = M.nets.ADM(cfg).eval()
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if "norm" in k:
                    v.add_(0.25 * torch.randn_like(v))
        sd = net.state_dict()
        torch.manual_seed(110 + i)
        x = torch.randn(2, 1, 32, 32)
        t = torch.tensor([0.4, -1.1])
        arrs = dict(sd_arrays(sd), x=x, t=t)
        with torch.inference_mode():
            arrs["out_f32"] = net(x, t)
            te = net.time_embedding(t, None)
            h = net.input_layer(x)
            arrs["stem"] = h
            arrs["enc00"] = net.encoder.layers[0].input_blocks[0](h, te)
        net64 = M.nets.ADM(cfg).double().eval()
        net64.load_state_dict({k: v.double() for k, v in sd.items()})
        with torch.inference_mode():
            arrs["out_f64"] = net64(x.double(), t.double())
        npz(f"adm8_{tag}", **arrs)


def autoregressive():
    """KarrasModule.autoregressive_sample (LatentSpaceAutoregressive mixin,
    autoregressivesample.py:27-203): a forecast loop that calls sample() once per step with a sliding window of its own
    predictions as the channel condition y['y'].  Weights: the punetg8_cond fixture's network (same seed)."""
    torch.manual_seed(70)
    cfg = M.nets.PUNetGConfig(model_channels=8, input_channels=3, output_channels=1)
    net = M.nets.PUNetGCond(cfg, channel_conditional_items=["y"]).eval()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k:
                v.add_(0.25 * torch.randn_like(v))
    ref = np.load(os.path.join(OUT, "punetg8_cond.npz"))
    assert all(np.array_equal(ref["sd/" + k], v.numpy()) for k, v in net.state_dict().items())
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).eval()
    torch.manual_seed(120)
    y0 = torch.randn(2, 16, 16)                      # cond_time = 2 frames of 1 channel
    arrs = dict(y0=y0)
    import contextlib, io
    for tag, kw in (("plain", dict(nsamples=2)), ("batched", dict(nsamples=3, maximum_batch_size=2))):
        torch.manual_seed(121)                       # sample() draws its white noise from the CPU generator
        with contextlib.redirect_stdout(io.StringIO()):
            res = module.autoregressive_sample(latent_shape=[1, 16, 16], nsteps_forecast=5, cond_time=2,
                                               nsteps_diffusion=3, y={"y": y0.clone()}, y_already_encoded=True,
                                               return_intermediate=True, **kw)
        for k, v in res.items():
            arrs[f"{tag}_{k}"] = v
    # cond_time = 3 pins how the window is assembled while fewer than cond_time predictions exist (the loop re-reads
    # the window it wrote on the previous step as "initial conditions", autoregressivesample.py:147-161), with a
    # parameter-light stand-in network (ours; tests define the same class)
    tiny = M.KarrasModule(TinyCondNet(), M.KarrasModuleConfig.from_edm(), conditional=True).eval()
    y3 = torch.randn(6, 8, 8)                        # 3 frames x 2 channels
    arrs["tiny_y0"] = y3
    torch.manual_seed(122)
    with contextlib.redirect_stdout(io.StringIO()):
        res = tiny.autoregressive_sample(nsamples=2, latent_shape=[2, 8, 8], nsteps_forecast=6, cond_time=3,
                                         nsteps_diffusion=3, y={"y": y3.clone()}, y_already_encoded=True)
    arrs["tiny_forecasts"] = res["forecasts"]
    arrs["tiny_final_forecast"] = res["final_forecast"]
    npz("autoreg8", **arrs)


class TinyCondNet(torch.nn.Module):
    """model(x, c_noise, y): mixes x with a weighted sum of the condition frames, so every frame of the window matters."""

    def __init__(self):
        super().__init__()
        self.gain = torch.nn.Parameter(torch.tensor(0.3))

    def forward(self, x, t, y=None):
        f = y["y"].reshape(y["y"].shape[0], 3, 2, *y["y"].shape[2:])
        wts = torch.tensor([0.2, -0.5, 0.9]).view(1, 3, 1, 1, 1).to(x)
        return self.gain * x + (f * wts).sum(dim=1) + 0.1 * t.view(-1, 1, 1, 1)


SMALL_VOLUME_NET = dict(channel_expansion=[2], number_resnet_downward_block=1, number_resnet_upward_block=1,
                        number_resnet_attn_block=1, number_resnet_before_attn_block=1, number_resnet_after_attn_block=1)
VOLUME_CASES = (("3d", {}), ("3d_circular", dict(convolution_type="circular")),
                # round 3: kernel sizes other than 3 on volumes (1^3 in / out layers, 5^3 blocks and transitions; 5^3 everywhere, periodic)
                # -- on a two-level network with one block per stage: a 5^3 kernel has 125 taps, the fixture is mostly weights
                ("3d_k5", dict(kernel_size=5, in_out_kernel_size=1, transition_kernel_size=5, **SMALL_VOLUME_NET)),
                ("3d_k5_circular", dict(kernel_size=5, in_out_kernel_size=5, transition_kernel_size=5, convolution_type="circular",
                                        **SMALL_VOLUME_NET)))


def volumes_k5():
    volumes(only=("3d_k5", "3d_k5_circular"))


def volumes(only=None):
    """PUNetG with dimension = 3 (Conv3d / MaxPool3d / Upsample / ThreeDimensionalAttention), default
    and circular convolutions, 16^3 volumes."""
    for i, (tag, over) in enumerate(VOLUME_CASES):
        if only is not None and tag not in only:
            continue
        torch.manual_seed(150 + i)
        cfg = M.nets.PUNetGConfig(model_channels=8, dimension=3, **over)
        net = M.nets.PUNetG(cfg).eval()
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if "gnorm" in k or k.endswith("in_proj_bias") or k.endswith("out_proj.bias"):
                    v.add_(0.25 * torch.randn_like(v))
        sd = net.state_dict()
        torch.manual_seed(160 + i)
        x = torch.randn(2, 1, 16, 16, 16)
        t = torch.tensor([0.3, -1.1])
        arrs = dict(sd_arrays(sd), x=x, t=t)
        with torch.inference_mode():
            arrs["out_f32"] = net(x, t)
            h = net.convin(x)
            arrs["convin"] = h
            arrs["down0"] = net.downsamplers[0](h)
            if len(net.upsamplers) > 1:
                arrs["up1"] = net.upsamplers[1](arrs["down0"])
            arrs["resblock"] = net.downward_blocks[0][0](h, net.time_projection(t))
        net64 = M.nets.PUNetG(cfg).double().eval()
        net64.load_state_dict({k: v.double() for k, v in sd.items()})
        with torch.inference_mode():
            arrs["out_f64"] = net64(x.double(), t.double())
        if tag == "3d":
            module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).eval()
            wn = torch.randn(2, 1, 16, 16, 16)
            arrs["white_noise"] = wn
            arrs["hist_heun_N4_f32"] = module.propagate_white_noise(wn, nsteps=4, record_history=True)
        npz(f"punetg8_{tag}", **arrs)


class ToyAutoencoder(torch.nn.Module):
    """Parameter-free stand-in for a latent autoencoder (ours, not the reference's): 2x2 pixel-unshuffle with a gain.
    tests/test_gpu_sampler.py defines the same three lines."""

    def encode(self, x):
        return torch.nn.functional.pixel_unshuffle(x, 2) * 0.5

    def decode(self, z):
        return torch.nn.functional.pixel_shuffle(z * 2.0, 2)


def latent():
    """the latent boundary of KarrasModule -- autoencoder.encode/decode and the EDM batch-norm
    map around the loop (karrasmodule.py:842-851, 867-905, 1192-1241; aux_scripts/batchnorm.py:86-170)."""
    torch.manual_seed(90)
    cfg = M.nets.PUNetGConfig(input_channels=4, output_channels=4, model_channels=8)
    net = M.nets.PUNetG(cfg).eval()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k:
                v.add_(0.25 * torch.randn_like(v))
    arrs = dict(sd_arrays(net.state_dict()))
    torch.manual_seed(91)
    x = torch.randn(2, 1, 32, 32) * 0.7 + 0.2
    wn = torch.randn(2, 4, 16, 16)
    arrs.update(x=x, white_noise=wn)
    # the layer on its own, per-channel statistics and affine (the KarrasModule instance is DimensionAgnosticBatchNorm(
    # sigma=sigma_data): one broadcast statistic)
    from diffsci.models.karras import edmbatchnorm
    bn = edmbatchnorm.DimensionAgnosticBatchNorm(num_channels=4, affine=True, sigma=0.5).eval()
    bn.running_mean = torch.tensor([0.3, -0.2, 0.05, 1.1])
    bn.running_var = torch.tensor([2.5, 0.4, 1.0, 0.09])
    with torch.no_grad():
        bn.weight.copy_(torch.tensor([1.5, 0.7, -1.2, 0.9]))
        bn.bias.copy_(torch.tensor([0.1, -0.3, 0.0, 0.4]))
        arrs["bnC_in"] = wn
        arrs["bnC_normalize"] = bn.normalize(wn)
        arrs["bnC_unnormalize"] = bn.unnormalize(wn)
    for tag, mean, var in (("bn1", torch.tensor([0.3]), torch.tensor([2.5])),):
        module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(has_edm_batch_norm=True),
                                autoencoder=ToyAutoencoder()).eval()
        module.edm_batch_norm.running_mean = mean.clone()
        module.edm_batch_norm.running_var = var.clone()
        arrs[tag + "_mean"], arrs[tag + "_var"] = mean, var
        with torch.inference_mode():
            z = module.encode(x)
            arrs[tag + "_encode"] = z
            arrs[tag + "_decode_encode"] = module.decode(z)
        arrs[tag + "_sample_N4"] = module.propagate_white_noise(wn, nsteps=4, latent_shape=True)
        arrs[tag + "_latent_N4"] = module.propagate_white_noise(wn, nsteps=4, latent_shape=True, return_in_latent_space=True)
        arrs[tag + "_hist_N3"] = module.propagate_white_noise(wn, nsteps=3, latent_shape=True, record_history=True)
    # batch-norm only (no autoencoder)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(has_edm_batch_norm=True)).eval()
    module.edm_batch_norm.running_mean = torch.tensor([-0.4])
    module.edm_batch_norm.running_var = torch.tensor([0.6])
    arrs["plain_sample_N4"] = module.propagate_white_noise(wn, nsteps=4)
    arrs["plain_sd_keys"] = np.array(sorted(k for k in module.state_dict() if not k.startswith("model.")))
    npz("latent8", **arrs)


# ---------------------------------------------------------------- API surface of the mirrored classes
SURFACE_CLASSES = {      # module (under diffsci.models) -> classes
    "karras.karrasmodule": ["KarrasModule", "KarrasModuleConfig"],
    "karras.flowfield": ["SIModule", "SIModuleConfig", "SIScheduler"],
    "karras.schedulers": ["Scheduler", "EDMScheduler", "VPScheduler", "VEScheduler"],
    "karras.integrators": ["Integrator", "EulerIntegrator", "HeunIntegrator", "EulerMaruyamaIntegrator", "KarrasIntegrator"],
    "karras.preconditioners": ["KarrasPreconditioner", "EDMPreconditioner", "NullPreconditioner", "SR3Preconditioner",
                               "VPPreconditioner", "VEPreconditioner"],
    "karras.noisesamplers": ["NoiseSampler", "EDMNoiseSampler", "VPNoiseSampler", "VENoiseSampler", "UniformNoiseSampler"],
    "karras.schedulingfunctions": ["SchedulingFunctions", "EDMSchedulingFunctions", "VPSchedulingFunctions",
                                   "VESchedulingFunctions"],
    "karras.autoregressivesample": ["LatentSpaceAutoregressive"],
    "nets.punetg": ["PUNetG", "PUNetGCond"],
    "nets.punetg_config": ["PUNetGConfig"],
    "nets.adm": ["ADM", "ADMConfig", "ADMBaseBlock", "ADMEncoderBlock", "ADMDecoderBlock", "ADMEncoder", "ADMDecoder",
                 "ADMMiddleBlock", "ADMTimeEmbedding"],
    "nets.mlp": ["MLPUncond", "MLPCond"],
    "nets.embedder": ["PorosityEmbedder"],
}
SURFACE_FUNCTIONS = {"karras.integrators": ["name_to_integrator"], "karras.schedulingfunctions": ["name_to_scheduling_functions"]}


def _default_repr(v):
    import inspect
    if v is inspect.Parameter.empty:
        return "<required>"
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if isinstance(v, (list, tuple)) and all(x is None or isinstance(x, (bool, int, float, str)) for x in v):
        return list(v)
    return "<" + type(v).__name__ + ">"


def _signature(fn):
    import inspect
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        if name == "self":
            continue
        out.append([name, p.kind.name, _default_repr(p.default)])
    return out


def api_surface():
    """class -> {method -> [[parameter, kind, default], ...]} for every public method the reference's classes define themselves
    (torch.nn.Module / Lightning machinery excluded), plus the module-level factory functions.  Data only: names and defaults."""
    import inspect
    import json
    import importlib
    surface = {"classes": {}, "functions": {}, "exports": {}}
    wanted = set()
    for ns, names in SURFACE_CLASSES.items():
        mod = importlib.import_module("diffsci.models." + ns)
        for cname in names:
            cls = getattr(mod, cname)
            wanted.add(cname)
            methods = {}
            for klass in cls.__mro__:
                if not klass.__module__.startswith("diffsci."):
                    continue
                for mname, obj in vars(klass).items():
                    if mname.startswith("_") and mname != "__init__":
                        continue
                    if mname in methods:
                        continue                       # the most derived definition wins
                    kind = "method"
                    if isinstance(obj, staticmethod):
                        obj, kind = obj.__func__, "staticmethod"
                    elif isinstance(obj, classmethod):
                        obj, kind = obj.__func__, "classmethod"
                    elif isinstance(obj, property):
                        methods[mname] = {"kind": "property"}
                        continue
                    if not inspect.isfunction(obj):
                        continue
                    sig = _signature(obj)
                    if kind == "classmethod" and sig and sig[0][0] == "cls":
                        sig = sig[1:]
                    methods[mname] = {"kind": kind, "params": sig}
            surface["classes"][ns + "." + cname] = {
                "bases": [b.__name__ for b in cls.__mro__[1:] if b.__module__.startswith("diffsci.")], "methods": methods}
    for ns, names in SURFACE_FUNCTIONS.items():
        mod = importlib.import_module("diffsci.models." + ns)
        for fname in names:
            wanted.add(fname)
            surface["functions"][ns + "." + fname] = _signature(getattr(mod, fname))
    # which of those names the package namespaces re-export
    for ns in ("", "karras", "nets"):
        mod = importlib.import_module("diffsci.models" + ("." + ns if ns else ""))
        surface["exports"][ns] = sorted(n for n in wanted if hasattr(mod, n))
    path = os.path.join(OUT, "api_surface.json")
    with open(path, "w") as f:
        json.dump(surface, f, indent=1, sort_keys=True)
    print(f"api_surface: {os.path.getsize(path)/1024:.1f} KiB, {len(surface['classes'])} classes")


# ---------------------------------------------------------------- a Lightning-format checkpoint (karrasmodule.py:410-429)
def checkpoint():
    """What a reference user holds: a .ckpt written by Lightning's Trainer -- a dict whose "state_dict" carries the module's
    keys ("model.*", "edm_batch_norm.*").  Written here from a reference KarrasModule in that format (Lightning itself is absent
    from the image: the dict layout is its documented one), next to what the reference samples from those weights."""
    torch.manual_seed(120)
    cfg = M.nets.PUNetGConfig(model_channels=8)
    net = M.nets.PUNetG(cfg).eval()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "gnorm" in k:
                v.add_(0.25 * torch.randn_like(v))
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(has_edm_batch_norm=True)).eval()
    module.edm_batch_norm.running_mean = torch.tensor([0.15])
    module.edm_batch_norm.running_var = torch.tensor([1.7])
    ckpt = {"epoch": 3, "global_step": 1234, "pytorch-lightning_version": "2.2.0",
            "state_dict": {k: v.clone() for k, v in module.state_dict().items()},
            "loops": {}, "callbacks": {}, "optimizer_states": [], "lr_schedulers": [], "hyper_parameters": {}}
    path = os.path.join(OUT, "ckpt8_lightning.ckpt")
    torch.save(ckpt, path)
    print(f"ckpt8_lightning.ckpt: {os.path.getsize(path)/1024:.1f} KiB, {len(ckpt['state_dict'])} tensors")
    torch.manual_seed(121)
    wn = torch.randn(2, 1, 32, 32)
    npz("ckpt8", white_noise=wn, sample_N4=module.propagate_white_noise(wn, nsteps=4),
        keys=np.array(sorted(ckpt["state_dict"])))


if __name__ == "__main__":
    which = sys.argv[1:] or ["schedule", "toy", "punetg", "porosity", "inpaint", "vpve", "circular", "si", "punetgcond", "langevin", "adm", "variants", "latent", "adm_norms", "autoregressive", "si_latent", "si_inpaint", "volumes", "si_generic", "vp_karras", "adm_blocks", "spatial_cond", "api_surface", "checkpoint"]
    for name in which:
        globals()[name]()
