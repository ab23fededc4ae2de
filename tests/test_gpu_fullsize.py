"""BASELINE configs 3 and 5 at their FULL sizes on a real MI355X (they had only been
checked for finiteness).  The CPU oracle cannot run whole workloads of this size in test time, so -- as
test_full_size_config2_properties does for config 2 -- the full batch is checked through properties that hold
for any correct sampler (samples are independent: permutation / split invariance; replay determinism) and the
oracle runs on one or two samples of the very same full-size tensors: a single evaluation and a short trajectory,
in fp32 and (one sample) fp64.

Tolerance (stated, fp32): rel-L2 <= 1e-5 against the oracle's fp32 result, or -- for trajectories of a few giant
steps through a random-init network, which are ill-conditioned in the reference's own arithmetic -- 4x the oracle's
own fp32-vs-fp64 distance on the same input."""
import math
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import adm_ref, embedder_ref, punetg_ref  # noqa: E402
from oracle import karras_ref as K  # noqa: E402
from tests.golden_util import rel_l2  # noqa: E402

REL = 1e-5


def _bounded(name, got, want32, want64, rel_tol=None, k32=4.0, k64=1.0):
    """The stated tolerance max(REL, k x the oracle's own fp32-vs-fp64 distance), with the clause that binds printed and the
    fp64 clause asserted on its own line: against the fp32 oracle the bound is usually the SECOND clause at these sizes (torch's
    CPU fp32 is 2-3e-4 from its fp64), so the assertion that carries information is the fp64 one."""
    rel_tol = REL if rel_tol is None else rel_tol
    ref_err = rel_l2(want32, want64)
    e32, e64 = rel_l2(got, want32), rel_l2(got, want64)
    b32, b64 = max(rel_tol, k32 * ref_err), max(rel_tol, k64 * ref_err)
    print(f"[{name}] oracle fp32 vs fp64 {ref_err:.2e} | HIP vs fp32 oracle {e32:.2e} <= {b32:.2e} "
          f"({'REL' if rel_tol >= k32 * ref_err else f'{k32:g} x oracle error'} binds) | HIP vs fp64 oracle {e64:.2e} <= {b64:.2e} "
          f"({'REL' if rel_tol >= k64 * ref_err else f'{k64:g} x oracle error'} binds)")
    assert e64 < b64, f"{name}: HIP vs the fp64 oracle {e64:.3e} exceeds max(REL = {rel_tol:g}, {k64:g} x {ref_err:.3e}) -- the informative clause"
    assert e32 < b32, f"{name}: HIP vs the fp32 oracle {e32:.3e} exceeds max(REL = {rel_tol:g}, {k32:g} x {ref_err:.3e})"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def M():
    import diffsci_amd.models as M
    return M


def _f64(sd):
    return {k: (w.double() if w.is_floating_point() else w) for k, w in sd.items()}


def test_attention_at_config5_length(dev):
    """L = 4096 tokens (the 64 x 64 bottleneck of a 256 x 256 field), E = 256: both attention kernels against the
    fp64 definition.  r1 tested up to L = 1024 only."""
    from diffsci_amd import ops
    B, E, L = 1, 256, 4096
    g = torch.Generator().manual_seed(E + L)
    qkv = torch.randn(B, 3 * E, L, generator=g)
    q, k, v = (t.transpose(1, 2).double() for t in qkv.split(E, dim=1))
    want = (torch.softmax((q * math.sqrt(1.0 / E)) @ k.transpose(1, 2), dim=-1) @ v).transpose(1, 2)
    for precision in ("fp16x3", "fp32"):
        got = ops.attention(qkv.to(dev), E, precision=precision).cpu()
        assert rel_l2(got, want) < 2e-6, precision
        assert (got.double() - want).abs().max().item() < 2e-5, precision
    # peaked logits (a few keys dominate each query): the online-softmax rescale path at this length
    qkv[:, :E] *= 6.0
    q = qkv[:, :E].transpose(1, 2).double()
    want = (torch.softmax((q * math.sqrt(1.0 / E)) @ k.transpose(1, 2), dim=-1) @ v).transpose(1, 2)
    got = ops.attention(qkv.to(dev), E, precision="fp16x3").cpu()
    assert rel_l2(got, want) < 2e-6


def test_full_size_config3_adm128_sigma_churn(M, dev):
    """BASELINE config 3: ADM 128 base channels, channel_expansion [1, 2, 4, 4] (attention at 16 x 16), 115.8 M
    parameters, x = [32, 3, 256, 256], sigma-churn sampler (reference: nets/adm.py:199-216, integrators.py:72-113)."""
    cfg = dict(input_channels=3, output_channels=3, model_channels=128, time_embed_dim=128, output_embed_dim=512,
               channel_expansion=[1, 2, 4, 4])
    torch.manual_seed(0)
    net = M.ADM(M.ADMConfig(**cfg))
    with torch.no_grad():                                      # non-trivial norm affines and biases
        for k, w in net.state_dict().items():
            if "norm" in k or k.endswith("bias"):
                w.add_(0.1 * torch.randn_like(w))
    assert abs(sum(p.numel() for p in net.parameters()) / 1e6 - 115.8) < 0.2
    sd = {k: w.detach().clone() for k, w in net.state_dict().items()}
    ocfg = adm_ref.default_config(**cfg)
    onet = adm_ref.make_net(sd, ocfg)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    B = 32
    wn = torch.randn(B, 3, 256, 256, generator=g)
    eps = torch.randn(3, B, 3, 256, 256, generator=g)
    x = wn.to(dev)
    # ---- one evaluation of the whole batch; oracle on two samples (fp32) and one (fp64)
    rows = [3, 20]
    sig = 2.5
    got_d = module.get_denoiser((x * sig).contiguous(), torch.full((B,), sig, device=dev))[0][rows].cpu()
    t0 = time.time()
    with torch.inference_mode():
        want_d = K.denoiser(onet, wn[rows] * sig, torch.full((2,), sig))
        want_d64 = K.denoiser(adm_ref.make_net(_f64(sd), ocfg), (wn[rows[:1]] * sig).double(), torch.full((1,), sig).double())
    print(f"[config 3] oracle: 2 fp32 + 1 fp64 evaluations in {time.time() - t0:.1f} s")
    assert rel_l2(got_d, want_d) < REL
    ref_err = rel_l2(want_d[:1], want_d64)
    assert rel_l2(got_d[:1], want_d64) < max(4 * ref_err, 2e-6)
    # ---- 3-step sigma-churn trajectory with injected noise: full batch on the GPU, sample 7 on the oracle
    grid = module.config.noisescheduler.create_steps(4)
    h = module.propagate_white_noise(x, nsteps=3, integrator="karras", eps=eps.to(dev), record_history=True)
    assert h.shape == (4, B, 3, 256, 256) and torch.isfinite(h).all()
    r = 7
    t0 = time.time()
    with torch.inference_mode():
        want_t = K.propagate_white_noise(onet, wn[r:r + 1], 3, integrator="karras", eps=eps[:, r:r + 1], sigma_grid=grid,
                                         record_history=True)
        want_t64 = K.propagate_white_noise(adm_ref.make_net(_f64(sd), ocfg), wn[r:r + 1].double(), 3, integrator="karras",
                                           eps=eps[:, r:r + 1].double(), sigma_grid=grid.double(), record_history=True)
    print(f"[config 3] oracle: 3-step churn trajectory of one sample, fp32 + fp64, in {time.time() - t0:.1f} s")
    got_t = h[:, r:r + 1].cpu()
    assert torch.equal(got_t[0], want_t[0])                                        # x * sigma_max is exact
    assert rel_l2(got_t[1], want_t[1]) < REL                                       # first step: well conditioned
    _bounded("config 3, sigma-churn trajectory", got_t, want_t, want_t64, k64=4.0)
    # ---- samples are independent: permutation / split; replay of the captured plan is bit-reproducible
    out = h[-1]
    perm = torch.randperm(B, generator=g)
    outp = module.propagate_white_noise(x[perm.to(dev)].contiguous(), nsteps=3, integrator="karras",
                                        eps=eps[:, perm].contiguous().to(dev))
    assert torch.equal(outp, out[perm.to(dev)])
    half = module.propagate_white_noise(x[16:].contiguous(), nsteps=3, integrator="karras", eps=eps[:, 16:].contiguous().to(dev))
    assert rel_l2(half.cpu(), out[16:].cpu()) < 1e-6                               # other tile -> XCD assignment only
    again = module.propagate_white_noise(x, nsteps=3, integrator="karras", eps=eps.to(dev))
    assert torch.equal(again, out)
    # ---- the whole workload: 50 steps, noise generated in the kernels; reproducible from the generator seed
    torch.manual_seed(2)
    full = module.propagate_white_noise(x, nsteps=50, integrator="karras")
    assert full.shape == (B, 3, 256, 256) and torch.isfinite(full).all() and float(full.std()) > 1e-3
    torch.manual_seed(2)
    assert torch.equal(module.propagate_white_noise(x, nsteps=50, integrator="karras"), full)
    assert net.conv_precision == "fp16x3"                                          # the range guard never fired


def test_full_size_config5_conditional_punetg_cfg(M, dev):
    """BASELINE config 5, one GPU's share: conditional PUNetG-64 with the dict-style PorosityEmbedder, x = [16, 4, 256,
    256], classifier-free guidance g = 2 (two network evaluations per score; attention over L = 4096 tokens), Heun
    (reference: nets/punetg.py:389-416, nets/embedder.py:198-229, karrasmodule.py:702-716)."""
    mcfg = dict(input_channels=4, output_channels=4)
    torch.manual_seed(0)
    net = M.PUNetG(M.PUNetGConfig(**mcfg), conditional_embedding=M.nets.PorosityEmbedder(dembed=64))
    with torch.no_grad():
        for k, w in net.state_dict().items():
            if "gnorm" in k or k.endswith("bias"):
                w.add_(0.1 * torch.randn_like(w))
    sd = {k: w.detach().clone() for k, w in net.state_dict().items()}
    ocfg = punetg_ref.default_config(**mcfg)

    def oracle_net(sdx):
        return punetg_ref.make_net(sdx, ocfg, embed=lambda y: embedder_ref.porosity_embed(sdx, "conditional_embedding.", y))
    onet = oracle_net(sd)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    B = 16
    wn = torch.randn(B, 4, 256, 256, generator=g)
    x = wn.to(dev)
    y = {"porosity": torch.tensor([0.2])}                         # un-batched, as sample() takes it (karrasmodule.py:916-917)
    yd = {"porosity": y["porosity"].to(dev)}
    yb = {"porosity": y["porosity"].unsqueeze(0)}
    # ---- one guided evaluation of the whole batch; oracle on sample 9 (fp32 and fp64)
    sig, r = 1.7, 9
    got_d = module.get_denoiser((x * sig).contiguous(), torch.full((B,), sig, device=dev),
                                y={"porosity": yd["porosity"].unsqueeze(0)}, guidance=2.0)[0][r:r + 1].cpu()
    t0 = time.time()
    with torch.inference_mode():
        want_d = K.denoiser(onet, wn[r:r + 1] * sig, torch.full((1,), sig), y=yb, guidance=2.0, conditional=True)
        want_d64 = K.denoiser(oracle_net(_f64(sd)), (wn[r:r + 1] * sig).double(), torch.full((1,), sig).double(),
                              y={"porosity": yb["porosity"].double()}, guidance=2.0, conditional=True)
    print(f"[config 5] oracle: one guided evaluation, fp32 + fp64, in {time.time() - t0:.1f} s")
    # At this size torch's own fp32 CPU arithmetic is 1.9e-4 away from its fp64 result (tools/diag_cfg5.py: the HIP
    # path is 9e-6 from fp64, every stage 1e-7..2e-6 from the oracle), so the fp32 oracle is compared at the second
    # clause of the stated tolerance and the fp64 oracle at the first whenever the reference's own error allows
    _bounded("config 5, one guided evaluation", got_d, want_d, want_d64)     # fp64 clause: no worse than the reference arithmetic itself
    # ---- 2-step Heun with guidance: full batch on the GPU, sample 9 on the oracle
    grid = module.config.noisescheduler.create_steps(3)
    h = module.propagate_white_noise(x, y=yd, guidance=2.0, nsteps=2, record_history=True)
    assert h.shape == (3, B, 4, 256, 256) and torch.isfinite(h).all()
    t0 = time.time()
    with torch.inference_mode():
        want_t = K.propagate_white_noise(onet, wn[r:r + 1], 2, y=y, guidance=2.0, conditional=True, sigma_grid=grid,
                                         record_history=True)
        want_t64 = K.propagate_white_noise(oracle_net(_f64(sd)), wn[r:r + 1].double(), 2, y={"porosity": y["porosity"].double()},
                                           guidance=2.0, conditional=True, sigma_grid=grid.double(), record_history=True)
    print(f"[config 5] oracle: 2-step guided Heun of one sample, fp32 + fp64, in {time.time() - t0:.1f} s")
    got_t = h[:, r:r + 1].cpu()
    _bounded("config 5, first guided Heun step", got_t[1], want_t[1], want_t64[1], k64=4.0)
    _bounded("config 5, 2-step guided trajectory", got_t, want_t, want_t64, k64=4.0)
    # ---- independence / replay on the full batch
    out = h[-1]
    perm = torch.randperm(B, generator=g)
    outp = module.propagate_white_noise(x[perm.to(dev)].contiguous(), y=yd, guidance=2.0, nsteps=2)
    assert torch.equal(outp, out[perm.to(dev)])
    half = module.propagate_white_noise(x[8:].contiguous(), y=yd, guidance=2.0, nsteps=2)
    assert rel_l2(half.cpu(), out[8:].cpu()) < 1e-6
    assert torch.equal(module.propagate_white_noise(x, y=yd, guidance=2.0, nsteps=2), out)
    other = module.propagate_white_noise(x, y={"porosity": torch.tensor([0.7], device=dev)}, guidance=2.0, nsteps=2)
    assert rel_l2(other, out) > 1e-4                               # the replay follows the condition's value
    # ---- a longer run of the full configuration: 10 of the 100 steps (38 network calls), finite and reproducible
    full = module.propagate_white_noise(x, y=yd, guidance=2.0, nsteps=10)
    assert torch.isfinite(full).all() and float(full.std()) > 1e-3
    assert torch.equal(module.propagate_white_noise(x, y=yd, guidance=2.0, nsteps=10), full)
