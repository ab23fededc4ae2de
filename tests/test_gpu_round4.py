"""Round 4 on a real MI355X: the persistent producer / consumer convolution against the one-tile kernel, and the parts of the
reference's Python surface added this round (Lightning checkpoints, MLPCond, the public ADM containers, the analytic-score
interpolation of propagate_partial_toward_sample) with GPU compute."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden_util import load, rel_l2  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def M():
    import diffsci_amd.models as M
    return M


def test_persistent_convolution_is_bit_identical_to_the_one_tile_kernel():
    """ds_conv3p.hip (producer / consumer workgroups walking many tiles) accumulates in the order of ds_conv3h.hip: outputs, tile
    statistics and output maxima of ten launch shapes -- fused loader, raw input, residuals, periodic padding, tap offsets,
    one and two channel tiles, uneven item counts, config 2's level-0 / level-1 sizes -- agree bit for bit, and with fp64 to 2e-6.
    (The library reads DS_CONV_PC once per process: tools/conv3p_check.py runs the reference arm in a child process.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("DS_CONV_PC", "DS_CONV_PC_MIN")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv3p_check.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and "ALL OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    assert p.stdout.count(" ok") >= 10


def test_persistent_image_input_convolution_is_bit_identical_to_the_one_shot_kernel():
    """ds_conv3p.hip with pre-split image input (the 256-channel level of config 2: producers issue DMA only) against ds_conv3h.hip's
    image-input kernel (DS_CONV_PC_IMG=0, child process): config 2's level-2 shape with and without residual, two residuals, uneven
    item counts, a launch below the persistent kernel's minimum -- outputs, tile statistics, output maxima bit for bit; fp64 to 2e-6."""
    env = {k: v for k, v in os.environ.items() if k not in ("DS_CONV_PC", "DS_CONV_PC_MIN", "DS_CONV_PC_IMG")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv3p_img_check.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and "ALL OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    assert p.stdout.count(" ok") >= 6


def test_sixteen_byte_patch_loads_are_bit_identical_to_the_one_pixel_staging_plan():
    """ds_conv3h.hip's VEC staging plan (units of 2 channels x 4 pixels fetched by 16-byte loads, table rows through LDS) against its
    one-pixel items (DS_CONV_VEC=0, child process), DS_CONV_PC=0 in both arms: 18 launch shapes -- one and two channel tiles, ragged
    heights, row tap offsets, periodic padding on a single tile column, raw inputs, 32 ... 160 input channels -- bit for bit."""
    env = {k: v for k, v in os.environ.items() if k not in ("DS_CONV_PC", "DS_CONV_PC_MIN", "DS_CONV_VEC")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv3p_check.py"), "--var", "DS_CONV_VEC"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "ALL OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    assert p.stdout.count(" ok") >= 18


def test_round_four_kernels_change_no_output_bit_of_the_headline_workload(tmp_path):
    """PUNetG-64 on [64,1,128,128], 4-step Heun through the captured plan and one eager evaluation: with every switch of the round off
    (one-shot kernels, one-pixel staging: the round-3 code paths) and with the defaults (persistent kernels on all three levels, 16-byte
    patch loads) the results are bit-identical, and replays are reproducible in both (tools/pc_determinism.py)."""
    ref = str(tmp_path / "ref.pt")
    tool = os.path.join(ROOT, "tools", "pc_determinism.py")
    base = {k: v for k, v in os.environ.items() if not k.startswith("DS_CONV_")}
    p = subprocess.run([sys.executable, tool, "--save", ref], cwd=ROOT, env=dict(base, DS_CONV_PC="0", DS_CONV_VEC="0", DS_CONV_TWO_EARLY="0"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    p = subprocess.run([sys.executable, tool, "--compare", ref], cwd=ROOT, env=base, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "identical to the saved run: True" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_output_layer_convolution_with_sixteen_byte_loads_is_bit_identical():
    """ds_conv2d_direct on whole 64-column tiles (16-byte patch loads, the next chunk's loads in front of this chunk's arithmetic)
    against the general kernel (DS_DIRECT_VEC=0, child process): the output layers of configs 2, 3 and 5, small periodic and
    ragged-height cases -- bit for bit, and against fp64 (tools/direct_vec_check.py)."""
    env = {k: v for k, v in os.environ.items() if k != "DS_DIRECT_VEC"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "direct_vec_check.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0 and "ALL OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_exact_input_layer_on_a_periodic_network_raises(M, dev):
    """the exact-fp32 input layer zero-pads; a periodic network that has it switched on by hand must raise, not compute
    with the wrong padding (precision.escalate_input itself refuses periodic networks)."""
    net = M.PUNetG(M.PUNetGConfig(model_channels=8, convolution_type="circular")).to(dev).eval()
    net.exact_input_layer = True
    x, t = torch.randn(2, 1, 16, 16, device=dev), torch.rand(2, device=dev)
    with pytest.raises(NotImplementedError, match="periodic padding"):
        net(x, t)


def test_lightning_checkpoint_samples_like_the_reference(M, dev):
    """karrasmodule.py:410-429: a .ckpt in Lightning's layout, written from a reference module by make_golden.py, loaded through
    the mirrored classmethod; four Heun steps from the recorded noise against what the reference sampled from those weights."""
    v, _ = load("ckpt8")
    module = M.KarrasModule.load_from_checkpoint(os.path.join(ROOT, "tests", "golden", "ckpt8_lightning.ckpt"),
                                                 model=M.PUNetG(M.PUNetGConfig(model_channels=8)),
                                                 config=M.KarrasModuleConfig.from_edm(has_edm_batch_norm=True)).to(dev).eval()
    got = module.propagate_white_noise(v["white_noise"].to(dev), nsteps=4).cpu()
    assert rel_l2(got, v["sample_N4"]) < REL


def test_mlp_cond_against_torch(M, dev):
    """mlp.py:61-121 on ds_linear: the same stack on [x, t, y]."""
    torch.manual_seed(3)
    net = M.MLPCond(3, 2, [16, 8])
    ref = torch.nn.Sequential(*[torch.nn.ReLU() if isinstance(m, torch.nn.Identity) else m for m in net.net])
    x, t, y = torch.randn(37, 3), torch.rand(37) * 3, torch.randn(37, 2)
    want = ref.double()(torch.cat([x, t[:, None], y], dim=-1).double())
    net.float()
    ref.float()
    got = net.to(dev)(x.to(dev), t.to(dev), y.to(dev)).cpu()
    assert rel_l2(got, want) < 1e-6
    one = net(x.to(dev), t.to(dev), y[:1].to(dev)).cpu()                             # one condition row for the whole batch
    want1 = ref.cpu()(torch.cat([x, t[:, None], y[:1].expand(37, -1)], dim=-1))
    assert rel_l2(one, want1) < 1e-6


def test_public_adm_containers_compose_to_the_network(M, dev):
    """adm.py:120-216: ADM = input_layer -> ADMEncoder -> ADMMiddleBlock -> ADMDecoder -> output_layer around ADMTimeEmbedding.  The
    public containers (same constructor arguments and state_dict keys as the reference's) run block by block on standalone
    kernels; the whole-network class folds norms and is pinned by the reference goldens -- the two must agree, for both decoder
    types and both skip integrations."""
    from diffsci_amd import ops
    for decoder_type, skip in ((1, "concat"), (2, "concat"), (1, "add")):
        torch.manual_seed(10 + decoder_type)
        cfg = M.ADMConfig(input_channels=2, output_channels=3, model_channels=16, time_embed_dim=16, output_embed_dim=32,
                          channel_expansion=[1, 2], number_resnet_attn_block=2, skip_integration_type=skip, decoder_type=decoder_type)
        net = M.ADM(cfg)
        with torch.no_grad():
            for k, w in net.state_dict().items():
                if "norm" in k or k.endswith("bias"):
                    w.add_(0.2 * torch.randn_like(w))
        net = net.to(dev).eval()
        sd = net.state_dict()
        kw = dict(first_norm=cfg.first_resblock_norm, second_norm=cfg.second_resblock_norm)
        enc = M.nets.ADMEncoder(cfg.model_channels, cfg.output_embed_dim, cfg.extended_channel_expansion,
                                cfg.number_resnet_downward_block, cfg.convolution_type, has_residual=True, has_attn=False,
                                attn_residual=cfg.attn_residual, **kw)
        mid = M.nets.ADMMiddleBlock(cfg.middle_channel, cfg.output_embed_dim, cfg.num_blocks_middle_block, conv_type=cfg.convolution_type,
                                    has_residual=True, has_attn=cfg.middle_block_attn_config, attn_residual=cfg.attn_residual, **kw)
        dec = M.nets.ADMDecoder(cfg.model_channels, cfg.output_embed_dim, cfg.extended_channel_expansion[::-1],
                                cfg.number_resnet_upward_block, cfg.convolution_type, has_residual=True, has_attn=False,
                                attn_residual=cfg.attn_residual, skip_integration_type=skip, decoder_type=decoder_type, **kw)
        temb = M.nets.ADMTimeEmbedding(cfg.time_embed_dim, cfg.output_embed_dim, cfg.time_projection_scale)
        for part, prefix in ((enc, "encoder."), (mid, "middle_block."), (dec, "decoder."), (temb, "time_embedding.")):
            part.load_state_dict({k[len(prefix):]: w for k, w in sd.items() if k.startswith(prefix)})      # strict: the keys are the reference's
            part.to(dev).eval()
        x = torch.randn(3, 2, 32, 32, device=dev)
        t = torch.tensor([0.3, 1.0, 2.5], device=dev)
        want = net(x, t)
        te = temb(t)
        h = ops.conv(x, ops.pack_conv(sd["input_layer.weight"], "fp16x3"), bias=sd["input_layer.bias"])
        h, inter = enc(h, te)
        assert len(inter) == enc.nlayers + 1 and enc.channels_outs == [16, 32]
        h = mid(h, te)
        h = dec(h, te, inter)
        assert len(inter) == 1                                                         # pop=True consumed one skip per layer; the stem's copy stays, as in the reference
        got = ops.conv(h, ops.pack_conv(sd["output_layer.weight"], "fp16x3"), bias=sd["output_layer.bias"])
        assert rel_l2(got, want) < REL, (decoder_type, skip)


def test_partial_propagation_interpolates_with_an_analytic_score(M, dev):
    """karrasmodule.py:951-963: score = alpha trained + (1 - alpha) analytic with alpha = interp_fn(sigma); the reference's
    positional order (integrator before analytical_score / interp_fn), alpha = 1 reproduces the plain partial run, alpha = 0 the
    run on the analytic score alone, a per-sigma blend lies on the scheduler's own path for the blended score."""
    torch.manual_seed(5)
    net = M.MLPUncond(2, [20])
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
    x = (torch.randn(64, 2) * 80.0).to(dev)

    def analytic(xc, sc):                                   # N(0, 0.7^2 I): score -x / (sigma^2 + s^2), evaluated on the host as in the reference
        assert xc.device.type == "cpu" and sc.device.type == "cpu"
        return -xc / (sc[:, None] ** 2 + 0.49)
    plain = module.propagate_partial_toward_sample(x, 2, 9, None, 12, False, "heun")
    one = module.propagate_partial_toward_sample(x, 2, 9, None, 12, False, "heun", analytic, lambda s: torch.ones_like(s))
    assert rel_l2(one, plain) < 1e-6
    zero = module.propagate_partial_toward_sample(x, 2, 9, None, 12, False, "heun", analytic, lambda s: torch.zeros_like(s))
    sch = module.config.noisescheduler
    sch.set_temporary_integrator("heun")
    want0 = sch.propagate_partial(x, lambda xx, ss: (-xx / (ss[:, None] ** 2 + 0.49)).contiguous(), 12, 2, 9)
    sch.unset_temporary_integrator()
    assert rel_l2(zero, want0) < 1e-6
    half = module.propagate_partial_toward_sample(x, 2, 9, nsteps=12, integrator="heun", analytical_score=analytic,
                                                  interp_fn=lambda s: 0.25 + 0 * s)
    sch.set_temporary_integrator("heun")
    want_h = sch.propagate_partial(
        x, lambda xx, ss: (0.25 * module.get_score(xx, ss) + 0.75 * (-xx / (ss[:, None] ** 2 + 0.49))).contiguous(), 12, 2, 9)
    sch.unset_temporary_integrator()
    assert rel_l2(half, want_h) < 1e-6
    with pytest.raises(AssertionError):
        module.propagate_partial_toward_sample(x, 2, 9, nsteps=12, interp_fn=lambda s: s)


def test_condition_lists_and_scalars_under_capture_eager(M, dev):
    """a user network evaluated as given and captured with torch.cuda.CUDAGraph reads a plan-owned copy of the
    condition.  SIModule.integrate_flow_field hands y to the network untouched, so it may hold tensors inside lists / tuples and
    Python scalars: the tensors must follow the caller's values from run to run (cloned and refreshed like dict members), and
    a scalar, which a captured graph cannot re-read, must select a different plan instead of replaying the first value."""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(0.3))

        def forward(self, x, t, y=None):
            fields, gain = y["fields"], y["gain"]
            return self.w * x + fields[0] * gain + fields[1][0]
    mod = M.SIModule(M.SIModuleConfig(scheduler="linear"), Net()).to(dev).eval()
    x = torch.randn(4, 1, 8, 8, generator=torch.Generator().manual_seed(2)).to(dev)
    ts = torch.linspace(1, 0, 4)

    def cond(a, b, gain):
        return {"fields": [torch.full((1, 1, 8, 8), a, device=dev), (torch.full((1, 1, 8, 8), b, device=dev),)], "gain": gain}
    keys = ((1.0, 2.0, 0.5), (3.0, -1.0, 0.5), (1.0, 2.0, 2.0), (1.0, 2.0, 0.5))
    want = {k: mod.integrate_flow_field(x, ts, y=cond(*k)) for k in set(keys)}        # step by step: no capture, no plan-owned copy
    assert len(mod._plans.plans) == 0
    mod.capture_eager = True
    got = [(k, mod.integrate_flow_field(x, ts, y=cond(*k))) for k in keys]
    assert len(mod._plans.plans) == 2                                                 # gain 0.5 and gain 2.0; the tensors' values share a plan
    for k, o in got:
        assert torch.equal(o, want[k]), k
    assert not torch.equal(want[(1.0, 2.0, 0.5)], want[(3.0, -1.0, 0.5)]) and not torch.equal(want[(1.0, 2.0, 0.5)], want[(1.0, 2.0, 2.0)])
