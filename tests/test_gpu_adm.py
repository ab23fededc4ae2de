"""ADM score network on a real MI355X against goldens generated from
the reference (tests/golden/adm8_*.npz) and against the CPU oracle on fresh inputs.

Tolerance (fp32, stated): the group-1 statistics are accumulated in fp64 (the reference: fp32
cascade sums), convolutions accumulate in MFMA order; fields agree to rel-L2 <= 1e-5 and the
error against the reference's own fp64 run stays within 4x the reference's fp32-vs-fp64 error."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import adm_ref  # noqa: E402
from oracle import karras_ref as K  # noqa: E402
from tests.golden_util import load, rel_l2  # noqa: E402

REL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def M():
    import diffsci_amd.models as M
    return M


def _net(M, dev, skip):
    v, sd = load(f"adm8_{skip}")
    net = M.ADM(M.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16, skip_integration_type=skip))
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.to(dev), v, sd


def test_state_dict_keys_match_reference(M):
    _, sd = load("adm8_concat")
    net = M.ADM(M.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16))
    assert set(net.state_dict().keys()) == set(sd.keys())
    for k, w in net.state_dict().items():
        assert tuple(w.shape) == tuple(sd[k].shape), k


def test_group1_norm_kernels(dev):
    from diffsci_amd import ops
    import torch.nn.functional as F
    torch.manual_seed(0)
    for (B, C, H, W) in [(2, 8, 32, 32), (3, 24, 16, 8), (1, 5, 4, 4), (2, 64, 64, 64)]:
        x = torch.randn(B, C, H, W) * 3 + 0.7
        w, b = torch.randn(C), torch.randn(C)
        film = torch.randn(B, 2 * C)
        xd = x.to(dev)
        st = ops.gnorm1_stats(xd, 0)
        want = F.silu(F.group_norm(x.double(), 1, w.double(), b.double(), 1e-5))
        got = ops.gnorm1_apply(xd, st, w.to(dev), b.to(dev), 0).cpu()
        assert rel_l2(got, want) < 5e-7
        got = ops.gnorm1_apply(xd, st, w.to(dev), b.to(dev), 0, pool=True).cpu()
        assert rel_l2(got, F.avg_pool2d(want, 2)) < 5e-7
        st = ops.gnorm1_stats(xd, 1)
        y = adm_ref.group1_rms_norm(x.double(), w.double(), b.double())
        want = F.silu(y * film[:, :C, None, None].double() + film[:, C:, None, None].double())
        got = ops.gnorm1_apply(xd, st, w.to(dev), b.to(dev), 1, film=film.to(dev)).cpu()
        assert rel_l2(got, want) < 5e-7
        got = ops.gnorm1_apply(xd, st, w.to(dev), b.to(dev), 1, film=film[:1].contiguous().to(dev)).cpu()
        want = F.silu(y * film[:1, :C, None, None].double() + film[:1, C:, None, None].double())
        assert rel_l2(got, want) < 5e-7
        # residual-branch pooling is bit-exact (same summation order as torch)
        got = ops.gnorm1_apply(xd, None, None, None, 2, pool=True).cpu()
        assert torch.equal(got, F.avg_pool2d(x, 2))
    a, b2 = torch.randn(2, 3, 4, 4), torch.randn(2, 5, 4, 4)
    assert torch.equal(ops.concat2(a.to(dev), b2.to(dev)).cpu(), torch.cat([a, b2], 1))
    h, ye = torch.randn(4, 16), torch.randn(4, 16)
    torch.testing.assert_close(ops.add_act(h.to(dev), ye.to(dev), act=1).cpu(), F.silu(h + ye), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(ops.add_act(h.to(dev), ye[:1].to(dev), act=0).cpu(), h + ye[:1], rtol=0, atol=0)


@pytest.mark.parametrize("skip", ["concat", "add"])
def test_adm_layers_and_forward_vs_reference(M, dev, skip):
    net, v, sd = _net(M, dev, skip)
    te = net.embed_time(v["t"].to(dev))
    assert (te.cpu() - v["te"]).abs().max() < 2e-6
    pk = net.packed_weights()
    stem = net._conv(net.input_layer, v["x"].to(dev), pk)
    assert rel_l2(stem.cpu(), v["stem"]) < 2e-6
    films = net.time_shifts(v["te"].to(dev))
    b0, _ = net._block(net.encoder.layers[0].input_blocks[0], v["stem"].to(dev), films[0], pk, net._ws)
    assert rel_l2(b0.cpu(), v["enc00"]) < 5e-6
    b1, _ = net._block(net.encoder.layers[0].input_blocks[1], v["enc00"].to(dev), films[1], pk, net._ws)
    assert rel_l2(b1.cpu(), v["enc01_down"]) < 5e-6
    out = net(v["x"].to(dev), v["t"].to(dev)).cpu()
    assert rel_l2(out, v["out_f32"]) < REL
    ref_err = rel_l2(v["out_f32"], v["out_f64"])
    assert rel_l2(out, v["out_f64"]) < max(4 * ref_err, 2e-6)


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("skip", ["concat", "add"])
def test_adm_trajectories_vs_reference(M, dev, skip, use_graph):
    net, v, _ = _net(M, dev, skip)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm())
    module.use_graph = use_graph
    wn = v["white_noise"].to(dev)
    o = module.propagate_white_noise(wn, nsteps=6).cpu()
    assert rel_l2(o, v["out_heun_N6_f32"]) < REL
    h = module.propagate_white_noise(wn, nsteps=4, record_history=True, integrator="karras",
                                     eps=v["eps_karras_N4"].to(dev)).cpu()
    assert h.shape == v["hist_karras_N4_f32"].shape
    assert rel_l2(h, v["hist_karras_N4_f32"]) < REL


ADM_NORMS = {"rms_ln": dict(first_resblock_norm="GroupRMS", second_resblock_norm="GroupLN"),
             "ln_ln_noaffine": dict(first_resblock_norm="GroupLN", second_resblock_norm="GroupLN", affine_norm=False),
             "dec2": dict(decoder_type=2)}                    # ADMDecoderLayer2: every decoder block joins the skip


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("tag", sorted(ADM_NORMS))
def test_adm_norm_choices(M, dev, tag, fuse):
    """make_norm_layers (adm.py:385-406): GroupNorm(1, C) or GroupRMSNorm(1, C) in either slot (FiLM follows the
    second); affine_norm=False is ignored by the reference's ADM, so the norms stay affine -- folded into the convolutions and as standalone kernels."""
    v, sd = load("adm8_" + tag)
    net = M.ADM(M.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16, **ADM_NORMS[tag]))
    r = net.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    net = net.to(dev)
    net.fuse_norm = fuse
    pk = net.packed_weights()
    films = net.time_shifts(net.embed_time(v["t"].to(dev)))
    b0, _ = net._block(net.encoder.layers[0].input_blocks[0], v["stem"].to(dev), films[0], pk, net._ws)
    assert rel_l2(b0.cpu(), v["enc00"]) < 5e-6
    out = net(v["x"].to(dev), v["t"].to(dev)).cpu()
    assert rel_l2(out, v["out_f32"]) < REL
    assert rel_l2(out, v["out_f64"]) < max(4 * rel_l2(v["out_f32"], v["out_f64"]), 2e-6)


@pytest.mark.parametrize("precision", ["bf16x6", "fp32"])
def test_adm_other_precisions(M, dev, precision):
    net, v, _ = _net(M, dev, "concat")
    net.conv_precision = precision
    out = net(v["x"].to(dev), v["t"].to(dev)).cpu()
    assert rel_l2(out, v["out_f32"]) < REL


def test_adm_against_oracle_on_fresh_inputs(M, dev):
    """A different shape (deeper, non-square, conditional, per-sample sigma) against the oracle."""
    torch.manual_seed(3)
    cfg = dict(model_channels=16, time_embed_dim=16, output_embed_dim=32, channel_expansion=[1, 2, 2],
               number_resnet_downward_block=1, number_resnet_upward_block=3, input_channels=3, output_channels=3)
    net = M.ADM(M.ADMConfig(**cfg), conditional_embedding=torch.nn.Embedding(4, 32))
    with torch.no_grad():
        for k, w in net.state_dict().items():
            if "norm" in k or k.endswith("bias"):
                w.add_(0.2 * torch.randn_like(w))
    sd = {k: w.clone() for k, w in net.state_dict().items() if not k.startswith("conditional_embedding")}
    emb = net.conditional_embedding.weight.detach().clone()
    x = torch.randn(3, 3, 32, 64)
    t = torch.tensor([0.3, -0.8, 1.9])
    y = torch.tensor([1, 3, 0])
    ocfg = adm_ref.default_config(**cfg)
    with torch.inference_mode():
        want = adm_ref.adm_forward(sd, ocfg, x, t, emb[y])
        want64 = adm_ref.adm_forward({k: w.double() for k, w in sd.items()}, ocfg, x.double(), t.double(),
                                     emb[y].double())
    net = net.to(dev)
    got = net(x.to(dev), t.to(dev), y.to(dev)).cpu()
    assert rel_l2(got, want) < REL
    assert rel_l2(got, want64) < max(4 * rel_l2(want, want64), 2e-6)


def test_adm_rejects_unsupported_configurations(M):
    with pytest.raises(NotImplementedError, match="decoder_type"):
        M.ADM(M.ADMConfig(decoder_type=3))
    with pytest.raises(NotImplementedError, match="dimension"):
        M.ADM(M.ADMConfig(dimension=3))


@pytest.mark.parametrize("skip", ["concat", "add"])
def test_adm_fused_and_standalone_norms_agree(M, dev, skip):
    net, v, _ = _net(M, dev, skip)
    x, t = v["x"].to(dev), v["t"].to(dev)
    net.fuse_norm, net.fuse_max_cot = True, 99           # every layer folded
    fused = net(x, t).cpu()
    net.fuse_max_cot = 0                                  # tables built, no layer folded
    assert rel_l2(net(x, t).cpu(), fused) < 2e-6
    net.fuse_norm = False
    plain = net(x, t).cpu()
    assert rel_l2(fused, plain) < 2e-6
    assert rel_l2(plain, v["out_f32"]) < REL and rel_l2(fused, v["out_f32"]) < REL


def test_adm_circular_convolutions(M, dev):
    v, sd = load("adm8_circular")
    net = M.ADM(M.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16, convolution_type="circular"))
    r = net.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    net = net.to(dev)
    for fuse in (False, True):
        net.fuse_norm = fuse
        assert rel_l2(net(v["x"].to(dev), v["t"].to(dev)).cpu(), v["out_f32"]) < REL


def test_reference_adm_test_shape(M, dev):
    """The reference's own ADM test (tests/test_adm.py): default ADM(skip='add') on [4, 1, 16, 16] -- the
    middle-block attention then runs over 4 x 4 = 16 positions (generic attention path), 16 x 16 tiles."""
    torch.manual_seed(5)
    net = M.ADM(M.ADMConfig(skip_integration_type="add"))
    sd = {k: w.clone() for k, w in net.state_dict().items()}
    x, t = torch.randn(4, 1, 16, 16), torch.rand(4)
    cfg = adm_ref.default_config(skip_integration_type="add")
    with torch.inference_mode():
        want = adm_ref.adm_forward(sd, cfg, x, t)
    got = net.to(dev)(x.to(dev), t.to(dev)).cpu()
    assert got.shape == x.shape
    assert rel_l2(got, want) < REL


ADM_BLOCK_CASES = {
    "enc2d": ("ADMEncoderBlock", dict()),
    "enc2d_down": ("ADMEncoderBlock", dict(has_downsample=True)),
    "enc3d": ("ADMEncoderBlock", dict(dimension=3)),
    "enc3d_full": ("ADMEncoderBlock", dict(has_residual=True, has_attn=True, has_downsample=True, attn_residual=True, dimension=3)),
    "dec2d_skip": ("ADMDecoderBlock", dict(channels_skip=12, has_residual=True, has_attn=True, has_upsample=True)),
    "dec3d_skip_add": ("ADMDecoderBlock", dict(channels_skip=16, has_residual=True, has_upsample=True, dimension=3,
                                               skip_integration_type="add", first_norm="GroupRMS", second_norm="GroupLN")),
    "enc3d_circ": ("ADMEncoderBlock", dict(has_residual=True, has_downsample=True, dimension=3, conv_type="circular")),
}


@pytest.mark.parametrize("precision", ["fp16x3", "fp32"])
@pytest.mark.parametrize("tag", sorted(ADM_BLOCK_CASES))
def test_adm_blocks_on_fields_and_volumes(M, dev, tag, precision):
    """The reference drives ADM blocks on their own, on 2-D fields and 3-D volumes (tests/test_adm.py:7-70: cin 16,
    cout 32, cembed 24, 14^2 / 14^3 inputs; AvgPool3d, attention over the 7^3 = 343 flattened voxels).  Same classes,
    constructor arguments and state_dict keys here; outputs against the reference's."""
    v, sd_all = load("adm_blocks")
    sd = {k[len(tag) + 1:]: w for k, w in sd_all.items() if k.startswith(tag + "/")}
    cls, kw = ADM_BLOCK_CASES[tag]
    blk = getattr(M.nets, cls)(16, 32, 24, **kw)
    r = blk.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    blk = blk.to(dev).eval()
    blk.conv_precision = precision
    args = [v[tag + "/x"].to(dev), v[tag + "/te"].to(dev)] + ([v[tag + "/skip"].to(dev)] if tag + "/skip" in v else [])
    out = blk(*args).cpu()
    assert out.shape == v[tag + "/out_f32"].shape
    assert rel_l2(out, v[tag + "/out_f32"]) < REL
    assert rel_l2(out, v[tag + "/out_f64"]) < max(4 * rel_l2(v[tag + "/out_f32"], v[tag + "/out_f64"]), 2e-6)


def test_adm_block_shapes_of_the_reference_test(M, dev):
    """tests/test_adm.py of the reference, lines 7-70, on the HIP blocks (shapes only, as there)."""
    cin, cout, cembed = 16, 32, 24
    E, D = M.nets.ADMEncoderBlock, M.nets.ADMDecoderBlock
    x, te = torch.randn(1, cin, 14, 14, device=dev), torch.randn(1, cembed, device=dev)
    assert E(cin, cout, cembed).to(dev)(x, te).shape == (1, cout, 14, 14)
    assert E(cin, cout, cembed, has_downsample=True).to(dev)(x, te).shape == (1, cout, 7, 7)
    x3 = torch.randn(1, cin, 14, 14, 14, device=dev)
    assert E(cin, cout, cembed, dimension=3).to(dev)(x3, te).shape == (1, cout, 14, 14, 14)
    assert E(cin, cout, cembed, has_residual=True, has_attn=True, has_downsample=True, attn_residual=True,
             dimension=3).to(dev)(x3, te).shape == (1, cout, 7, 7, 7)
    assert D(cin, cout, cembed).to(dev)(x, te).shape == (1, cout, 14, 14)
    assert D(cin, cout, cembed, has_upsample=True).to(dev)(x, te).shape == (1, cout, 28, 28)
    assert D(cin, cout, cembed, has_residual=True, has_attn=True, has_upsample=True).to(dev)(x, te).shape == (1, cout, 28, 28)
    xskip = torch.randn(1, 12, 14, 14, device=dev)
    assert D(cin, cout, cembed, 12, has_residual=True, has_attn=True, has_upsample=True).to(dev)(x, te, xskip).shape == (1, cout, 28, 28)
    with pytest.raises(NotImplementedError, match="the reference's own ADM cannot run on volumes"):
        M.ADM(M.ADMConfig(dimension=3))
