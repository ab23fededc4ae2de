"""Kernel-level parity on a real MI355X: each C-ABI entry point against the CPU oracle /
the reference's torch-CPU arithmetic on the same seeded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import karras_ref as K  # noqa: E402
from oracle import punetg_ref  # noqa: E402
from tests.golden_util import rel_l2  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _ops():
    from diffsci_amd import ops
    return ops


def _coef(**kw):
    from diffsci_amd._native import EvalCoef
    base = dict(c_out=1.0, c_skip=0.0, sigma_sq=1.0, neg_mult=-1.0, neg_lang=0.0, guidance=1.0,
                one_minus_guidance=0.0, input_kind=0, stochastic=0)
    base.update(kw)
    return EvalCoef(**base)


def test_library_reports_gfx950(dev):
    import ctypes
    from diffsci_amd import _native as N
    cu, lds = ctypes.c_int(), ctypes.c_int()
    name = ctypes.create_string_buffer(64)
    N.check(N.lib().ds_device_info(ctypes.byref(cu), ctypes.byref(lds), name, 64), "ds_device_info")
    assert name.value.decode().startswith("gfx950"), name.value
    assert cu.value == 256


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 4096, 2 * 128 * 128 + 4])
def test_stepper_kernels_bit_exact(dev, n):
    """The step kernels follow the reference's op order: bit-identical to torch-CPU fp32."""
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    x, f1, f2, fu1, fu2, eps = (torch.randn(n, generator=g) * s for s in (80.0, 1.0, 1.0, 1.0, 1.0, 1.0))
    sig1, sig2 = torch.tensor(57.586), torch.tensor(40.786)
    dt = sig2 - sig1
    rows = []
    for sg in (sig1, sig2):
        cs, co, ci, cn = K.edm_precond(sg)
        rows.append(dict(c_out=float(co), c_skip=float(cs), sigma_sq=float(sg ** 2), neg_mult=float(-(sg * (1 + 0 * sg)))))
    ci2 = float(K.edm_precond(sig2)[2])

    def drift_cpu(xx, ff, fuu, r, g_):
        Fv = ff if fuu is None else (1 - g_) * fuu + g_ * ff
        D = r["c_out"] * Fv + r["c_skip"] * xx
        sc = (D - xx) / r["sigma_sq"]
        return r["neg_mult"] * sc

    D = lambda t: t.to(dev)  # noqa: E731
    for g_, u1, u2 in ((1.0, None, None), (2.0, fu1, fu2)):
        k1 = _coef(guidance=g_, one_minus_guidance=1 - g_, **rows[0])
        k2 = _coef(guidance=g_, one_minus_guidance=1 - g_, **rows[1])
        d1 = drift_cpu(x, f1, u1, rows[0], g_)
        xe = x + float(dt) * d1
        xo, xi = torch.empty(n, device=dev), torch.empty(n, device=dev)
        ops.euler(D(x), D(f1), k1, float(dt), fu=None if u1 is None else D(u1), x_out=xo, xin_out=xi, c_in_next=ci2)
        assert torch.equal(xo.cpu(), xe)
        assert torch.equal(xi.cpu(), ci2 * xe)
        d2 = drift_cpu(xe, f2, u2, rows[1], g_)
        want = x + (0.5 * (d1 + d2)) * float(dt)
        ops.heun(D(x), D(f1), k1, D(f2), k2, float(dt), f1u=None if u1 is None else D(u1),
                 f2u=None if u2 is None else D(u2), x_out=xo, xin_out=xi, c_in_next=0.25)
        assert torch.equal(xo.cpu(), want)
        assert torch.equal(xi.cpu(), 0.25 * want)
        got = ops.drift(D(x), D(f1), k1, fu=None if u1 is None else D(u1))
        assert torch.equal(got.cpu(), d1)
    # Euler-Maruyama move and churn
    k = _coef(stochastic=1, neg_lang=-0.7 * 57.586, **rows[0])
    sc = ((rows[0]["c_out"] * f1 + rows[0]["c_skip"] * x) - x) / rows[0]["sigma_sq"]
    d = rows[0]["neg_mult"] * sc
    d = d + (-0.7 * 57.586) * sc
    dtf = float(dt)
    want = x + d * dtf + (1.3 * eps) * 0.9
    xo = torch.empty(n, device=dev)
    ops.euler(D(x), D(f1), k, dtf, x_out=xo, eps=D(eps), noise_coef=1.3, sqrt_abs_dt=0.9)
    assert torch.equal(xo.cpu(), want)
    xh, xi = torch.empty(n, device=dev), torch.empty(n, device=dev)
    ops.churn(D(x), D(eps), 3.25, xhat_out=xh, xin_out=xi, c_in=0.125)
    assert torch.equal(xh.cpu(), x + 3.25 * eps)
    assert torch.equal(xi.cpu(), 0.125 * (x + 3.25 * eps))
    assert torch.equal(ops.scale(D(x), 80.0).cpu(), x * 80.0)
    assert torch.equal(ops.add(D(x), D(f1)).cpu(), x + f1)


@pytest.mark.parametrize("shape", [(2, 8, 32, 32), (3, 5, 8, 8), (2, 4, 64, 64), (1, 3, 128, 128),
                                   (2, 2, 256, 256), (2, 3, 5, 7), (1, 2, 2, 2)])
@pytest.mark.parametrize("kind", [0, 1])
def test_inorm_silu(dev, shape, kind):
    ops = _ops()
    g = torch.Generator().manual_seed(sum(shape) + kind)
    x = torch.randn(shape, generator=g) * 3 + 0.5
    C = shape[1]
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    if kind == 0:
        want = F.silu(F.group_norm(x, C, w, b, 1e-5))
    else:
        want = F.silu(punetg_ref.group_rms_norm(x, w, b))
    got = ops.inorm_silu(x.to(dev), w.to(dev), b.to(dev), kind).cpu()
    # fp32 tolerance: statistics are reduced in a different order than torch's CPU kernels
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6)
    # in-place form
    xd = x.to(dev)
    ops.inorm_silu(xd, w.to(dev), b.to(dev), kind, out=xd)
    torch.testing.assert_close(xd.cpu(), want, rtol=2e-5, atol=2e-6)


CONV_CASES = [
    # B, Cin, Cout, H, W, ks, mode
    (2, 8, 8, 32, 32, 3, 0),
    (1, 1, 64, 16, 40, 3, 0),      # convin-like, ragged width
    (2, 64, 1, 24, 24, 3, 0),      # convout-like
    (2, 16, 24, 9, 13, 3, 0),      # nothing divides anything
    (1, 64, 128, 16, 16, 3, 1),    # DownSampler: maxpool fused in the load
    (1, 24, 12, 16, 32, 3, 2),     # UpSampler: nearest fused in the load
    (2, 32, 96, 8, 8, 1, 0),       # attention in_proj
    (1, 70, 33, 8, 40, 1, 0),
    (1, 128, 128, 64, 64, 3, 0),   # multiple chunks and channel tiles
    (2, 32, 64, 16, 16, 3, 0),     # narrow maps: the 16 x 16 pixel-tile geometry
    (1, 24, 40, 20, 48, 3, 0),     # W = 48: three 16-wide tiles instead of two 32-wide
    (1, 24, 12, 16, 16, 3, 2),     # UpSampler onto a 16 x 16 map
    (2, 16, 32, 8, 8, 3, 1),       # DownSampler onto an 8 x 8 map
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(dev, case):
    ops = _ops()
    B, Cin, Cout, H, W, ks, mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    Hin, Win = (2 * H, 2 * W) if mode == 1 else ((H // 2, W // 2) if mode == 2 else (H, W))
    x = torch.randn(B, Cin, Hin, Win, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    bias = torch.randn(Cout, generator=g)
    shift = torch.randn(B, Cout, generator=g)
    r1, r2 = torch.randn(B, Cout, H, W, generator=g), torch.randn(B, Cout, H, W, generator=g)
    src = F.max_pool2d(x, 2) if mode == 1 else (F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == 2 else x)
    ref64 = F.conv2d(src.double(), w.double(), bias.double(), padding="same")
    want = ref64 + shift.double()[:, :, None, None] + r1.double() + r2.double()
    wp = ops.pack_conv_weight(w.to(dev))
    got = ops.conv2d(x.to(dev), wp, Cout, ks, bias=bias.to(dev), shift=shift.to(dev), res1=r1.to(dev),
                     res2=r2.to(dev), load_mode=mode).cpu()
    # exact-fp32 MFMA = one k-ordered fmaf chain per output (K up to 1152 here); torch's CPU conv
    # sums in blocks, so its worst-case rounding error is a few times smaller.  Bound ours by 8x.
    err = (got.double() - want).abs().max().item()
    ref32 = F.conv2d(src, w, bias, padding="same") + shift[:, :, None, None] + r1 + r2
    err32 = (ref32.double() - want).abs().max().item()
    assert err <= max(8 * err32, 1e-5), (err, err32)
    # plain variant: no epilogue terms, broadcast shift
    got = ops.conv2d(x.to(dev), wp, Cout, ks, shift=shift[:1].to(dev).contiguous(), load_mode=mode).cpu()
    want2 = F.conv2d(src.double(), w.double(), None, padding="same") + shift.double()[:1, :, None, None]
    assert (got.double() - want2).abs().max().item() <= max(8 * err32, 1e-5)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] == 3] + [(2, 256, 256, 32, 32, 3, 0), (1, 40, 72, 20, 36, 3, 0)])
@pytest.mark.parametrize("prec", ["bf16x6", "fp16x3"])
def test_conv2d_split_mfma_is_fp32_accurate(dev, case, prec):
    """The split-bf16 MFMA path: error against fp64 at the level of torch's own fp32 convolution
    (and no worse than the exact-fp32 MFMA kernel's bound) -- it is fp32 arithmetic, not bf16."""
    ops = _ops()
    B, Cin, Cout, H, W, ks, mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 1)
    Hin, Win = (2 * H, 2 * W) if mode == 1 else ((H // 2, W // 2) if mode == 2 else (H, W))
    x = torch.randn(B, Cin, Hin, Win, generator=g) * 3.0
    x[0, 0, 0, :4] = torch.tensor([1e-30, -3e-39, 6.0e4, -1e4])           # tiny / denormal / large operands
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    bias = torch.randn(Cout, generator=g)
    shift = torch.randn(B, Cout, generator=g)
    r1 = torch.randn(B, Cout, H, W, generator=g)
    src = F.max_pool2d(x, 2) if mode == 1 else (F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == 2 else x)
    want = F.conv2d(src.double(), w.double(), bias.double(), padding="same") + shift.double()[:, :, None, None] + r1.double()
    ref32 = F.conv2d(src, w, bias, padding="same") + shift[:, :, None, None] + r1
    pw = ops.pack_conv(w.to(dev), prec)
    assert pw.kind == prec
    got = ops.conv(x.to(dev), pw, bias=bias.to(dev), shift=shift.to(dev), res1=r1.to(dev), load_mode=mode).cpu()
    err = (got.double() - want).abs().max().item()
    err32 = (ref32.double() - want).abs().max().item()
    rel = rel_l2(got, want)
    rel32 = rel_l2(ref32, want)
    assert err <= max(4 * err32, 1e-5), (err, err32)
    assert rel <= max(3 * rel32, 3e-7), (rel, rel32)
    exact = ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp32"), bias=bias.to(dev), shift=shift.to(dev),
                     res1=r1.to(dev), load_mode=mode).cpu()
    assert rel <= 2 * rel_l2(exact, want) + 1e-7


CONV1X1_CASES = [
    # B, Cin, Cout, H, W, mode (0 plain, 2 nearest-up, 3 avg-pool)
    (2, 32, 96, 8, 8, 0),          # attention in_proj
    (1, 70, 33, 8, 40, 0),         # ragged channels
    (2, 16, 24, 9, 13, 0),         # nothing divides anything
    (3, 128, 128, 64, 64, 0),      # ADM residual projection, several tiles per XCD
    (1, 1, 8, 16, 32, 0),          # a single input channel (one clamped chunk)
    (2, 48, 64, 16, 32, 2),        # decoder block: nearest-up folded into the load
    (2, 40, 72, 12, 20, 3),        # encoder block: AvgPool2d(2) folded into the load
    (1, 256, 512, 32, 32, 3),
    (1, 1024, 256, 16, 16, 2),     # 64 steps, 16 x 16 tile geometry
    (2, 64, 64, 16, 16, 0),
    (1, 32, 48, 24, 48, 2),        # W = 48
    (2, 16, 16, 8, 8, 3),
]


@pytest.mark.parametrize("case", CONV1X1_CASES)
def test_conv1x1_fp16x3(dev, case):
    """ds_conv1x1_h3: fp32-level error against fp64, never worse than twice the exact-fp32 MFMA kernel."""
    ops = _ops()
    B, Cin, Cout, H, W, mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 2)
    Hin, Win = (2 * H, 2 * W) if mode == 3 else ((H // 2, W // 2) if mode == 2 else (H, W))
    x = torch.randn(B, Cin, Hin, Win, generator=g) * 3.0
    x[0, 0, 0, :4] = torch.tensor([1e-30, -3e-39, 6.0e4, -1e4])
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    bias, shift = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
    r1, r2 = torch.randn(B, Cout, H, W, generator=g), torch.randn(B, Cout, H, W, generator=g)
    src = F.avg_pool2d(x, 2) if mode == 3 else (F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == 2 else x)
    want = (F.conv2d(src.double(), w.double(), bias.double()) + shift.double()[:, :, None, None] + r1.double() + r2.double())
    ref32 = F.conv2d(src, w, bias) + shift[:, :, None, None] + r1 + r2
    pw = ops.pack_conv(w.to(dev), "fp16x3")
    assert pw.kind == "fp16x3" and pw.ks == 1
    got = ops.conv(x.to(dev), pw, bias=bias.to(dev), shift=shift.to(dev), res1=r1.to(dev), res2=r2.to(dev),
                   load_mode=mode).cpu()
    err, err32 = (got.double() - want).abs().max().item(), (ref32.double() - want).abs().max().item()
    rel, rel32 = rel_l2(got, want), rel_l2(ref32, want)
    assert err <= max(4 * err32, 1e-5), (err, err32)
    assert rel <= max(3 * rel32, 3e-7), (rel, rel32)
    if mode != 3:
        exact = ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp32"), bias=bias.to(dev), shift=shift.to(dev),
                         res1=r1.to(dev), res2=r2.to(dev), load_mode=mode).cpu()
        assert rel <= 2 * rel_l2(exact, want) + 1e-7
    # no epilogue terms
    got = ops.conv(x.to(dev), pw, load_mode=mode).cpu()
    assert rel_l2(got, F.conv2d(src.double(), w.double())) <= max(3 * rel_l2(F.conv2d(src, w), F.conv2d(src.double(), w.double())), 3e-7)


def test_split_mfma_handles_exact_fp16_ties(dev):
    """Regression: inputs lying EXACTLY halfway between two fp16 values.  hipcc rounds the stored
    hi piece (v_cvt_pk_f16_f32) and the remainder's reference (v_cvt_f16_f32) with different tie
    rules unless the split derives both from the same bits; the symptom was hi+lo off by one fp16
    ulp (2^-11 relative) on ~1 value in 8000 -- invisible on random data, so force it here."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 1, 32, 16, 32
    k = torch.randint(1024, 2048, (B, C, H, W), generator=g).float()
    sign = torch.where(torch.rand(B, C, H, W, generator=g) < 0.5, -1.0, 1.0)
    x = sign * (k + 0.5) * 2.0 ** -13            # every element is a tie between fp16 neighbours (in [0.125, 0.25))
    assert torch.equal(x.half().float() != x, torch.ones_like(x, dtype=torch.bool))
    w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)
    want = F.conv2d(x.double(), w.double(), padding="same")
    ref32 = F.conv2d(x, w, padding="same")
    for prec in ("fp16x3", "bf16x6"):
        got = ops.conv(x.to(dev), ops.pack_conv(w.to(dev), prec)).cpu()
        assert rel_l2(got, want) <= max(3 * rel_l2(ref32, want), 3e-7), prec
    # attention: tie-valued queries
    E, L = 32, 64
    qkv = torch.randn(1, 3 * E, L, generator=g)
    qkv[:, :E] = (sign[0, :E, 0, :1] * (k[0, :E, :2, :].reshape(E, 64) + 0.5) * 2.0 ** -13) / math.sqrt(1.0 / E)
    q, kk, v = (t.transpose(1, 2).double() for t in qkv.split(E, dim=1))
    att = torch.softmax((q * math.sqrt(1.0 / E)) @ kk.transpose(1, 2), dim=-1) @ v
    got = ops.attention(qkv.to(dev), E, precision="fp16x3").cpu()
    assert rel_l2(got, att.transpose(1, 2)) < 2e-6


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
@pytest.mark.parametrize("B,E,L", [(2, 32, 64), (1, 64, 96), (2, 128, 256), (2, 256, 1024), (2, 384, 64), (3, 512, 256)])
def test_attention(dev, B, E, L, precision):
    ops = _ops()
    g = torch.Generator().manual_seed(E + L)
    qkv = torch.randn(B, 3 * E, L, generator=g)
    q, k, v = (t.transpose(1, 2).double() for t in qkv.split(E, dim=1))    # [B, L, E]
    att = torch.softmax((q * math.sqrt(1.0 / E)) @ k.transpose(1, 2), dim=-1) @ v
    want = att.transpose(1, 2)
    got = ops.attention(qkv.to(dev), E, precision=precision).cpu()
    assert rel_l2(got, want) < 2e-6
    assert (got.double() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,E,L", [(2, 32, 16), (1, 48, 35), (2, 256, 25), (3, 32, 4), (1, 96, 64)])
def test_attention_generic_lengths(dev, B, E, L):
    """Sequence lengths / head widths outside the MFMA kernels' grid go through ds_attention_generic."""
    ops = _ops()
    g = torch.Generator().manual_seed(E + L)
    qkv = torch.randn(B, 3 * E, L, generator=g)
    q, k, v = (t.transpose(1, 2).double() for t in qkv.split(E, dim=1))
    want = (torch.softmax((q * math.sqrt(1.0 / E)) @ k.transpose(1, 2), dim=-1) @ v).transpose(1, 2)
    for precision in ("fp16x3", "fp32"):
        got = ops.attention(qkv.to(dev), E, precision=precision).cpu()
        assert rel_l2(got, want) < 2e-6


def test_linear_and_fourier(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x, w, b = torch.randn(7, 65, generator=g), torch.randn(40, 65, generator=g), torch.randn(40, generator=g)
    for act, fn in ((0, lambda t: t), (1, F.silu), (2, F.relu)):
        got = ops.linear(x.to(dev), w.to(dev), b.to(dev), act=act).cpu()
        torch.testing.assert_close(got, fn(F.linear(x, w, b)), rtol=1e-5, atol=1e-5)
    t = torch.tensor([2.19, -3.1, 0.0, 0.37])
    W = torch.randn(32, generator=g) * 30
    want = punetg_ref.fourier_features(t, W)
    got = ops.fourier_features(t.to(dev), W.to(dev)).cpu()
    # arguments reach ~1e3 rad: compare against the fp64 sine of the same fp32 argument
    arg = ((t[:, None] * torch.tensor(2 * math.pi, dtype=torch.float32)) * W).double()
    exact = torch.cat([torch.sin(arg), torch.cos(arg)], -1)
    assert (got.double() - exact).abs().max().item() < 1e-7
    assert (got - want).abs().max().item() < 5e-7
    add = torch.randn(1, 64, generator=g)
    got = ops.fourier_features(t.to(dev), W.to(dev), add=add.to(dev)).cpu()
    assert (got - (want + add)).abs().max().item() < 5e-7


def test_bad_arguments_fail_loudly(dev):
    ops = _ops()
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.scale(torch.zeros(4), 2.0)
    with pytest.raises(TypeError):
        ops.scale(torch.zeros(4, device=dev, dtype=torch.float64), 2.0)
    import ctypes
    from diffsci_amd import _native as N
    q = torch.zeros(1, 96, 40, device=dev)
    assert N.lib().ds_attention(q.data_ptr(), q.data_ptr(), 1, 32, 40, None) != 0      # the MFMA kernels want L % 32 == 0
    assert b"multiple of 32" in N.lib().ds_last_error()
    del ctypes
    with pytest.raises(ValueError):
        ops.conv2d(torch.zeros(1, 4, 8, 8, device=dev), torch.zeros(10, device=dev), 4, 3)


FUSED_NORM_CASES = [
    # B, Cin, Cout, H, W, mode
    (2, 16, 24, 32, 32, 0),
    (1, 40, 72, 20, 36, 0),        # ragged channels (last chunk 8 of 16), ragged tiles
    (2, 64, 64, 16, 16, 0),        # 16 x 16 tile geometry
    (1, 24, 12, 16, 32, 2),        # nearest-upsampling load
    (1, 8, 8, 9, 13, 0),           # ragged width: element-wise epilogue path
]


@pytest.mark.parametrize("case", FUSED_NORM_CASES)
def test_conv_tile_stats_and_fused_prenorm(dev, case):
    """Norm fusion around the fp16x3 convolution: (1) the epilogue's per-(channel, tile) sums add up
    to the plane statistics of the stored output; (2) the table built from them reproduces the norm
    parameters; (3) a convolution with that table applied in its loader equals norm+SiLU followed by
    the plain convolution."""
    ops = _ops()
    B, Cin, Cout, H, W, mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 7)
    Hin, Win = (H // 2, W // 2) if mode == 2 else (H, W)
    x0 = torch.randn(B, Cin, Hin, Win, generator=g) * 2.0 + 0.3
    w0 = torch.randn(Cin, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    res = torch.randn(B, Cin, Hin, Win, generator=g)
    nt = ops.conv_tile_count(Hin, Win)
    ts = torch.full((B, Cin, nt, 4), float("nan"), device=dev)
    # producer: y = conv(x0) + res, leaving tile statistics
    y = ops.conv(x0.to(dev), ops.pack_conv(w0.to(dev), "fp16x3"), res1=res.to(dev), tile_stats=ts)
    yc = y.cpu().double()
    def sums(t):                                      # (K, S, Q, n) per tile -> plane sums of x and x^2
        K, S, Q, n = t.cpu().double().unbind(-1)
        return (n * K + S).sum(-1), (Q + 2 * K * S + n * K * K).sum(-1), n.sum(-1)
    assert torch.isfinite(ts).all()
    sx, sxx, n = sums(ts)
    assert torch.equal(n, torch.full_like(n, Hin * Win))
    torch.testing.assert_close(sx, yc.sum(dim=(2, 3)), rtol=1e-6, atol=1e-4)
    torch.testing.assert_close(sxx, (yc * yc).sum(dim=(2, 3)), rtol=1e-6, atol=1e-4)
    # the 1x1 kernel leaves the same statistics
    w1 = torch.randn(Cin, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    ts1 = torch.zeros((B, Cin, nt, 4), device=dev)
    y1 = ops.conv(x0.to(dev), ops.pack_conv(w1.to(dev), "fp16x3"), tile_stats=ts1).cpu().double()
    torch.testing.assert_close(sums(ts1)[1], (y1 * y1).sum(dim=(2, 3)), rtol=1e-6, atol=1e-4)
    wn, bn = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    pw = ops.pack_conv(w.to(dev), "fp16x3")
    up = (lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")) if mode == 2 else (lambda t: t)
    for kind in (0, 1):
        # PUNetG: per-(b, c) norms
        tab_full = ops.inorm_table(ts, wn.to(dev), bn.to(dev), kind, Hin * Win).cpu().double()
        assert tab_full.shape[1] == ops.table_channels(Cin) and not tab_full[:, Cin:, :3].any()   # zero rows pad the last chunk
        # fourth column: 2^-k, the sample's activation exponent -- one power of two per sample (padding rows included) that puts
        # a bound on |(x - M)*A + C| at 2^13
        inv = tab_full[..., 3]
        assert (inv == inv[:, :1]).all() and (torch.log2(inv[:, 0]) % 1 == 0).all()
        tab = tab_full[:, :Cin]
        arg = ((yc - tab[..., 0, None, None]) * tab[..., 1, None, None] + tab[..., 2, None, None]).abs().amax(dim=(1, 2, 3))
        assert (arg / inv[:, 0] < 2.0 ** 14).all() and (arg / inv[:, 0] > 2.0 ** 5).all()
        mean = yc.mean(dim=(2, 3)) if kind == 0 else torch.zeros(B, Cin, dtype=torch.float64)
        den = (yc.var(dim=(2, 3), unbiased=False) + 1e-5).sqrt() if kind == 0 else ((yc * yc).mean(dim=(2, 3)) + 1e-5).sqrt()
        torch.testing.assert_close(tab[..., 0], mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(tab[..., 1], wn.double() / den, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(tab[..., 2], bn.double().expand(B, Cin), rtol=0, atol=0)
        a = ops.inorm_silu(y, wn.to(dev), bn.to(dev), kind)
        want = ops.conv(a, pw, load_mode=mode).cpu()
        fused = ops.conv(y, pw, load_mode=mode, prenorm=ops.inorm_table(ts, wn.to(dev), bn.to(dev), kind, Hin * Win)).cpu()
        assert rel_l2(fused, want) < 2e-6, (kind, rel_l2(fused, want))
        ref64 = F.conv2d(up(F.silu(((yc - mean[..., None, None]) / den[..., None, None]) * wn.double()[None, :, None, None]
                                   + bn.double()[None, :, None, None])), w.double(), padding="same")
        assert rel_l2(fused, ref64) < 3e-6
        # ADM: per-sample norms (+ FiLM for the RMS kind)
        film = torch.randn(B, 2 * Cin, generator=g)
        tab = ops.gnorm1_table(ts, wn.to(dev), bn.to(dev), kind, Cin * Hin * Win, film=film.to(dev) if kind else None)
        st = ops.gnorm1_stats(y, kind)
        a = ops.gnorm1_apply(y, st, wn.to(dev), bn.to(dev), kind, film=film.to(dev) if kind else None)
        want = ops.conv(a, pw, load_mode=mode).cpu()
        fused = ops.conv(y, pw, load_mode=mode, prenorm=tab).cpu()
        assert rel_l2(fused, want) < 2e-6, ("g1", kind, rel_l2(fused, want))
    # concat of two tensors: statistics are additive
    tab2 = ops.gnorm1_table(ts, torch.cat([wn, wn]).to(dev), torch.cat([bn, bn]).to(dev), 0, 2 * Cin * Hin * Win,
                            stats_b=ts).cpu()
    tab1 = ops.gnorm1_table(ts, wn.to(dev), bn.to(dev), 0, Cin * Hin * Win).cpu()
    torch.testing.assert_close(tab2[:, :Cin, :3], tab1[:, :Cin, :3], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(tab2[:, Cin:2 * Cin, :3], tab1[:, :Cin, :3], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", [(2, 16, 24, 32, 32, 0), (1, 40, 8, 20, 36, 0), (2, 8, 16, 16, 16, 0), (1, 8, 8, 9, 13, 0),
                                  (1, 16, 32, 16, 32, 1), (1, 24, 12, 16, 32, 2), (1, 4, 4, 8, 8, 0)])
def test_conv_circular_padding(dev, case):
    """DS_PAD_CIRCULAR: periodic padding (CircularConv2d) applied to the plain / pooled / upsampled image."""
    ops = _ops()
    B, Cin, Cout, H, W, mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 11)
    Hin, Win = (2 * H, 2 * W) if mode == 1 else ((H // 2, W // 2) if mode == 2 else (H, W))
    x = torch.randn(B, Cin, Hin, Win, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    src = F.max_pool2d(x, 2) if mode == 1 else (F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == 2 else x)
    pad = F.pad(F.pad(src.double(), (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1), mode="circular")
    want = F.conv2d(pad, w.double(), bias.double())
    ref32 = F.conv2d(F.pad(F.pad(src, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1), mode="circular"), w, bias)
    pw = ops.pack_conv(w.to(dev), "fp16x3")
    got = ops.conv(x.to(dev), pw, bias=bias.to(dev), load_mode=mode, circular=True).cpu()
    assert rel_l2(got, want) <= max(3 * rel_l2(ref32, want), 3e-7)
    zero = ops.conv(x.to(dev), pw, bias=bias.to(dev), load_mode=mode).cpu()
    assert rel_l2(zero, want) > 1e-3                                    # and it is not the zero-padded result
    with pytest.raises(NotImplementedError, match="periodic padding"):
        ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp32"), load_mode=mode, circular=True)


@pytest.mark.parametrize("case", [(2, 64, 1, 32, 32, False), (1, 13, 3, 20, 36, False), (2, 8, 4, 9, 13, False),
                                  (1, 128, 3, 64, 128, False), (2, 16, 2, 16, 16, True), (1, 5, 4, 18, 70, True)])
def test_conv_direct_small_cout(dev, case):
    """ds_conv2d_direct (output layers, Cout <= 4): exact fp32 FMA chains."""
    ops = _ops()
    B, Cin, Cout, H, W, circ = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 13)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    if circ:
        pad = lambda t: F.pad(F.pad(t, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1), mode="circular")   # noqa: E731
        want, ref32 = F.conv2d(pad(x.double()), w.double(), bias.double()), F.conv2d(pad(x), w, bias)
    else:
        want, ref32 = F.conv2d(x.double(), w.double(), bias.double(), padding="same"), F.conv2d(x, w, bias, padding="same")
    got = ops.conv_direct(x.to(dev), w.to(dev), bias.to(dev), circular=circ).cpu()
    err, err32 = (got.double() - want).abs().max().item(), (ref32.double() - want).abs().max().item()
    assert err <= max(8 * err32, 1e-5), (err, err32)
    assert rel_l2(got, want) <= max(4 * rel_l2(ref32, want), 3e-7)
    got = ops.conv_direct(x.to(dev), w.to(dev), None, circular=circ).cpu()
    nb64, nb32 = want - bias.double()[None, :, None, None], ref32 - bias[None, :, None, None]
    assert rel_l2(got, nb64) <= max(4 * rel_l2(nb32, nb64), 6e-7)
    with pytest.raises(RuntimeError, match="1..4 supported"):
        ops.conv_direct(x.to(dev), torch.randn(5, Cin, 3, 3, device=dev))


@pytest.mark.parametrize("env", [{}, {"DS_CONV_SHAPE": "32"}, {"DS_CONV_WAVES16": "8"}, {"DS_CONV_TWO": "2", "DS_CONV1_TWO": "2"},
                                 {"DS_CONV_TWO": "0", "DS_CONV1_TWO": "0"}],
                         ids=["shipped", "mfma-32x32x16", "16x16x32-eight-waves", "two-channel-tiles-everywhere", "one-channel-tile"])
def test_convolution_family_fuzz(env):
    """A short run of tools/conv_fuzz.py: random shapes through every load / padding / fusion combination -- with the
    shipped kernel selection and with the alternatives the library keeps behind environment switches (the 32x32x16
    instruction shape incl. its eight-wave fused-loader instances; the eight-wave form of the 16x16x32 variant; the
    two-channel-tiles-per-workgroup kernel on every even tile count and both loaders, and off), which are read once per
    process: hence the subprocess."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_fuzz.py"), "--n", "80", "--seed", "7"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all 80 cases passed" in r.stdout


@pytest.mark.parametrize("case", [(2, 16, 24, 32, 32), (1, 40, 72, 16, 16), (1, 8, 8, 10, 14), (2, 24, 130, 64, 32)])
def test_conv_upsampled_residual(dev, case):
    """DS_RES1_UPSAMPLED: the residual is stored at half resolution and added nearest-upsampled
    (convresidual(upsample(x)) = upsample(convresidual(x)) for ADM's 1x1 residual branch)."""
    ops = _ops()
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 17)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    rlow = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    want = F.conv2d(x.double(), w.double(), bias.double(), padding="same") + F.interpolate(rlow.double(), scale_factor=2.0, mode="nearest")
    ref32 = F.conv2d(x, w, bias, padding="same") + F.interpolate(rlow, scale_factor=2.0, mode="nearest")
    ts = torch.zeros(B, Cout, ops.conv_tile_count(H, W), 4, device=dev)
    got = ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp16x3"), bias=bias.to(dev), res1=rlow.to(dev), res1_upsampled=True,
                   tile_stats=ts).cpu()
    assert rel_l2(got, want) <= max(3 * rel_l2(ref32, want), 3e-7)
    K, S, Q, n = ts.cpu().double().unbind(-1)
    torch.testing.assert_close((n * K + S).sum(-1), want.sum(dim=(2, 3)), rtol=1e-5, atol=1e-3)
    with pytest.raises(ValueError, match="res1_upsampled"):
        ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp16x3"), res1=rlow.to(dev)[:, :, :-1], res1_upsampled=True)


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, Hl, Wl, circular, prenorm, res1 ("hi" | "low" | None)
    (2, 16, 64, 8, 32, False, False, None),
    (1, 40, 72, 16, 16, False, False, "hi"),        # 16 x 16 tiles, ragged channel tiles
    (2, 24, 130, 16, 64, True, False, "low"),       # periodic padding, low-resolution residual
    (1, 50, 33, 32, 32, False, True, "hi"),         # normalisation + SiLU folded into the loader
    (1, 16, 16, 48, 16, True, True, None),          # 16 x 16 tiles, three tile rows, one chunk
    (3, 96, 64, 8, 64, False, False, "hi"),         # six chunks: two rounds of the weight-slot pattern
])
def test_conv_upsample_parity_kernel(dev, case):
    """ds_conv2d_h3_up (upsample x2 + 3x3 conv as four 2x2 parity kernels at low resolution) against the fp64
    definition, and its tile statistics."""
    ops = _ops()
    B, Cin, Cout, Hl, Wl, circ, pre, res = case
    H, W = 2 * Hl, 2 * Wl
    g = torch.Generator().manual_seed(sum(int(v or 0) * (i + 3) for i, v in enumerate(case[:5])))
    x = torch.randn(B, Cin, Hl, Wl, generator=g) * 2 + 0.3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    shift = torch.randn(B, Cout, generator=g)
    res2 = torch.randn(B, Cout, H, W, generator=g)
    r1 = None if res is None else torch.randn(B, Cout, *((H, W) if res == "hi" else (Hl, Wl)), generator=g)
    assert ops.N.lib().ds_conv2d_h3_up_supported(Hl, Wl)

    def definition(dt):
        xx = x.to(dt)
        if pre:
            tab = table.to(dt)[:, :Cin]
            xx = F.silu((xx - tab[:, :, 0, None, None]) * tab[:, :, 1, None, None] + tab[:, :, 2, None, None])
        up = F.interpolate(xx, scale_factor=2.0, mode="nearest")
        if circ:
            y = F.conv2d(F.pad(up, (1, 1, 1, 1), mode="circular"), w.to(dt), bias.to(dt))
        else:
            y = F.conv2d(up, w.to(dt), bias.to(dt), padding=1)
        y = y + shift.to(dt)[:, :, None, None]
        if r1 is not None:
            y = y + (r1.to(dt) if res == "hi" else F.interpolate(r1.to(dt), scale_factor=2.0, mode="nearest"))
        return y + res2.to(dt)

    table = None
    if pre:
        table = torch.zeros(B, ops.table_channels(Cin), 4)
        table[:, :Cin, 0] = torch.randn(B, Cin, generator=g) * 0.2
        table[:, :Cin, 1] = torch.rand(B, Cin, generator=g) + 0.5
        table[:, :Cin, 2] = torch.randn(B, Cin, generator=g) * 0.2
    want, ref32 = definition(torch.float64), definition(torch.float32)
    pw = ops.pack_conv(w.to(dev), "fp16x3", upsampled=True)
    assert pw.up is not None and pw.up_wshift == pw.wshift - 2
    ts = torch.zeros(B, Cout, ops.conv_tile_count(H, W), 4, device=dev)
    kw = dict(bias=bias.to(dev), shift=shift.to(dev), res1=None if r1 is None else r1.to(dev), res1_upsampled=res == "low",
              res2=res2.to(dev), load_mode=ops.N.DS_LOAD_UPSAMPLE2, circular=circ,
              prenorm=None if table is None else table.to(dev))
    got = ops.conv(x.to(dev), pw, tile_stats=ts, **kw).cpu()
    assert rel_l2(got, want) <= max(3 * rel_l2(ref32, want), 3e-7)
    K, S, Q, n = ts.cpu().double().unbind(-1)
    assert torch.equal(n.sum(-1), torch.full((B, Cout), float(H * W), dtype=torch.float64))
    torch.testing.assert_close((n * K + S).sum(-1), want.sum(dim=(2, 3)), rtol=1e-5, atol=2e-3)
    mean = (n * K + S).sum(-1) / (H * W)
    m2 = (Q + 2 * (K - mean[..., None]) * S + n * (K - mean[..., None]) ** 2).sum(-1)
    torch.testing.assert_close(m2, ((want - want.mean(dim=(2, 3), keepdim=True)) ** 2).sum(dim=(2, 3)), rtol=2e-5, atol=1e-3)
    # the generic gather kernel (no parity packing) computes the same thing
    generic = ops.conv(x.to(dev), ops.pack_conv(w.to(dev), "fp16x3"), **kw).cpu()
    assert rel_l2(got, generic.double()) < 5e-7


def test_conv_upsample_parity_fallback_and_errors(dev):
    """Shapes that are not whole tiles take ds_conv2d_h3's UPSAMPLE2 loader; the raw entry point refuses them."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 12, 20, generator=g)
    w = torch.randn(8, 16, 3, 3, generator=g) / 12
    assert not ops.N.lib().ds_conv2d_h3_up_supported(12, 20)
    pw = ops.pack_conv(w.to(dev), "fp16x3", upsampled=True)
    got = ops.conv(x.to(dev), pw, load_mode=ops.N.DS_LOAD_UPSAMPLE2).cpu()
    want = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), padding=1)
    assert rel_l2(got, want) < 3e-7
    out = torch.empty(1, 8, 24, 40, device=dev)
    rc = ops.N.lib().ds_conv2d_h3_up(out.data_ptr(), x.to(dev).data_ptr(), pw.up.data_ptr(), 0, None, None, 0, None, None,
                                     1, 16, 8, 12, 20, 0, None, None, None, None, None)
    assert rc != 0 and b"whole number" in ops.N.lib().ds_last_error()


def test_graph_capture_refuses_allocations(dev):
    """A captured region must not allocate: the graph bakes addresses that torch's allocator would hand to someone
    else after the capture (this once corrupted replays of the channel-conditioned network)."""
    ops = _ops()
    x = torch.randn(4096, device=dev)
    out = torch.empty_like(x)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with ops.Graph() as g:                       # pre-allocated output: fine
            ops.scale(x, 2.0, out=out)
        g.launch()
        side.synchronize()
        assert torch.equal(out, x * 2.0)
        with pytest.raises(RuntimeError, match="allocation"):
            with ops.Graph():
                ops.scale(x, 3.0)                    # allocates its result inside the capture
    torch.cuda.current_stream(dev).wait_stream(side)


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, D, H, W, mode (0 plain, 1 maxpool, 2 upsample), circular
    (2, 3, 5, 4, 8, 8, 0, False),
    (1, 9, 20, 5, 7, 9, 0, True),            # ragged tiles and channel tiles, periodic padding
    (2, 6, 16, 4, 6, 10, 1, False),          # MaxPool3d(2) in the loader
    (1, 5, 17, 6, 8, 4, 2, False),           # nearest x2 upsampling in the loader
    (1, 4, 8, 2, 4, 6, 1, True),
    (1, 4, 8, 4, 4, 8, 2, True),
])
def test_conv3d_direct(dev, case):
    """ds_conv3d_direct against torch's fp64 conv3d with the same pooling / upsampling / padding and epilogue terms."""
    ops = _ops()
    B, Cin, Cout, D, H, W, mode, circ = case
    g = torch.Generator().manual_seed(sum(int(v) * (i + 2) for i, v in enumerate(case)))
    shp = {0: (D, H, W), 1: (2 * D, 2 * H, 2 * W), 2: (D // 2, H // 2, W // 2)}[mode]
    x = torch.randn(B, Cin, *shp, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(Cin * 27)
    bias, shift = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
    r1, r2 = torch.randn(B, Cout, D, H, W, generator=g), torch.randn(B, Cout, D, H, W, generator=g)
    src = x.double()
    if mode == 1:
        src = F.max_pool3d(src, 2)
    elif mode == 2:
        src = F.interpolate(src, scale_factor=2.0, mode="nearest")
    if circ:
        src = F.pad(src, (1, 1, 1, 1, 1, 1), mode="circular")
        want = F.conv3d(src, w.double(), bias.double())
    else:
        want = F.conv3d(src, w.double(), bias.double(), padding=1)
    want = want + shift.double()[:, :, None, None, None] + r1.double() + r2.double()
    got = ops.conv3d(x.to(dev), w.to(dev), bias=bias.to(dev), shift=shift.to(dev), res1=r1.to(dev), res2=r2.to(dev),
                     load_mode=mode, circular=circ).cpu()
    assert got.shape == want.shape and rel_l2(got, want) < 3e-7
    with pytest.raises(ValueError, match="weight must be"):
        ops.conv3d(x.to(dev), w.to(dev)[:, :, :, :, :2])


@pytest.mark.parametrize("case", [
    (2, 16, 24, 4, 8, 32, 0, False),
    (1, 9, 70, 5, 12, 20, 0, True),
    (2, 16, 32, 3, 8, 16, 1, False),         # MaxPool3d(2): depth pairs in the slice copy, (H, W) in the 2-D loader
    (1, 24, 17, 6, 16, 32, 2, False),        # nearest x2: depth in the slice copy, (H, W) by the parity kernel
    (1, 8, 8, 4, 8, 12, 2, True),
    (2, 8, 8, 2, 4, 6, 1, True),
])
def test_conv3d_on_the_matrix_cores(dev, case):
    """ops.conv3d_mfma (three 2-D fp16x3 launches over a slice-major, depth-padded copy of the volume) against torch's
    fp64 conv3d and against the direct fp32 kernel."""
    ops = _ops()
    B, Cin, Cout, D, H, W, mode, circ = case
    g = torch.Generator().manual_seed(sum(int(v) * (i + 5) for i, v in enumerate(case)))
    shp = {0: (D, H, W), 1: (2 * D, 2 * H, 2 * W), 2: (D // 2, H // 2, W // 2)}[mode]
    x = torch.randn(B, Cin, *shp, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(Cin * 27)
    bias, shift = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
    r1, r2 = torch.randn(B, Cout, D, H, W, generator=g), torch.randn(B, Cout, D, H, W, generator=g)
    src = x.double()
    if mode == 1:
        src = F.max_pool3d(src, 2)
    elif mode == 2:
        src = F.interpolate(src, scale_factor=2.0, mode="nearest")
    if circ:
        want = F.conv3d(F.pad(src, (1, 1, 1, 1, 1, 1), mode="circular"), w.double(), bias.double())
    else:
        want = F.conv3d(src, w.double(), bias.double(), padding=1)
    want = want + shift.double()[:, :, None, None, None] + r1.double() + r2.double()
    kw = dict(bias=bias.to(dev), shift=shift.to(dev), res1=r1.to(dev), res2=r2.to(dev), load_mode=mode, circular=circ)
    got = ops.conv3d_mfma(x.to(dev), ops.pack_conv3d(w.to(dev), upsampled=mode == 2), **kw).cpu()
    assert got.shape == want.shape and rel_l2(got, want) < 5e-7
    direct = ops.conv3d(x.to(dev), w.to(dev), **kw).cpu()
    assert rel_l2(got, direct.double()) < 5e-7
    one = ops.conv3d_mfma(x.to(dev), ops.pack_conv3d(w.to(dev)), bias=bias.to(dev), shift=shift[:1].to(dev), load_mode=mode,
                          circular=circ).cpu()                      # one shift row shared by the batch
    assert rel_l2(one, want - r1.double() - r2.double() + (shift[:1] - shift).double()[:, :, None, None, None]) < 5e-7


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, H, W, k, mode (0 plain, 1 max-pool, 2 nearest-up)
    (2, 24, 40, 20, 36, 5, 0),
    (1, 64, 64, 32, 32, 5, 1),
    (2, 16, 72, 16, 48, 5, 2),
    (1, 8, 8, 11, 13, 7, 0),           # taps reach past a tiny, odd image on every side
    (2, 32, 16, 16, 32, 7, 1),
    (1, 40, 24, 24, 16, 7, 2),
])
def test_conv_kernels_larger_than_3x3(dev, case):
    """kernel_size / in_out_kernel_size / transition_kernel_size of 5 and 7 (punetg_config.py:19-25): a k x k 'same'
    convolution as ceil(k/3)^2 shifted 3x3 fp16x3 convolutions over zero-padded blocks of the taps (DS_TAP_OFFSET),
    accumulated in place -- fp32-accurate against fp64, with the pooling / upsampling loaders and every epilogue term."""
    ops = _ops()
    B, Cin, Cout, H, W, k, mode = case
    g = torch.Generator().manual_seed(sum(case))
    Hin, Win = (2 * H, 2 * W) if mode == 1 else ((H // 2, W // 2) if mode == 2 else (H, W))
    x = torch.randn(B, Cin, Hin, Win, generator=g) * 2.0
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    bias, shift = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
    r1, r2 = torch.randn(B, Cout, H, W, generator=g), torch.randn(B, Cout, H, W, generator=g)
    src = F.max_pool2d(x, 2) if mode == 1 else (F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == 2 else x)
    want = (F.conv2d(src.double(), w.double(), bias.double(), padding="same") + shift.double()[:, :, None, None]
            + r1.double() + r2.double())
    ref32 = F.conv2d(src, w, bias, padding="same") + shift[:, :, None, None] + r1 + r2
    pw = ops.pack_conv(w.to(dev), "fp16x3")
    assert pw.ks == k and len(pw.subs) == ((k + 2) // 3) ** 2
    stats = torch.zeros(B, Cout, ops.conv_tile_count(H, W), 4, device=dev)
    got = ops.conv(x.to(dev), pw, bias=bias.to(dev), shift=shift.to(dev), res1=r1.to(dev), res2=r2.to(dev), load_mode=mode,
                   tile_stats=stats).cpu()
    assert rel_l2(got, want) <= max(3 * rel_l2(ref32, want), 3e-7)
    assert (got.double() - want).abs().max().item() <= max(4 * (ref32.double() - want).abs().max().item(), 1e-5)
    # the statistics describe the final sum (written by the last block's launch)
    s = stats.cpu().double()
    n, mean = s[..., 3].sum(-1), (s[..., 0] * s[..., 3] + s[..., 1]).sum(-1) / s[..., 3].sum(-1)
    assert torch.equal(n, torch.full_like(n, H * W))
    torch.testing.assert_close(mean, want.mean(dim=(2, 3)), rtol=1e-5, atol=1e-5)
    with pytest.raises(NotImplementedError, match="fp16x3 convolution only"):
        ops.pack_conv(w.to(dev), "bf16x6")


@pytest.mark.parametrize("B,E,L", [(2, 256, 1024), (1, 64, 2048), (3, 32, 96), (2, 128, 160)])
def test_attention_presplit_images_match_in_kernel_staging(dev, B, E, L):
    """ds_attention_h3_ws (K / V split once per sample into LDS-layout images, tiles staged by LDS-DMA, S(k+1) issued
    beside softmax(k)) computes exactly what ds_attention_h3 (every workgroup stages and splits its own tiles) does."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + E + L)
    qkv = (torch.randn(B, 3 * E, L, generator=g) * 1.5).to(dev)
    old = ops.ATTN_IMAGES_MIN_L, ops.ATTN_IMAGES_MIN_L_WIDE
    try:
        ops.ATTN_IMAGES_MIN_L = ops.ATTN_IMAGES_MIN_L_WIDE = 1 << 30
        assert ops.attention_workspace_floats(B, E, L) == 0
        a = ops.attention(qkv, E, precision="fp16x3")
        ops.ATTN_IMAGES_MIN_L = ops.ATTN_IMAGES_MIN_L_WIDE = 0
        assert ops.attention_workspace_floats(B, E, L) == 2 * B * E * L
        b = ops.attention(qkv, E, precision="fp16x3")
        with pytest.raises(ValueError, match="workspace too small"):
            ops.attention(qkv, E, precision="fp16x3", workspace=torch.empty(16, device=dev))
    finally:
        ops.ATTN_IMAGES_MIN_L, ops.ATTN_IMAGES_MIN_L_WIDE = old
    assert torch.equal(a, b)
