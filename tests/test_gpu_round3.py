"""Round 3 on a real MI355X: the fp16x3 kernels over the whole fp32 range of magnitudes.

fp16 has five exponent bits; the reference's fp32 convolutions take raw user fields (punetg.py:719-735) and c_in = 1
parameterisations (preconditioners.py:139-161) at any magnitude.  Every launch whose input is not normalised by construction
takes a per-sample activation exponent (include/diffsci_hip.h: in_amax / out_amax); these tests scale WHOLE inputs by 2^-k
(no O(1) bias, shift or residual to hide behind) and hold the results to the same fp64 bounds as unit-scale data."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import punetg_ref  # noqa: E402
from tests.golden_util import rel_l2  # noqa: E402

REL = 1e-5
SHIFTS = [0, 8, 16, 24, 40, -20, -60]          # the whole input times 2^-k: 4e-3 .. 9e-13, and 1e6, 1e18


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from diffsci_amd import ops
    return ops


@pytest.fixture(scope="module")
def M():
    import diffsci_amd.models as M
    return M


def _bits_of_max(t):
    """float bits of the per-sample max |t| (what an out_amax row must hold)."""
    return t.reshape(t.shape[0], -1).abs().amax(dim=1).contiguous().view(torch.int32)


CONV_CASES = [
    # B, Cin, Cout, H, W, ks, mode (0 plain, 1 max-pool, 2 nearest-up, 3 avg-pool [1x1])
    (2, 32, 64, 16, 32, 3, 0),
    (1, 64, 64, 16, 16, 3, 1),       # DownSampler
    (2, 64, 32, 16, 64, 3, 2),       # UpSampler on the parity kernels (whole 8 x 32 tiles at low resolution)
    (1, 40, 24, 18, 26, 3, 2),       # UpSampler on the gather loader (ragged)
    (1, 1, 8, 16, 32, 3, 0),         # input layer
    (2, 19, 70, 9, 13, 3, 0),        # nothing divides anything
    (2, 32, 96, 8, 8, 1, 0),         # attention in-projection
    (2, 40, 72, 12, 20, 1, 3),       # ADM convresidual(AvgPool2d(2)(x))
    (2, 48, 64, 16, 32, 1, 2),       # ADM convresidual(nearest x2 (x))
]


@pytest.mark.parametrize("k", SHIFTS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_fp16x3_convolutions_on_scaled_inputs(dev, ops, case, k):
    B, Cin, Cout, H, W, ks, mode = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 7)
    Hin, Win = (2 * H, 2 * W) if mode in (1, 3) else ((H // 2, W // 2) if mode == 2 else (H, W))
    x = torch.randn(B, Cin, Hin, Win, generator=g) * 2.0 ** -k
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    pool = {1: lambda t: F.max_pool2d(t, 2), 3: lambda t: F.avg_pool2d(t, 2), 2: lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")}
    src = pool[mode](x) if mode else x
    want = F.conv2d(src.double(), w.double(), padding="same")
    ref32 = F.conv2d(src, w, padding="same")
    pw = ops.pack_conv(w.to(dev), "fp16x3", upsampled=(mode == 2 and ks == 3))
    slots = ops.amax_new(B, dev)
    got = ops.conv(x.to(dev), pw, load_mode=mode, out_amax=slots)          # in_amax=None: reduced by ops
    rel, rel32 = rel_l2(got.cpu(), want), rel_l2(ref32, want)
    assert rel <= max(3 * rel32, 3e-7), (rel, rel32)
    err, err32 = (got.cpu().double() - want).abs().max().item(), (ref32.double() - want).abs().max().item()
    assert err <= max(4 * err32, 1e-6 * want.abs().max().item()), (err, err32)
    # the epilogue's record of its own output, for the next raw-input launch
    assert torch.equal(slots, _bits_of_max(got))
    # the producer's row gives the same result as the reduction, bit for bit
    row = ops.absmax_rows(x.to(dev))
    assert torch.equal(row, _bits_of_max(x.to(dev)))
    again = ops.conv(x.to(dev), pw, load_mode=mode, in_amax=row)
    assert torch.equal(again, got)
    if k >= 24:
        # what the exponent is for: without it the same launch is far outside the tolerance
        bare = ops.conv(x.to(dev), pw, load_mode=mode, in_amax=ops.NORMALISED)
        assert rel_l2(bare.cpu(), want) > 1e-3


def test_activation_exponents_are_per_sample(dev, ops):
    """Samples of very different magnitudes in one batch: each result equals the sample run alone, bit for bit."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 32, 16, 32, generator=g)
    x[1] *= 2.0 ** -30
    x[2] *= 2.0 ** 25
    x[3] = 0
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    bias = torch.randn(64, generator=g)
    pw = ops.pack_conv(w.to(dev), "fp16x3")
    full = ops.conv(x.to(dev), pw, bias=bias.to(dev))
    for b in range(4):
        one = ops.conv(x[b:b + 1].to(dev), pw, bias=bias.to(dev))
        assert torch.equal(full[b:b + 1], one), b
    want = F.conv2d(x.double(), w.double(), bias.double(), padding="same")
    for b in range(3):
        assert rel_l2(full[b].cpu(), want[b]) < 3e-7, b
    assert torch.equal(full[3].cpu(), bias[:, None, None].expand(64, 16, 32))


@pytest.mark.parametrize("k", [0, 8, 16, 24, 40])
@pytest.mark.parametrize("shape", [(2, 32, 32, 32), (1, 64, 16, 64)])
def test_image_input_routes_on_scaled_inputs(dev, ops, shape, k):
    """The norm-fed launches: the norm resets the scale, so the image-input convolution and the parity kernels see the same
    activation whatever the magnitude of x (up to eps inside the variance, which the fp64 reference shares)."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5 + k)
    x = (torch.randn(B, C, H, W, generator=g) + 0.3) * 2.0 ** -k
    gw, gb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
    a64 = F.silu(F.group_norm(x.double(), C, gw.double(), gb.double(), eps=1e-5))
    a32 = F.silu(F.group_norm(x, C, gw, gb, eps=1e-5))
    img = ops.inorm_silu_images(x.to(dev), gw.to(dev), gb.to(dev), 0, eps=1e-5)
    pw = ops.pack_conv(w.to(dev), "fp16x3", upsampled=True)
    got = ops.conv_img(img, pw, B, C, H, W).cpu()
    want, ref32 = F.conv2d(a64, w.double(), padding="same"), F.conv2d(a32, w, padding="same")
    assert rel_l2(got, want) <= max(3 * rel_l2(ref32, want), 1e-6)
    if ops.conv_up_img_supported(pw, H, W):
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")          # noqa: E731
        got = ops.conv_up_img(img, pw, B, C, H, W).cpu()
        want, ref32 = F.conv2d(up(a64), w.double(), padding="same"), F.conv2d(up(a32), w, padding="same")
        assert rel_l2(got, want) <= max(3 * rel_l2(ref32, want), 1e-6)


@pytest.mark.parametrize("k", [0, 10, 24, 40, -12, -30])
@pytest.mark.parametrize("E,L,B", [(64, 256, 2), (256, 1024, 1), (128, 2048, 1), (32, 96, 3)])
def test_attention_on_scaled_operands(dev, ops, E, L, B, k):
    """softmax(q k^T / sqrt(E)) v with q, k, v times 2^-k: the logits shrink (a near-uniform softmax) or blow up (a one-hot one);
    the output carries v's scale either way."""
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, 3 * E, L, generator=g) * 2.0 ** -k
    if k < 0:
        qkv[:, :2 * E] *= 2.0 ** k                        # keep the logits where they are: only v carries the factor
    q, kk, v = (qkv[:, i * E:(i + 1) * E].double().transpose(1, 2) for i in range(3))
    want = (torch.softmax(q @ kk.transpose(1, 2) / math.sqrt(E), dim=-1) @ v).transpose(1, 2)
    q, kk, v = (t.float() for t in (q, kk, v))
    ref32 = (torch.softmax(q @ kk.transpose(1, 2) / math.sqrt(E), dim=-1) @ v).transpose(1, 2)
    slots = ops.amax_new(B, dev)
    got = ops.attention(qkv.to(dev), E, precision="fp16x3", out_amax=slots)
    assert rel_l2(got.cpu(), want) <= max(4 * rel_l2(ref32, want), 1e-6)
    assert torch.equal(slots, _bits_of_max(got))


@pytest.mark.parametrize("mag", [3e2, 5e5])
def test_attention_with_huge_logits(dev, ops, mag):
    """q, k of magnitude 5e5 (an untrained network's residual stream): logits ~1e12 with an fp32 ulp of 65536.  The reference's
    softmax subtracts the row maximum first and stays finite (a one-hot, up to ties inside an ulp); so must the kernel."""
    E, L, B = 32, 64, 2
    g = torch.Generator().manual_seed(17)
    qkv = torch.randn(B, 3 * E, L, generator=g)
    qkv[:, :2 * E] *= mag
    got = ops.attention(qkv.to(dev), E, precision="fp16x3").cpu()
    assert torch.isfinite(got).all()
    q, kk, v = (qkv[:, i * E:(i + 1) * E].double().transpose(1, 2) for i in range(3))
    want = (torch.softmax(q @ kk.transpose(1, 2) / math.sqrt(E), dim=-1) @ v).transpose(1, 2)
    # rows whose two largest logits are closer than fp32 can tell apart may pick either key: compare the others
    S = q @ kk.transpose(1, 2) / math.sqrt(E)
    top2 = S.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-5 * top2[..., 0].abs() + 20.0
    assert clear.float().mean() > 0.9
    err = (got.double() - want).abs().transpose(1, 2)[clear]
    assert err.max() < 1e-4


def _zero_bias_sd(cfg, seed):
    """Reference-initialised weights with the convolution biases removed: the magnitude of the input is then the magnitude of
    the first activations (with biases a tiny input drowns in them and any arithmetic passes)."""
    sd = punetg_ref.random_state_dict(cfg, seed=seed)
    for k in sd:
        if k.endswith(".bias") and (k.startswith("convin") or "conv1.bias" in k or "conv2.bias" in k or "samplers" in k):
            sd[k] = torch.zeros_like(sd[k])
    return sd


@pytest.mark.parametrize("k", [0, 14, 27, -14])
def test_network_on_small_and_large_inputs(M, dev, k):
    """PUNetG on inputs of magnitude 2^-k (6e-5, 7e-9; 1.6e4), biases zeroed: against the CPU oracle in fp32 and in fp64."""
    cfg = punetg_ref.default_config(model_channels=32)
    sd = _zero_bias_sd(cfg, 3)
    net = M.PUNetG(M.PUNetGConfig(model_channels=32))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 1, 32, 32, generator=g) * 2.0 ** -k
    t = torch.tensor([0.4, -1.1])
    got = net(x.to(dev), t.to(dev)).cpu()
    with torch.inference_mode():
        ref = punetg_ref.punetg_forward(sd, cfg, x, t)
        ref64 = punetg_ref.punetg_forward({n: w.double() for n, w in sd.items()}, cfg, x.double(), t.double())
    assert rel_l2(got, ref) < REL
    assert rel_l2(got, ref64) <= max(4 * rel_l2(ref, ref64), 2e-6)


def test_conditional_network_with_a_tiny_field(M, dev):
    """PUNetGCond (punetg.py:719-735) with x at unit scale and a channel field of magnitude 1e-8.
    (a) reference-initialised weights: the field's contribution is 1e-8 of the output; one exponent per sample serves.
    (b) the adversarial checkpoint: the field's input weights are 1e8 times larger, so the field matters as much as x.  One
        exponent per sample cannot serve both channels: the input layer's channel reduction raises the flag, the guard moves that
        layer to the exact-fp32 kernel (a warning, once) and the result matches the oracle."""
    import warnings
    cfg = punetg_ref.default_config(model_channels=16, input_channels=2)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 1, 32, 32, generator=g)
    field = torch.randn(2, 1, 32, 32, generator=g) * 1e-8
    t = torch.tensor([0.2, 0.9])
    for adversarial in (False, True):
        sd = punetg_ref.random_state_dict(cfg, seed=4)
        if adversarial:
            sd["convin.weight"][:, 1] *= 1e8
        net = M.nets.PUNetGCond(M.PUNetGConfig(model_channels=16, input_channels=2, output_channels=1), channel_conditional_items=["f"])
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).eval()
        with torch.inference_mode():
            want = punetg_ref.punetg_forward(sd, cfg, torch.cat([x, field], dim=1), t)
            base = punetg_ref.punetg_forward(sd, cfg, torch.cat([x, torch.zeros_like(field)], dim=1), t)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            got = net(x.to(dev), t.to(dev), {"f": field.to(dev)}).cpu()
        assert rel_l2(got, want) < REL
        if adversarial:
            assert rel_l2(want, base) > 0.05                                  # the field matters ...
            assert net.exact_input_layer and any("exact-fp32" in str(r.message) for r in rec)
            module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True)
            wn = torch.randn(2, 1, 32, 32, generator=g)
            eager_or_graph = []
            for use_graph in (False, True):
                module.use_graph = use_graph
                eager_or_graph.append(module.propagate_white_noise(wn.to(dev), y={"f": field.to(dev)}, nsteps=3).cpu())
            assert torch.equal(*eager_or_graph)
        else:
            assert not net.exact_input_layer and not rec


def test_sampler_raises_the_input_flag_inside_a_captured_run(M, dev):
    """The same disparity met first inside a captured sampling run: the run is repeated once with the exact input layer."""
    import warnings
    from oracle import karras_ref as K
    cfg = punetg_ref.default_config(model_channels=8, input_channels=2)
    sd = punetg_ref.random_state_dict(cfg, seed=6)
    sd["convin.weight"][:, 1] *= 1e8
    net = M.nets.PUNetGCond(M.PUNetGConfig(model_channels=8, input_channels=2, output_channels=1), channel_conditional_items=["f"])
    net.load_state_dict(sd, strict=True)
    module = M.KarrasModule(net.to(dev).eval(), M.KarrasModuleConfig.from_edm(), conditional=True)
    g = torch.Generator().manual_seed(2)
    wn = torch.randn(2, 1, 16, 16, generator=g)
    field = torch.randn(1, 1, 16, 16, generator=g) * 1e-8
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = module.propagate_white_noise(wn.to(dev), y={"f": field.to(dev)}, nsteps=4).cpu()
    assert net.exact_input_layer and sum("exact-fp32" in str(r.message) for r in rec) == 1
    ref = punetg_ref.make_net(sd, cfg)
    want = K.propagate_white_noise(lambda xx, tt, y=None: ref(torch.cat([xx, field.expand(xx.shape[0], -1, -1, -1)], dim=1), tt),
                                   wn, 4)
    assert rel_l2(got, want) < 5e-5


@pytest.mark.parametrize("std", [1e-4, 1e-8])
def test_null_preconditioner_with_small_data(M, dev, std):
    """c_in = 1 (NullPreconditioner, preconditioners.py:139-161): the network sees the state itself, here of std 1e-4 / 1e-8
    -- the denoiser of the reference's own toy test protocol, on a UNet with its biases removed."""
    from oracle import karras_ref as K
    cfg = punetg_ref.default_config(model_channels=16)
    sd = _zero_bias_sd(cfg, 9)
    net = M.PUNetG(M.PUNetGConfig(model_channels=16))
    net.load_state_dict(sd, strict=True)
    conf = M.KarrasModuleConfig.from_edm()
    conf.preconditioner = M.karras.NullPreconditioner()
    module = M.KarrasModule(net.to(dev).eval(), conf)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 1, 32, 32, generator=g) * std
    sigma = torch.tensor([0.5, 2.0])
    got = module.get_denoiser(x.to(dev), sigma.to(dev))[0].cpu()
    ref = punetg_ref.make_net(sd, cfg)
    with torch.inference_mode():
        want = K.denoiser(ref, x, sigma, precond=K.null_precond)
    assert rel_l2(got, want) < REL


@pytest.mark.parametrize("launcher", [True, False])
def test_bench_multi_rank_code_on_one_rank(launcher):
    """What one GPU can exercise of bench.py's N > 1 path: init_process_group("nccl") (RCCL), the row shard of the global
    noise, gather_samples (an all_gather_into_tensor on the GPU), the barriers and the all_reduce(MAX) of the time -- once
    under the driver's launcher line (torch.distributed.run, one rank) and once with the in-process rendezvous of --force-dist."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = ["--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "0", "--nsteps", "4", "--batch", "8", "--size", "64",
            "--no-cpu-baseline", "--no-other-precisions", "--no-other-configs"]
    if launcher:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py")] + tail
    else:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["world_size"] == 1 and line["config"]["backend"].startswith("nccl") and line["n_gpus"] == 1 and line["value"] > 0
    mg = line["multi_gpu"]
    assert len(mg["per_rank_ms_per_step"]) == 1 and mg["gather_ms"][0] >= 0 and mg["noise_draw_ms"][0] > 0
    assert "dp1" in line["config"]["parallelism"] and line["config"]["global_batch"] == 8


def test_bench_with_two_ranks_sharing_the_gpu():
    """bench.py under the driver's launcher line with TWO ranks (DIFFSCI_BENCH_SHARE_GPU=1: both on the box's one GPU, gloo for
    the rendezvous / gather / barrier / MAX-reduction since RCCL refuses two ranks on a device): the weak-scaling accounting --
    per-rank batch, global batch, one line from rank 0 -- executed with GPU compute."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nsteps", "4",
           "--batch", "8", "--size", "64", "--no-cpu-baseline", "--no-other-precisions", "--no-other-configs"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", DIFFSCI_BENCH_SHARE_GPU="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                               # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["backend"].startswith("gloo")
    assert line["config"]["global_batch"] == 16
    mg = line["multi_gpu"]                                              # the fields a scaling curve will be read from, one entry per rank
    assert len(mg["per_rank_ms_per_step"]) == 2 and len(mg["gather_ms"]) == 2 and len(mg["noise_draw_ms"]) == 2
    assert all(v > 0 for v in mg["per_rank_ms_per_step"]) and mg["gather_bytes_per_rank"] == 8 * 64 * 64 * 4
    assert max(mg["per_rank_ms_per_step"]) <= line["ms_per_step"] * 1.05
    assert line["scaling"] == "weak" and line["value"] > 0 and "dp2" in line["config"]["parallelism"]
    assert abs(line["value"] - 16 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 0.01 * line["value"]      # whole-job samples over the max time


def test_vp_runs_the_tabulated_captured_loop(M, dev):
    """VP (non-constant scaling s(t), Scheduler.rhs schedulers.py:275-293) on the fused stepper: the step rows carry s, s'/s and the
    multiplier, the kernels divide the state by s on the way into the network, and the run is captured like an EDM one -- same
    launches per step -- against the reference's goldens (Heun, Euler, sigma-churn with the s(t_hat)/s(t) rescaling)."""
    from tests.golden_util import load
    v, _ = load("vpve8")
    _, sd = load("punetg8_forward")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    nodes = {}
    for tag in ("vp", "ve"):
        cfg = M.KarrasModuleConfig.from_vp(M=2) if tag == "vp" else M.KarrasModuleConfig.from_ve()
        sch = cfg.noisescheduler
        orig = sch.create_steps
        sch.create_steps = (lambda n, tag=tag, orig=orig: v[f"{tag}_steps_{n - 1}"].clone() if f"{tag}_steps_{n - 1}" in v else orig(n))
        module = M.KarrasModule(net, cfg)
        wn = v["white_noise"].to(dev)
        ref_err = rel_l2(v[f"{tag}_punetg_heun_N6"], v[f"{tag}_punetg_heun_N6_f64"])
        tol = max(REL, 4 * ref_err)
        outs = []
        for use_graph in (False, True, True):
            module.use_graph = use_graph
            h = module.propagate_white_noise(wn, nsteps=6, record_history=True).cpu()
            assert rel_l2(h[:2], v[f"{tag}_punetg_heun_N6"][:2]) < REL and rel_l2(h, v[f"{tag}_punetg_heun_N6"]) < tol
            outs.append(h)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
        assert len(module._plans.plans) == 1                               # planned and captured, not the step-by-step loop
        nodes[tag] = next(iter(module._plans.plans.values()))[1].nodes
        o = module.propagate_white_noise(wn, nsteps=6, integrator="euler").cpu()
        assert rel_l2(o, v[f"{tag}_punetg_euler_N6"]) < tol
    assert nodes["vp"] <= nodes["ve"] + 1                                   # one extra launch: x / s in front of the first c_in * x


@pytest.mark.parametrize("n", [4096, 1031])
def test_step_kernels_raise_the_nonfinite_word(dev, ops, n):
    """ds_eval_coef.nonfinite: the step kernel ORs 1 into the word when a value of x_out is inf or NaN, and leaves it alone on a
    finite result (the range guard's result check, carried by a run's last step instead of a reduction + host read per run)."""
    from diffsci_amd.models.karras.steptable import EvalRow
    row = EvalRow(t=torch.tensor(1.0), sigma=1.0, c_in=1.0, c_out=1.0, c_skip=0.5, c_noise=0.0, sigma_sq=1.0, neg_mult=-1.0, neg_lang=0.0, stochastic=False)
    g = torch.Generator().manual_seed(n)
    x, f1, f2 = (torch.randn(n, generator=g).to(dev) for _ in range(3))
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    for bad_value in (None, float("inf"), float("nan"), -float("inf")):
        for where in (0, n - 1, n // 2):
            fa, fb = f1.clone(), f2.clone()
            if bad_value is not None:
                fa[where] = bad_value
            word.zero_()
            out = ops.euler(x, fa, row.coef(nonfinite=word), 0.1, x_out=torch.empty_like(x))
            assert int(word.item()) == (0 if bad_value is None else 1) and bool(torch.isfinite(out).all()) == (bad_value is None)
            word.zero_()
            if bad_value is not None:
                fa, fb[where] = f1.clone(), bad_value                      # the corrector's input this time
            out = ops.heun(x, fa, row.coef(), fb, row.coef(nonfinite=word), 0.1, x_out=torch.empty_like(x))
            assert int(word.item()) == (0 if bad_value is None else 1) and bool(torch.isfinite(out).all()) == (bad_value is None)
            word.zero_()
            ops.heun(x, fa, row.coef(nonfinite=word), fb, row.coef(), 0.1, x_out=torch.empty_like(x))      # k1's word is not the result's
            assert int(word.item()) == 0


def test_range_guard_reads_the_result_word_of_a_captured_run(M, dev):
    """Run level: a finite run leaves both guard words down and costs one host read; a result that overflows from finite inputs
    (an output layer scaled to 1e38) raises the word inside the captured graph and switches the network once, with the warning;
    a non-finite START is the caller's and switches nothing."""
    import warnings
    from diffsci_amd.models.nets import precision
    from tests.golden_util import load
    _, sd = load("punetg8_forward")

    def module_of(scale_out=1.0):
        net = M.PUNetG(M.PUNetGConfig(model_channels=8))
        net.load_state_dict(sd)
        with torch.no_grad():
            net.convout.weight.mul_(scale_out)
        return M.KarrasModule(net.to(dev).eval(), M.KarrasModuleConfig.from_edm())
    g = torch.Generator().manual_seed(11)
    wn = torch.randn(2, 1, 32, 32, generator=g).to(dev)
    module = module_of()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a = module.propagate_white_noise(wn, nsteps=4)
        b = module.propagate_white_noise(wn, nsteps=4)                     # the replay
    assert torch.equal(a, b) and torch.isfinite(a).all() and len(module._plans.plans) == 1
    assert precision.guard_words(module.model, dev).tolist() == [0, 0] and module.model.conv_precision == "fp16x3"
    bad = wn.clone()
    bad[1, 0, 3, 4] = float("nan")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        c = module.propagate_white_noise(bad, nsteps=4)
    assert not torch.isfinite(c[1]).all() and module.model.conv_precision == "fp16x3"
    assert precision.guard_words(module.model, dev).tolist() == [0, 0]     # read and cleared
    over = module_of(1e38)
    with pytest.warns(RuntimeWarning, match="fp16x3 convolution range"):
        d = over.propagate_white_noise(wn, nsteps=4)
    assert over.model.conv_precision == "bf16x6" and not torch.isfinite(d).all()


def test_opt_in_capture_of_a_user_torch_network(M, dev):
    """KarrasModule.capture_eager: a network evaluated as given (here a plain torch module with a dict condition, under
    classifier-free guidance) captured by torch.cuda.CUDAGraph around the HIP step kernels.  The condition the captured calls
    read is a plan-owned copy: replays follow new condition VALUES and new starts, bit-identical to the step-by-step run."""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv2d(1, 8, 3, padding=1)
            self.b = torch.nn.Conv2d(8, 1, 3, padding=1)
            self.e = torch.nn.Linear(2, 8)

        def forward(self, x, t, y=None):
            h = self.a(x) + t.view(-1, 1, 1, 1)
            if y is not None:
                h = h + self.e(y["label"]).view(-1, 8, 1, 1)
            return self.b(torch.nn.functional.silu(h))
    torch.manual_seed(5)
    module = M.KarrasModule(Net().to(dev).eval(), M.KarrasModuleConfig.from_edm(), conditional=True)
    g = torch.Generator().manual_seed(6)
    wn = torch.randn(3, 1, 16, 16, generator=g).to(dev)
    y1, y2 = ({"label": torch.randn(2, generator=g).to(dev)} for _ in range(2))
    with torch.inference_mode():
        want1 = module.propagate_white_noise(wn, y=y1, guidance=2.0, nsteps=5)
        want2 = module.propagate_white_noise(wn * 0.7, y=y2, guidance=2.0, nsteps=5)
        assert len(module._plans.plans) == 0
        module.capture_eager = True
        got1 = module.propagate_white_noise(wn, y=y1, guidance=2.0, nsteps=5)
        got2 = module.propagate_white_noise(wn * 0.7, y=y2, guidance=2.0, nsteps=5)          # a replay: new start, new condition
        got1b = module.propagate_white_noise(wn, y=y1, guidance=2.0, nsteps=5)
    assert len(module._plans.plans) == 1
    assert torch.equal(got1, want1) and torch.equal(got2, want2) and torch.equal(got1b, want1) and not torch.equal(want1, want2)


def test_opt_in_capture_of_a_user_torch_network_in_the_flow_sampler(M, dev):
    """The same switch on SIModule (flow matching): a plain torch network evaluated as given, captured on request."""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv2d(1, 6, 3, padding=1)
            self.b = torch.nn.Conv2d(6, 1, 3, padding=1)

        def forward(self, x, t, y=None):
            return self.b(torch.tanh(self.a(x) + t.view(-1, 1, 1, 1)))
    torch.manual_seed(8)
    mod = M.SIModule(M.SIModuleConfig(scheduler="linear"), Net()).to(dev).eval()
    noise = torch.randn(2, 1, 16, 16, generator=torch.Generator().manual_seed(9)).to(dev)
    with torch.inference_mode():
        want = mod.sample(2, [1, 16, 16], nsteps=5, orig_noise=noise)
        want2 = mod.sample(2, [1, 16, 16], nsteps=5, orig_noise=noise * 0.5)
        assert len(mod._plans.plans) == 0
        mod.capture_eager = True
        got = mod.sample(2, [1, 16, 16], nsteps=5, orig_noise=noise)
        got2 = mod.sample(2, [1, 16, 16], nsteps=5, orig_noise=noise * 0.5)
    assert len(mod._plans.plans) == 1 and torch.equal(got, want) and torch.equal(got2, want2) and not torch.equal(want, want2)


@pytest.mark.parametrize("shape", [(3, 4, 16, 16), (2, 1, 7, 9), (5, 3, 33), (2, 64, 8, 8), (1, 2, 1024, 600)])
def test_input_maxima_one_launch_and_fallback_agree(dev, ops, shape):
    """ds_input_amax (one launch: zero the arena, reduce, channel criterion) against the three-launch route it replaces
    (fill + ds_absmax_channels) and against torch: same rows, same flag -- on aligned, odd-sized and beyond-the-limit inputs
    (the last shape exceeds INPUT_AMAX_MAX_FLOATS per sample, which the arena routes to the fallback)."""
    from diffsci_amd.models.nets.punetg import _AmaxArena, _Workspace
    g = torch.Generator().manual_seed(sum(shape))
    B, C = shape[:2]
    x = torch.randn(*shape, generator=g).to(dev)
    x[0] *= 2.0 ** -20
    want = _bits_of_max(x)
    for case in ("benign", "tiny channel, big weights", "no weights"):
        xx = x.clone()
        wmax = None if case == "no weights" else torch.rand(C, generator=g).to(dev) + 0.5
        expect_flag = 0
        if C > 1 and case != "benign":
            xx[:, 0] *= 2.0 ** -30                               # a channel thirty binades below the others
            if wmax is not None:
                wmax[0] = 2.0 ** 30                              # ... whose weights make it matter
            expect_flag = 1
        want = _bits_of_max(xx)
        ws = _Workspace()
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        arena = _AmaxArena(ws, B, dev, zero=False)
        arena.i32.fill_(123)                                     # stale slots: the launch has to zero them
        row = arena.of_input(xx, flags[0:1], wmax)
        one_launch = xx[0].numel() <= ops.INPUT_AMAX_MAX_FLOATS and C <= 64
        assert torch.equal(row, want) and int(flags[0]) == expect_flag and int(flags[1]) == 0, (case, shape)
        used = 1 if one_launch else 1 + C
        assert arena.n == used and int(arena.i32[used:].abs().max()) == 0
        # the route the one-launch kernel replaced, on a fresh arena
        flags2 = torch.zeros(2, dtype=torch.int32, device=dev)
        out, scratch = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B * C, dtype=torch.int32, device=dev)
        ops.absmax_channels(xx, out, scratch, flags2[0:1], wmax)
        assert torch.equal(out, want) and int(flags2[0]) == expect_flag


def _two_rank_worker(rank, world, port, nsamples, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import diffsci_amd.models as M
        from diffsci_amd.parallel import sample_sharded
        from tests.golden_util import load
        dev = torch.device("cuda:0")                                  # both ranks share the box's one GPU
        _, sd = load("punetg8_forward")
        net = M.PUNetG(M.PUNetGConfig(model_channels=8))
        net.load_state_dict(sd)
        module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
        out = {}
        for integrator in ("heun", "karras"):
            torch.manual_seed(11)                                     # every rank holds the same generator state
            out[integrator] = sample_sharded(module, nsamples, [1, 32, 32], nsteps=4, seed=7, integrator=integrator).cpu()
        q.put((rank, {k: v.numpy().copy() for k, v in out.items()}))      # plain bytes (a tensor's shared-memory handle must be opened while this process lives)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nsamples", [6, 5])
def test_two_ranks_on_one_gpu_reproduce_the_single_process_batch(M, dev, nsamples):
    """The N > 1 path executed with GPU compute as far as a one-GPU box allows: two processes (gloo rendezvous and gather -- RCCL
    refuses two ranks on one device) shard the global noise, run their rows through the captured sampler on the same card and
    gather; every rank must hold the single-process batch bit for bit -- for the deterministic sampler and for the sigma-churn one,
    whose in-kernel noise is addressed by GLOBAL element index (even and ragged splits)."""
    import socket
    import torch.multiprocessing as mp
    from diffsci_amd.parallel import global_white_noise
    from tests.golden_util import load
    _, sd = load("punetg8_forward")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    net.load_state_dict(sd)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
    wn = global_white_noise(nsamples, [1, 32, 32], 7).to(dev)
    want = {}
    for integrator in ("heun", "karras"):
        torch.manual_seed(11)
        want[integrator] = module.propagate_white_noise(wn, nsteps=4, integrator=integrator).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, nsamples, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in got) == [0, 1]
    for _, out in got:
        for integrator in ("heun", "karras"):
            assert torch.equal(torch.from_numpy(out[integrator]), want[integrator]), integrator
    assert not torch.equal(want["heun"], want["karras"])


def test_plan_key_follows_the_step_coefficients(M, dev):
    """The plan key carries every scalar the captured kernels take as arguments (StepTable.digest): a preconditioner changed IN
    PLACE between two runs -- same grid, same network -- re-captures instead of replaying the old c_in / c_out / c_skip."""
    from tests.golden_util import load
    _, sd = load("punetg8_forward")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    net.load_state_dict(sd)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
    wn = torch.randn(2, 1, 32, 32, generator=torch.Generator().manual_seed(3)).to(dev)
    a = module.propagate_white_noise(wn, nsteps=4)
    assert len(module._plans.plans) == 1
    module.config.preconditioner.sigma_data.fill_(0.8)
    b = module.propagate_white_noise(wn, nsteps=4)
    module.use_graph = False
    want = module.propagate_white_noise(wn, nsteps=4)
    assert len(module._plans.plans) == 2 and torch.equal(b, want) and not torch.equal(a, b)
