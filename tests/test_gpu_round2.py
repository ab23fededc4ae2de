"""Round-2 behaviours on a real MI355X: in-kernel Philox noise, condition refresh under a captured plan, several
channel-condition items, per-sample conditions, odd-sized states (unaligned slices), a 1-rank RCCL group."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import karras_ref as K  # noqa: E402
from oracle import mlp_ref, philox_ref, punetg_ref  # noqa: E402
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


@pytest.fixture(scope="module")
def grids():
    v, _ = load("schedule")
    return v


def _pin_grid(module, grids):
    sch = getattr(getattr(module, "config", None), "noisescheduler", module)
    orig = sch.create_steps
    sch.create_steps = lambda n: grids[f"steps_{n - 1}"].clone() if f"steps_{n - 1}" in grids else orig(n)
    return sch


def _state(dev, seed, base):
    i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v        # noqa: E731
    return torch.tensor([i64(seed), i64(base)], dtype=torch.int64, device=dev)


# ----------------------------------------------------------------------------------------- Philox
@pytest.mark.parametrize("seed,base,off,n", [(0, 0, 0, 1), (1234, 0, 0, 4099), (2**63 + 17, 2**40 + 3, 5, 1027),
                                             (42, 2**32 - 2, 0, 64)])
def test_philox_stream_matches_oracle(dev, seed, base, off, n):
    """ds_philox_normal against the numpy restatement (oracle/philox_ref.py, pinned by the Random123 known answers
    in test_oracle_golden.py): same counters, same key, same uniform -> normal map; the transcendental functions
    differ in the last ulp, hence a tolerance of a few fp32 ulp.  The stream does not depend on pointer alignment."""
    from diffsci_amd import ops
    st = _state(dev, seed, base)
    got = ops.philox_normal(st, off, (n,)).cpu().double().numpy()
    want = philox_ref.normal(seed, base + off, n)
    assert np.abs(got - want).max() <= 4e-6 * (1.0 + np.abs(want).max())
    buf = torch.empty(n + 1, device=dev)
    shifted = ops.philox_normal(st, off, (n,), out=buf[1:]).cpu()       # 4-byte aligned only: the scalar path
    assert torch.equal(shifted, torch.from_numpy(got).float())


def test_philox_moments_and_distribution(dev):
    from scipy import stats
    from diffsci_amd import ops
    n = 1 << 22
    z = ops.philox_normal(_state(dev, 7, 0), 0, (n,)).cpu().double().numpy()
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.std() - 1) < 4 / np.sqrt(2 * n)
    assert abs(stats.skew(z)) < 0.01 and abs(stats.kurtosis(z)) < 0.02
    assert stats.kstest(z[:1 << 18], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 4 / np.sqrt(n)      # Box-Muller pairs are uncorrelated
    z2 = ops.philox_normal(_state(dev, 8, 0), 0, (n,)).cpu().double().numpy()
    assert abs(np.corrcoef(z, z2)[0, 1]) < 4 / np.sqrt(n)               # another key: another stream


@pytest.mark.parametrize("n", [3, 4096, 2 * 33 * 35])
def test_churn_and_euler_maruyama_generate_their_noise(dev, n):
    """eps = NULL + a Philox state: the injection kernels compute x + coef*eps with eps from the stream above,
    bit for bit what they compute from an injected copy of that stream."""
    from diffsci_amd import ops
    from diffsci_amd._native import EvalCoef
    g = torch.Generator().manual_seed(n)
    x, f = torch.randn(n, generator=g).to(dev) * 40, torch.randn(n, generator=g).to(dev)
    st = _state(dev, 99, 1000)
    eps = ops.philox_normal(st, 12, (n,))
    a = ops.churn(x, eps, 3.25, xhat_out=torch.empty_like(x), xin_out=torch.empty_like(x), c_in=0.125)
    xin = torch.empty_like(x)
    b = ops.churn(x, None, 3.25, xhat_out=torch.empty_like(x), xin_out=xin, c_in=0.125, philox=(st, 12))
    assert torch.equal(a, b) and torch.equal(xin, 0.125 * b) and torch.equal(b.cpu(), x.cpu() + 3.25 * eps.cpu())
    k = EvalCoef(c_out=0.4, c_skip=0.1, sigma_sq=2.0, neg_mult=-1.4, neg_lang=-0.7, guidance=1.0, one_minus_guidance=0.0,
                 input_kind=0, stochastic=1)
    a = ops.euler(x, f, k, -0.5, x_out=torch.empty_like(x), eps=eps, noise_coef=1.3, sqrt_abs_dt=0.9)
    b = ops.euler(x, f, k, -0.5, x_out=torch.empty_like(x), philox=(st, 12), noise_coef=1.3, sqrt_abs_dt=0.9)
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="exactly one of eps"):
        ops.churn(x, None, 1.0, xhat_out=torch.empty_like(x))


@pytest.fixture(scope="module")
def net8(M, dev):
    _, sd = load("punetg8_forward")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    net.load_state_dict(sd)
    return net.to(dev).eval()


@pytest.mark.parametrize("integrator", ["karras", "euler-maruyama"])
def test_stochastic_samplers_in_generator_mode(M, net8, dev, grids, integrator):
    """Without injected eps the noise is drawn inside the kernels from torch's CUDA generator state: the same seed
    reproduces the run bit for bit (eager and hipGraph alike), the next run continues the stream, and the trajectory
    equals the CPU oracle's when the oracle is fed the eps the kernels generated."""
    from diffsci_amd import ops
    v, sd = load("punetg8_forward")
    module = M.KarrasModule(net8, M.KarrasModuleConfig.from_edm())
    _pin_grid(module, grids)
    if integrator == "euler-maruyama":
        module.config.noisescheduler.langevin_const = 0.3
    wn = load("punetg8_traj")[0]["white_noise"].to(dev)
    gen = torch.cuda.default_generators[0]
    outs = {}
    for use_graph in (False, True):
        module.use_graph = use_graph
        torch.manual_seed(5)
        seed, off = gen.initial_seed(), gen.get_offset()
        a = module.propagate_white_noise(wn, nsteps=6, record_history=True, integrator=integrator)
        used = gen.get_offset() - off
        assert used == (6 * ops.philox_counters(wn.numel()) + 3) // 4 * 4
        b = module.propagate_white_noise(wn, nsteps=6, record_history=True, integrator=integrator)     # stream continues
        assert not torch.equal(a, b)
        torch.manual_seed(5)
        c = module.propagate_white_noise(wn, nsteps=6, record_history=True, integrator=integrator)
        assert torch.equal(a, c)
        outs[use_graph] = a
    assert torch.equal(outs[False], outs[True])
    st = _state(dev, seed, off)
    per = ops.philox_counters(wn.numel())
    eps = torch.stack([ops.philox_normal(st, i * per, wn.shape) for i in range(6)]).cpu()
    ref = punetg_ref.make_net(sd, punetg_ref.default_config(model_channels=8))
    kw = dict(langevin_const=0.3) if integrator == "euler-maruyama" else {}
    want = K.propagate_white_noise(ref, wn.cpu(), 6, integrator=integrator, record_history=True, eps=eps,
                                   sigma_grid=module.config.noisescheduler.create_steps(7), **kw)
    assert rel_l2(outs[True].cpu(), want) < REL
    with pytest.raises(ValueError, match="injected"):        # a generator-mode plan and an injected run are different plans
        from diffsci_amd.models.karras.engine import Loop, ModuleSource
        from diffsci_amd.models.karras.steptable import build_step_table
        sch = module.config.noisescheduler
        sch.set_temporary_integrator(integrator)
        table = build_step_table(sch, sch.integrator, 6, preconditioner=module.config.preconditioner)
        sch.unset_temporary_integrator()
        Loop(table, ModuleSource(module, None, 1.0, wn.shape[0], wn), wn).set_noise(eps.to(dev))
    module.config.noisescheduler.langevin_const = 1.0


# ----------------------------------------------------------------------------------------- odd sizes
@pytest.mark.parametrize("B", [1, 3, 5])
def test_odd_sized_states_with_history_and_noise(M, dev, grids, B):
    """History / eps slices of a [B, 2] state with odd B start at addresses that are not multiples of 16 bytes; the
    reference accepts any shape (record_history and the stochastic integrators raised DS_ERR_SHAPE)."""
    v, sd = load("mlp_cfg1")
    model = M.MLPUncond(2, [20])
    model.load_state_dict(sd)
    module = M.KarrasModule(model, M.KarrasModuleConfig.from_edm()).to(dev)
    _pin_grid(module, grids)
    ref = mlp_ref.make_net(sd)
    wn = v["white_noise"][:B].contiguous()
    for integ in ("heun", "euler"):
        h = module.propagate_white_noise(wn.to(dev), nsteps=18, record_history=True, integrator=integ).cpu()
        want = K.propagate_white_noise(ref, wn, 18, integrator=integ, record_history=True, sigma_grid=grids["steps_18"])
        assert rel_l2(h, want) < REL
    eps = v["eps_karras_N18"][:, :B].contiguous()
    h = module.propagate_white_noise(wn.to(dev), nsteps=18, record_history=True, integrator="karras", eps=eps.to(dev)).cpu()
    want = K.propagate_white_noise(ref, wn, 18, integrator="karras", record_history=True, eps=eps, sigma_grid=grids["steps_18"])
    assert rel_l2(h, want) < REL
    h = module.propagate_white_noise(wn.to(dev), nsteps=18, record_history=True, integrator="karras").cpu()   # generator mode
    assert torch.isfinite(h).all() and h.shape == want.shape
    # per-sample sigma: x[b] / out[b] views of a 3-float sample
    x = torch.randn(B, 3, generator=torch.Generator().manual_seed(B)).to(dev)

    class Net(torch.nn.Module):
        def forward(self, xx, t):
            return 0.5 * xx + t[:, None]
    mod = M.KarrasModule(Net(), M.KarrasModuleConfig.from_edm())
    sig = torch.linspace(0.5, 3.0, B)
    s = mod.get_score(x, sig).cpu()
    want = K.score(lambda xx, t: 0.5 * xx + t[:, None], x.cpu(), sig)
    torch.testing.assert_close(s, want, rtol=2e-6, atol=1e-6)


# ----------------------------------------------------------------------------------------- conditions under a captured plan
class _WideLabelEmbedding(torch.nn.Module):
    """A user conditional embedding over an 80-entry one-hot label (more than 64 elements: the plan cache of round 1
    keyed such tensors by address + checksum, which cannot tell two one-hot labels apart)."""

    def __init__(self, n, c):
        super().__init__()
        self.lin = torch.nn.Linear(n, c)

    def forward(self, y):
        return self.lin(y.reshape(1, -1))


def test_plan_replays_follow_the_condition_values(M, dev, grids):
    _, sd = load("punetg8_forward")
    torch.manual_seed(3)
    net = M.PUNetG(M.PUNetGConfig(model_channels=8), conditional_embedding=_WideLabelEmbedding(80, 8))
    net.load_state_dict(sd, strict=False)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev).eval()
    _pin_grid(module, grids)
    wn = load("punetg8_traj")[0]["white_noise"].to(dev)

    def onehot(i):                                    # a FRESH tensor per call, as a user would make it
        y = torch.zeros(80)
        y[i] = 1.0
        return y.to(dev)

    want = {}
    module.use_graph = False
    for i in (3, 7):
        want[i] = module.propagate_white_noise(wn, y=onehot(i), guidance=2.0, nsteps=4)
    assert rel_l2(want[3], want[7]) > 1e-3
    module.use_graph = True
    for i in (3, 7, 3, 7):                            # first call captures; the others replay with refreshed tables
        y = onehot(i)
        got = module.propagate_white_noise(wn, y=y, guidance=2.0, nsteps=4)
        del y
        torch.empty(80, device=dev)                   # give the allocator a chance to recycle the label's block
        assert torch.equal(got, want[i]), i
    assert len(module._plans.plans) == 1


def test_several_channel_condition_items(M, dev, grids):
    """PUNetGCond with two channel_conditional_items (punetg.py:719-733): the fields are concatenated into a buffer
    the network owns, so replays of a captured plan see the values of the current call."""
    v, sd = load("punetg8_cond")
    net = M.nets.PUNetGCond(M.PUNetGConfig(model_channels=8, input_channels=3, output_channels=1),
                            channel_conditional_items=["a", "b"])
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True)
    _pin_grid(module, grids)
    wn = v["white_noise"].to(dev)
    field = v["field"][0].to(dev)                                         # [2, H, W]: one channel per item
    assert field.shape[0] == 2

    def y_of(f):
        return {"a": f[0:1].clone(), "b": f[1:2].clone()}

    for use_graph in (False, True, True):
        module.use_graph = use_graph
        h = module.propagate_white_noise(wn, y=y_of(field), nsteps=4, record_history=True).cpu()
        assert rel_l2(h, v["hist_heun_N4_f32"]) < REL                     # cat([a, b]) is the fixture's field
    f2 = torch.stack([field[1] * 0.5, field[0] + 0.25])
    module.use_graph = False
    want = module.propagate_white_noise(wn, y=y_of(f2), nsteps=4)
    module.use_graph = True
    got = module.propagate_white_noise(wn, y=y_of(f2), nsteps=4)           # replay of the plan captured above
    assert torch.equal(got, want) and rel_l2(got.cpu(), v["hist_heun_N4_f32"][-1]) > 1e-3
    again = module.propagate_white_noise(wn, y=y_of(field), nsteps=4).cpu()
    assert rel_l2(again, v["hist_heun_N4_f32"][-1]) < REL
    out = net(v["x"].to(dev), v["t"].to(dev), {"a": v["field"][:, 0:1].to(dev), "b": v["field"][:, 1:2].to(dev)}).cpu()
    assert rel_l2(out, v["out_f32"]) < REL


class _TableEmbedding(torch.nn.Module):
    """conditional_embedding returning one row per sample ([B, C]) for integer labels [.., B]."""

    def __init__(self, n, c):
        super().__init__()
        self.table = torch.nn.Parameter(torch.randn(n, c))

    def forward(self, y):
        return self.table[y.reshape(-1).long()]


def test_per_sample_conditions_in_the_captured_sampler(M, dev, grids):
    """y with one condition per sample (the embedding returns [B, C]): the planned sampler tabulates the time shifts
    per (evaluation, sample).  Row b of the batched run equals a run of sample b alone with its own condition."""
    _, sd = load("punetg8_forward")
    torch.manual_seed(4)
    net = M.PUNetG(M.PUNetGConfig(model_channels=8), conditional_embedding=_TableEmbedding(10, 8))
    net.load_state_dict(sd, strict=False)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev).eval()
    _pin_grid(module, grids)
    wn = load("punetg8_traj")[0]["white_noise"].to(dev)
    B = wn.shape[0]
    labels = torch.arange(B, device=dev) * 3 % 10
    res = {}
    for use_graph in (False, True, True):
        module.use_graph = use_graph
        res[use_graph] = module.propagate_white_noise(wn, y=labels.float(), guidance=1.5, nsteps=4)
    assert torch.equal(res[False], res[True])
    for b in range(B):
        one = module.propagate_white_noise(wn[b:b + 1], y=labels[b:b + 1].float(), guidance=1.5, nsteps=4)
        assert rel_l2(res[True][b:b + 1], one) < 2e-6
    other = module.propagate_white_noise(wn, y=((labels + 1) % 10).float(), guidance=1.5, nsteps=4)   # replay, new values
    assert rel_l2(other, res[True]) > 1e-3


# ----------------------------------------------------------------------------------------- RCCL, one rank
def test_sample_sharded_on_a_one_rank_rccl_group(M, net8, dev, grids):
    """parallel.sample_sharded through a real `nccl` (= RCCL) process group of world size 1: the all-gather path,
    device placement and noise definition of the multi-GPU sampler with the real network."""
    import os
    import torch.distributed as dist
    from diffsci_amd import parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        module = M.KarrasModule(net8, M.KarrasModuleConfig.from_edm())
        _pin_grid(module, grids)
        got = parallel.sample_sharded(module, 5, [1, 32, 32], nsteps=4, seed=11)
        torch.manual_seed(11)
        wn = torch.randn(5, 1, 32, 32)
        want = module.propagate_white_noise(wn.to(dev), nsteps=4)
        assert got.shape == (5, 1, 32, 32) and torch.equal(got, want)
    finally:
        dist.destroy_process_group()


def test_volume_norm_folding(M, dev):
    """Round 2: the norms of a residual block on volumes are folded into the copies and the convolutions around them
    (ops.resblock3d_fused: norm1 in the volume -> slice copy, the intermediate slice-major with norm2 in conv2's loader,
    statistics from the slice -> volume copy).  Pieces against torch on the same tensors, the network against the
    standalone-norm route and the reference's golden."""
    from diffsci_amd import ops
    from tests.golden_util import load, rel_l2
    torch.manual_seed(11)
    B, C, D, H, W = 2, 8, 6, 16, 16
    h = torch.randn(B, C, D, H, W, device=dev) * 1.7 + 0.4
    w1, b1 = torch.randn(C, device=dev), torch.randn(C, device=dev)
    # the copy's statistics -> table -> activated slices
    s_plain = torch.empty(B * (D + 2), C, H, W, device=dev)
    ops.N.check(ops.N.lib().ds_volume_to_slices(ops._p(s_plain), ops._p(h), B, C, D, H * W, 0, 0, 1, ops._stream()), "to_slices")
    back = torch.empty_like(h)
    st = torch.empty(B, C, ops.volume_stat_tiles(D, H * W), 4, device=dev)
    r1 = torch.randn_like(h)
    ops._from_slices(back, s_plain, r1, None, B, C, D, H * W, st)
    assert torch.equal(back, h + r1)
    tab = ops.inorm_table(st, w1, b1, 0, D * H * W)
    v = (h + r1).double()
    mean, var = v.mean(dim=(2, 3, 4)), v.var(dim=(2, 3, 4), unbiased=False)
    assert torch.allclose(tab[:, :C, 0].double(), mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(tab[:, :C, 1].double(), w1.double() / torch.sqrt(var + 1e-5), rtol=1e-5)
    s_act = torch.empty_like(s_plain)
    ops.N.check(ops.N.lib().ds_volume_to_slices_act(ops._p(s_act), ops._p(back), ops._p(tab), B, C, D, H * W, 0, ops._stream()), "to_slices_act")
    want = torch.nn.functional.silu(torch.nn.functional.group_norm(back, C, w1, b1, 1e-5))
    got = s_act.view(B, D + 2, C, H, W)
    assert float(got[:, 0].abs().max()) == 0.0 and float(got[:, D + 1].abs().max()) == 0.0
    assert rel_l2(got[:, 1:D + 1].permute(0, 2, 1, 3, 4).cpu(), want.cpu()) < 2e-6
    # the network: folded against standalone norms and the reference
    g, sd = load("punetg8_3d")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8, dimension=3))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x, t = g["x"].to(dev), g["t"].to(dev)
    calls = []
    orig = ops.resblock3d_fused
    ops.resblock3d_fused = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        folded = net(x, t).cpu()
    finally:
        ops.resblock3d_fused = orig
    assert len(calls) >= 8, "the folded block did not run"
    net.fuse_norm = False
    plain = net(x, t).cpu()
    net.fuse_norm = True
    assert rel_l2(folded, plain) < 3e-6
    assert rel_l2(folded, g["out_f32"]) < 1e-5
    assert rel_l2(folded, g["out_f64"]) < max(4 * rel_l2(g["out_f32"], g["out_f64"]), 2e-6)


def test_volume_norm_folding_with_periodic_padding(M, dev):
    """Round 3 (SURVEY 8f-4 residue): the same folding under convolution_type='circular' (commonlayers.py:918-971 on volumes).
    Pad slices cannot be zero rows there: the activated copy wraps the depth axis itself, the slice-major intermediate gets its
    pad slices from ds_wrap_pad_slices and the per-slice table carries the sample's row on them."""
    from diffsci_amd import ops
    from tests.golden_util import load, rel_l2
    torch.manual_seed(12)
    B, C, D, H, W = 2, 8, 5, 16, 16
    h = torch.randn(B, C, D, H, W, device=dev) * 1.3 - 0.2
    w1, b1 = torch.randn(C, device=dev), torch.randn(C, device=dev)
    tab = torch.zeros(B, ops.table_channels(C), 4, device=dev)
    v = h.double()
    mean, var = v.mean(dim=(2, 3, 4)), v.var(dim=(2, 3, 4), unbiased=False)
    tab[:, :C, 0], tab[:, :C, 1], tab[:, :C, 2] = mean.float(), (w1.double() / torch.sqrt(var + 1e-5)).float(), b1
    s_act = torch.empty(B * (D + 2), C, H, W, device=dev)
    ops.N.check(ops.N.lib().ds_volume_to_slices_act(ops._p(s_act), ops._p(h), ops._p(tab), B, C, D, H * W, 1, ops._stream()), "to_slices_act")
    got = s_act.view(B, D + 2, C, H, W)
    want = torch.nn.functional.silu(torch.nn.functional.group_norm(h, C, w1, b1, 1e-5)).permute(0, 2, 1, 3, 4)
    assert rel_l2(got[:, 1:D + 1].cpu(), want.cpu()) < 2e-6
    assert torch.equal(got[:, 0], got[:, D]) and torch.equal(got[:, D + 1], got[:, 1])          # wrapped, activated
    s = torch.randn(B, D + 2, C, H, W, device=dev)
    keep = s.clone()
    ops.N.check(ops.N.lib().ds_wrap_pad_slices(ops._p(s), B, C, D, H * W, ops._stream()), "wrap")
    assert torch.equal(s[:, 1:D + 1], keep[:, 1:D + 1]) and torch.equal(s[:, 0], keep[:, D]) and torch.equal(s[:, D + 1], keep[:, 1])
    # one block against torch (fp64) on the same tensors
    blk_w = [torch.randn(C, C, 3, 3, 3, device=dev) / (27 * C) ** 0.5 for _ in range(2)]
    blk_b = [torch.randn(C, device=dev) * 0.1 for _ in range(2)]
    w2, b2 = torch.randn(C, device=dev), torch.randn(C, device=dev)
    shift = torch.randn(B, C, device=dev)
    p1, p2 = ops.pack_conv3d(blk_w[0]), ops.pack_conv3d(blk_w[1])
    out = ops.resblock3d_fused(h, tab, p1, blk_b[0], shift, p2, blk_b[1], w2, b2, 0, circular=True)
    F = torch.nn.functional
    pad = lambda t: F.pad(t, (1, 1, 1, 1, 1, 1), mode="circular")      # noqa: E731
    a = F.silu(F.group_norm(h.double(), C, w1.double(), b1.double(), 1e-5))
    y = F.conv3d(pad(a), blk_w[0].double(), blk_b[0].double()) + shift.double()[:, :, None, None, None]
    a2 = F.silu(F.group_norm(y, C, w2.double(), b2.double(), 1e-5))
    ref = F.conv3d(pad(a2), blk_w[1].double(), blk_b[1].double()) + h.double()
    assert rel_l2(out.double().cpu(), ref.cpu()) < 3e-6
    # the network: the folded block runs, and matches the standalone-norm route and the reference's golden
    g, sd = load("punetg8_3d_circular")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8, dimension=3, convolution_type="circular"))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x, t = g["x"].to(dev), g["t"].to(dev)
    calls = []
    orig = ops.resblock3d_fused
    ops.resblock3d_fused = lambda *a, **k: (calls.append(k.get("circular")), orig(*a, **k))[1]
    try:
        folded = net(x, t).cpu()
    finally:
        ops.resblock3d_fused = orig
    assert len(calls) >= 8 and all(calls), "the folded periodic block did not run"
    net.fuse_norm = False
    plain = net(x, t).cpu()
    net.fuse_norm = True
    assert rel_l2(folded, plain) < 3e-6
    assert rel_l2(folded, g["out_f32"]) < 1e-5
    assert rel_l2(folded, g["out_f64"]) < max(4 * rel_l2(g["out_f32"], g["out_f64"]), 2e-6)
    # ... and a captured sampling run on it
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm())
    wn = torch.randn(2, 1, 16, 16, 16, generator=torch.Generator().manual_seed(4)).to(dev)
    runs = []
    for use_graph in (False, True, True):
        module.use_graph = use_graph
        runs.append(module.propagate_white_noise(wn, nsteps=3))
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[1], runs[2]) and len(module._plans.plans) == 1


@pytest.mark.parametrize("integrator", ["karras", "euler-maruyama"])
def test_sharded_stochastic_run_reproduces_the_unsharded_one(M, dev, integrator):
    """parallel.sample_sharded tells the module where its rows sit in the global batch (KarrasModule.noise_shard); the
    in-kernel Philox stream is then addressed by GLOBAL element index, so two shards run one after the other from the same
    generator state are the unsharded run's rows bit for bit (and ranks seeded alike draw disjoint noise)."""
    from tests.golden_util import load
    _, sd = load("punetg8_forward")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    net.load_state_dict(sd, strict=True)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev)
    if integrator == "euler-maruyama":
        module.config.noisescheduler.langevin_const = 0.3
    torch.manual_seed(3)
    wn = torch.randn(4, 1, 32, 32, device=dev)
    per_row = wn[0].numel()
    for use_graph in (False, True):
        module.use_graph = use_graph
        torch.manual_seed(11)
        full = module.propagate_white_noise(wn, nsteps=5, integrator=integrator)
        parts = []
        for lo, hi in ((0, 1), (1, 4)):                          # a ragged split
            torch.manual_seed(11)                                 # every rank holds the same generator state
            module.noise_shard = (lo * per_row, 4 * per_row)
            try:
                parts.append(module.propagate_white_noise(wn[lo:hi], nsteps=5, integrator=integrator))
            finally:
                module.noise_shard = None
        assert torch.equal(torch.cat(parts), full)
        torch.manual_seed(11)
        alone = module.propagate_white_noise(wn[1:4], nsteps=5, integrator=integrator)     # unsharded addressing: other noise
        assert not torch.equal(alone, full[1:4])
    with pytest.raises(ValueError, match="noise_shard"):
        module.noise_shard = (2, 4 * per_row)
        try:
            module.propagate_white_noise(wn[:1], nsteps=5, integrator=integrator)
        finally:
            module.noise_shard = None


def test_no_time_input_with_a_condition(M, dev):
    """punetg.py:396-410: t = None means a zero time embedding, to which the embedded condition is still added."""
    from oracle import punetg_ref
    from tests.golden_util import load, rel_l2
    v, sd = load("punetg8_forward")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    torch.manual_seed(2)
    ye = torch.randn(2, 8)
    cfg = punetg_ref.default_config(model_channels=8)
    with torch.inference_mode():
        want = punetg_ref.punetg_forward(sd, cfg, v["x"], None, ye)
        zero = punetg_ref.punetg_forward(sd, cfg, v["x"], None, None)
    assert rel_l2(net(v["x"].to(dev), None, ye.to(dev)).cpu(), want) < 1e-5
    assert rel_l2(net(v["x"].to(dev)).cpu(), zero) < 1e-5
    assert rel_l2(want, zero) > 1e-3


def test_batched_classifier_free_guidance_is_bit_identical(M, dev):
    """engine.ModuleSource runs the conditional and the unconditional evaluation of a guided step as one evaluation of batch
    2B (tabulated time shifts: rows B.. are the unconditional half).  Every kernel treats samples independently, so the
    result is the two-evaluation result bit for bit -- eagerly, captured, and after a condition change under the plan."""
    from tests.golden_util import load
    v, sd = load("punetg8_porosity")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8, input_channels=4, output_channels=4),
                   conditional_embedding=M.nets.PorosityEmbedder(dembed=8))
    net.load_state_dict(sd, strict=True)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev)
    wn = v["white_noise"].to(dev)
    ys = [{"porosity": v["porosity"].to(dev)}, {"porosity": (v["porosity"] * 0.5 + 0.1).to(dev)}]
    outs = {}
    for batched in (False, True):
        module.batch_cfg = batched
        for use_graph in (False, True):
            module.use_graph = use_graph
            outs[(batched, use_graph)] = [module.propagate_white_noise(wn, y=y, guidance=2.0, nsteps=4, record_history=True).clone()
                                          for y in ys + ys[:1]]
    ref = outs[(False, False)]
    assert not torch.equal(ref[0], ref[1]) and torch.equal(ref[0], ref[2])
    for key, got in outs.items():
        for a, b in zip(got, ref):
            assert torch.equal(a, b), key


def test_volume_sampler_with_per_sample_conditions_is_captured(M, dev):
    """PUNetG(dimension=3) under the captured sampler with one embedded-condition row per sample: the per-slice expansion of
    the time-shift rows lands in a workspace buffer (a capture must not allocate), graph and eager runs agree bit for bit, and
    every sample equals the run of that sample alone with its own label."""
    from tests.golden_util import load
    g, sd = load("punetg8_3d")
    torch.manual_seed(4)
    net = M.PUNetG(M.PUNetGConfig(model_channels=8, dimension=3), conditional_embedding=_TableEmbedding(10, 8))
    net.load_state_dict(sd, strict=False)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev).eval()
    wn = g["white_noise"].to(dev)
    B = wn.shape[0]
    labels = (torch.arange(B, device=dev) * 3 % 10).float()
    outs = []
    for use_graph in (False, True, True):
        module.use_graph = use_graph
        outs.append(module.propagate_white_noise(wn, y=labels, guidance=1.0, nsteps=3).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert len(module._plans.plans) == 1
    module.use_graph = False
    for b in range(B):
        alone = module.propagate_white_noise(wn[b:b + 1], y=labels[b:b + 1], guidance=1.0, nsteps=3)
        assert torch.equal(alone[0], outs[0][b])
    assert not torch.equal(outs[0][0], module.propagate_white_noise(wn[:1], y=labels[1:2], guidance=1.0, nsteps=3)[0])
    # guided: one evaluation of batch 2B on volumes, against two evaluations, eagerly and captured
    res = {}
    for batched in (False, True):
        module.batch_cfg = batched
        for use_graph in (False, True):
            module.use_graph = use_graph
            res[(batched, use_graph)] = module.propagate_white_noise(wn, y=labels, guidance=2.0, nsteps=3).clone()
    for got in res.values():
        assert torch.equal(got, res[(False, False)])
    assert not torch.equal(res[(False, False)], outs[0])


def test_adm_guided_sampling_batched(M, dev):
    """ADM under classifier-free guidance: the batched evaluation (FiLM rows of the unconditional half behind the conditional
    ones) against two evaluations, eagerly and captured, with one condition row per sample."""
    from tests.golden_util import load
    v, sd = load("adm8_concat")
    torch.manual_seed(6)
    net = M.ADM(M.ADMConfig(model_channels=8, time_embed_dim=8, output_embed_dim=16), conditional_embedding=_TableEmbedding(10, 16))
    net.load_state_dict(sd, strict=False)
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev).eval()
    wn = torch.randn(3, 1, 32, 32, device=dev)
    labels = torch.tensor([1.0, 7.0, 4.0], device=dev)
    res = {}
    for batched in (False, True):
        module.batch_cfg = batched
        for use_graph in (False, True):
            module.use_graph = use_graph
            res[(batched, use_graph)] = module.propagate_white_noise(wn, y=labels, guidance=2.0, nsteps=3).clone()
    for got in res.values():
        assert torch.isfinite(got).all() and torch.equal(got, res[(False, False)])
    module.batch_cfg = True
    assert not torch.equal(res[(True, True)], module.propagate_white_noise(wn, y=labels, guidance=1.0, nsteps=3))


def _torch_images(a):
    """The image layout of ds_inorm_silu_images from an fp32 tensor, with torch ops."""
    B, C, H, W = a.shape
    nch = (C + 15) // 16
    ap = torch.zeros(B, nch * 16, H + 2, W + 2, device=a.device)
    ap[:, :C, 1:-1, 1:-1] = a
    hi = ap.half()
    lo = (ap - hi.float()).half()
    v = torch.stack([hi, lo], dim=1).view(B, 2, nch, 2, 8, H + 2, W + 2).permute(0, 2, 1, 3, 5, 6, 4).contiguous()
    return v.view(torch.float32).reshape(-1)


@pytest.mark.parametrize("shape", [(2, 32, 32, 32), (3, 40, 16, 24), (1, 64, 8, 8), (2, 96, 20, 12), (1, 32, 64, 64), (2, 64, 48, 40)])
def test_norm_images_and_image_input_convolution(dev, shape):
    """Standalone norm + SiLU written as the convolution's pre-split fp16 hi / lo images (ds_inorm_silu_images), and the
    convolution that stages them by LDS-DMA (ds_conv2d_h3_img): both bit-identical to the fp32 route (ds_inorm_silu, then
    ds_conv2d_h3 splitting in its loader), border and channel padding zero."""
    from diffsci_amd import ops
    B, C, H, W = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, device=dev) * 1.3 + 0.2
    w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    exact = H * W <= 1024                   # larger planes: the statistics are summed in another order than ds_inorm_silu's
    for kind in (0, 1):
        act = ops.inorm_silu(x, w, b, kind=kind)
        img = ops.inorm_silu_images(x, w, b, kind)
        if exact:
            assert torch.equal(img.view(torch.int32), _torch_images(act).view(torch.int32))
        else:
            nch = (C + 15) // 16
            v = img.view(torch.float16).view(B, nch, 2, 2, H + 2, W + 2, 8).float()
            dec = (v[:, :, 0] + v[:, :, 1]).permute(0, 1, 2, 5, 3, 4).reshape(B, nch * 16, H + 2, W + 2)
            assert float(dec[:, :, 0].abs().max()) == 0.0 and float(dec[:, :, :, -1].abs().max()) == 0.0
            assert float(dec[:, C:].abs().max()) == 0.0 if nch * 16 > C else True
            assert float((dec[:, :C, 1:-1, 1:-1] - act).norm() / act.norm()) < 1e-6
    even = ((C + 15) // 16) % 2 == 0
    wt = torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5)
    pw = ops.pack_conv(wt, "fp16x3")
    bias, res = torch.randn(C, device=dev), torch.randn(B, C, H, W, device=dev)
    shift = torch.randn(B, C, device=dev)
    if not even:
        with pytest.raises(RuntimeError, match="even number of 16-channel chunks"):
            ops.conv_img(img, pw, B, C, H, W)
        return
    ts_a = torch.zeros(B, C, ops.conv_tile_count(H, W), 4, device=dev)
    ts_b = torch.zeros_like(ts_a)
    want = ops.conv(act, pw, bias=bias, shift=shift, res1=res, tile_stats=ts_a, in_amax=ops.NORMALISED)   # as the networks call it
    got = ops.conv_img(img, pw, B, C, H, W, bias=bias, shift=shift, res1=res, tile_stats=ts_b)
    if exact:
        assert torch.equal(got, want) and torch.equal(ts_a, ts_b)
    else:
        assert float((got - want).norm() / want.norm()) < 2e-6


def test_network_with_norm_images_is_bit_identical(M, dev):
    """PUNetG with standalone norms: the image route (norm kernel writes pre-split images, the convolution DMAs them) against the
    fp32 route, eagerly and inside the captured sampler, and against the CPU oracle."""
    from oracle import punetg_ref
    from tests.golden_util import rel_l2
    cfg = punetg_ref.default_config(model_channels=32)
    sd = punetg_ref.random_state_dict(cfg, seed=5)
    net = M.PUNetG(M.PUNetGConfig(model_channels=32))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    torch.manual_seed(1)
    x, t = torch.randn(2, 1, 32, 32, device=dev), torch.tensor([0.3, -0.9], device=dev)
    from diffsci_amd import ops
    calls = []
    orig = ops.conv_img
    ops.conv_img = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        net.fuse_norm = False                                   # every block on standalone norms
        with_images = net(x, t).cpu()
        n_all = len(calls)
        net.fuse_norm = True                                    # shipped selection: only the 128-channel level here ... none fused above fuse_max_cot
        shipped = net(x, t).cpu()
    finally:
        ops.conv_img = orig
    assert n_all >= 20
    net.norm_images = False
    net.fuse_norm = False
    plain = net(x, t).cpu()
    net.fuse_norm, net.norm_images = True, True
    assert torch.equal(with_images, plain)
    with torch.inference_mode():
        want = punetg_ref.punetg_forward(sd, cfg, x.cpu(), t.cpu())
    assert rel_l2(with_images, want) < 1e-5 and rel_l2(shipped, want) < 1e-5
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev)
    wn = torch.randn(2, 1, 32, 32, device=dev)
    net.fuse_norm = False
    outs = []
    for images in (True, False):
        net.norm_images = images
        for use_graph in (False, True):
            module.use_graph = use_graph
            outs.append(module.propagate_white_noise(wn, nsteps=3).clone())
    net.fuse_norm, net.norm_images = True, True
    assert all(torch.equal(o, outs[0]) for o in outs)



@pytest.mark.parametrize("shape", [(2, 32, 16, 16), (1, 64, 24, 40), (3, 96, 8, 12), (2, 40, 16, 16)])
def test_group_norm_images_and_upsampled_residual(dev, shape):
    """GroupNorm(1, C) / GroupRMSNorm apply (+ FiLM, + AvgPool(2)) written as pre-split images (ds_gnorm1_apply_images) against
    ds_gnorm1_apply, and the image-input convolution with the nearest-x2 residual (DS_RES1_UPSAMPLED) against ds_conv2d_h3:
    bit-identical."""
    from diffsci_amd import ops
    B, C, H, W = shape
    torch.manual_seed(sum(shape) + 1)
    x = torch.randn(B, C, H, W, device=dev) * 0.8 - 0.1
    w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    film = torch.randn(B, 2 * C, device=dev) * 0.3
    for kind in (0, 1):
        st = ops.gnorm1_stats(x, kind, eps=1e-5)
        for pool, fl in ((False, None), (True, None), (False, film), (False, film[:1].contiguous())):
            act = ops.gnorm1_apply(x, st, w, b, kind, pool=pool, film=fl)
            img = ops.gnorm1_apply_images(x, st, w, b, kind, pool=pool, film=fl)
            assert torch.equal(img.view(torch.int32), _torch_images(act).view(torch.int32)), (kind, pool, fl is not None)
    if ((C + 15) // 16) % 2:
        return
    wt = torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5)
    pw = ops.pack_conv(wt, "fp16x3")
    bias = torch.randn(C, device=dev)
    low = torch.randn(B, C, H // 2, W // 2, device=dev)
    ts_a = torch.zeros(B, C, ops.conv_tile_count(H, W), 4, device=dev)
    ts_b = torch.zeros_like(ts_a)
    want = ops.conv(act, pw, bias=bias, res1=low, res1_upsampled=True, tile_stats=ts_a, in_amax=ops.NORMALISED)
    got = ops.conv_img(img, pw, B, C, H, W, bias=bias, res1=low, res1_upsampled=True, tile_stats=ts_b)
    assert torch.equal(got, want) and torch.equal(ts_a, ts_b)
    with pytest.raises(ValueError, match="res1_upsampled"):
        ops.conv_img(img, pw, B, C, H, W, res1=torch.zeros(B, C, H, W, device=dev), res1_upsampled=True)


def test_adm_with_norm_images_is_bit_identical(M, dev):
    """ADM: the standalone-norm blocks on the image route against the fp32 route, eagerly and captured."""
    from diffsci_amd import ops
    torch.manual_seed(8)
    net = M.ADM(M.ADMConfig(model_channels=32, time_embed_dim=16, output_embed_dim=32)).to(dev).eval()
    for p in net.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    x, t = torch.randn(2, 1, 32, 32, device=dev), torch.tensor([0.4, -0.7], device=dev)
    calls = []
    orig = ops.conv_img
    ops.conv_img = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    outs = {}
    try:
        for fuse in (False, True):
            net.fuse_norm = fuse
            for images in (True, False):
                net.norm_images = images
                n0 = len(calls)
                outs[(fuse, images)] = net(x, t).clone()
                assert (len(calls) > n0) == images
    finally:
        ops.conv_img = orig
    for fuse in (False, True):
        assert torch.isfinite(outs[(fuse, True)]).all() and torch.equal(outs[(fuse, True)], outs[(fuse, False)])
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev)
    wn = torch.randn(2, 1, 32, 32, device=dev)
    res = []
    for images in (True, False):
        net.norm_images = images
        for use_graph in (False, True):
            module.use_graph = use_graph
            res.append(module.propagate_white_noise(wn, nsteps=3).clone())
    net.norm_images = True
    assert all(torch.equal(r, res[0]) for r in res)


def test_group_norm_statistics_from_tile_statistics(M, dev):
    """ds_gnorm1_stats_tiles: the (mean, rstd) / (0, RMS denominator) pairs from the tile statistics two convolutions left (the
    channel concatenation of their outputs) against ds_gnorm1_stats over the concatenated tensor; and ADM with its standalone
    norms fed that way against passes over the tensors."""
    from diffsci_amd import ops
    torch.manual_seed(21)
    B, H, W = 3, 24, 40
    outs, tss = [], []
    for C in (32, 48):
        x = torch.randn(B, 16, H, W, device=dev)
        pw = ops.pack_conv(torch.randn(C, 16, 3, 3, device=dev) / 12, "fp16x3")
        ts = torch.zeros(B, C, ops.conv_tile_count(H, W), 4, device=dev)
        outs.append(ops.conv(x, pw, bias=torch.randn(C, device=dev) * 3, tile_stats=ts))
        tss.append(ts)
    cat = torch.cat(outs, dim=1).contiguous()
    for kind in (0, 1):
        want = ops.gnorm1_stats(cat, kind, eps=1e-5)
        got = ops.gnorm1_stats_tiles(tss[0], kind, cat[0].numel(), stats_b=tss[1], eps=1e-5)
        assert torch.allclose(got, want, rtol=2e-6, atol=1e-7), (kind, got, want)
        one = ops.gnorm1_stats_tiles(tss[1], kind, outs[1][0].numel(), eps=1e-5)
        assert torch.allclose(one, ops.gnorm1_stats(outs[1], kind, eps=1e-5), rtol=2e-6, atol=1e-7)
    net = M.ADM(M.ADMConfig(model_channels=32, time_embed_dim=16, output_embed_dim=32)).to(dev).eval()
    x, t = torch.randn(2, 1, 32, 32, device=dev), torch.tensor([0.4, -0.7], device=dev)
    calls = []
    orig = ops.gnorm1_stats_tiles
    ops.gnorm1_stats_tiles = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        a = net(x, t).clone()
    finally:
        ops.gnorm1_stats_tiles = orig
    net.tile_stats_norms = False
    b = net(x, t).clone()
    assert calls and float((a - b).norm() / b.norm()) < 2e-6


@pytest.mark.parametrize("shape", [(2, 32, 16, 32, 64), (1, 40, 16, 16, 24), (3, 96, 8, 64, 128), (2, 16, 32, 32, 8)])
def test_upsampling_convolution_with_image_input(dev, shape):
    """ds_conv2d_h3_up_img (parity kernels of conv3x3(nearest_x2(a)), a as pre-split images staged by LDS-DMA) against
    ds_conv2d_h3_up on the fp32 activation: bit-identical outputs and tile statistics."""
    from diffsci_amd import ops
    B, C, Hl, Wl, Co = shape
    torch.manual_seed(sum(shape))
    a = torch.randn(B, C, Hl, Wl, device=dev)
    pw = ops.pack_conv(torch.randn(Co, C, 3, 3, device=dev) / (3 * C ** 0.5), "fp16x3", upsampled=True)
    assert ops.conv_up_img_supported(pw, Hl, Wl)
    img = _torch_images(a)
    bias, shift = torch.randn(Co, device=dev), torch.randn(B, Co, device=dev)
    res = torch.randn(B, Co, 2 * Hl, 2 * Wl, device=dev)
    ts_a = torch.zeros(B, Co, ops.conv_tile_count(2 * Hl, 2 * Wl), 4, device=dev)
    ts_b = torch.zeros_like(ts_a)
    from diffsci_amd._native import DS_LOAD_UPSAMPLE2
    want = ops.conv(a, pw, bias=bias, shift=shift, res1=res, load_mode=DS_LOAD_UPSAMPLE2, tile_stats=ts_a, in_amax=ops.NORMALISED)
    got = ops.conv_up_img(img, pw, B, C, Hl, Wl, bias=bias, shift=shift, res1=res, tile_stats=ts_b)
    assert torch.equal(got, want) and torch.equal(ts_a, ts_b)
    assert not ops.conv_up_img_supported(pw, 12, 20)


def test_table_images_for_large_planes(M, dev):
    """Planes beyond the image norm kernel's 4096 floats: the activation comes from the fused loader's table
    (ds_table_apply_images) -- the kernel against torch on the table's formula, and PUNetG at 96 x 96 against the CPU oracle and
    against its fp32 route."""
    from diffsci_amd import ops
    from oracle import punetg_ref
    from tests.golden_util import rel_l2
    torch.manual_seed(31)
    B, C, H, W = 2, 40, 70, 90
    x = torch.randn(B, C, H, W, device=dev) * 1.2 + 0.3
    tab = torch.zeros(B, ops.table_channels(C), 4, device=dev)
    tab[:, :C, 0], tab[:, :C, 1], tab[:, :C, 2] = torch.randn(B, C, device=dev) * 0.3, torch.rand(B, C, device=dev) + 0.5, torch.randn(B, C, device=dev) * 0.2
    v = (x - tab[:, :C, 0, None, None]) * tab[:, :C, 1, None, None] + tab[:, :C, 2, None, None]
    want = _torch_images(torch.nn.functional.silu(v.double()).float())
    got = ops.table_apply_images(x, tab)
    nch = (C + 15) // 16
    dec = lambda im: (lambda t: (t[:, :, 0] + t[:, :, 1]))(im.view(torch.float16).view(B, nch, 2, 2, H + 2, W + 2, 8).float())   # noqa: E731
    assert float((dec(got) - dec(want)).abs().max()) < 4e-6 * float(dec(want).abs().max())
    assert float(dec(got)[:, :, :, 0].abs().max()) == 0.0 and float(dec(got)[:, :, :, :, -1].abs().max()) == 0.0      # zero border
    cfg = punetg_ref.default_config(model_channels=32)
    sd = punetg_ref.random_state_dict(cfg, seed=9)
    net = M.PUNetG(M.PUNetGConfig(model_channels=32))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    x, t = torch.randn(1, 1, 96, 96, device=dev), torch.tensor([0.2], device=dev)
    calls = []
    orig = ops.table_apply_images
    ops.table_apply_images = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        net.fuse_max_cot = 0                              # every block on standalone norms: level 0 has 9216-float planes
        with_images = net(x, t).cpu()
    finally:
        ops.table_apply_images = orig
    assert len(calls) >= 4
    net.norm_images = False
    plain = net(x, t).cpu()
    with torch.inference_mode():
        ref = punetg_ref.punetg_forward(sd, cfg, x.cpu(), t.cpu())
        ref64 = punetg_ref.punetg_forward({k: w.double() for k, w in sd.items()}, cfg, x.double().cpu(), t.double().cpu())
    # this deep random-weight network at 96 x 96 amplifies fp32 rounding: the bound is the reference's own fp32-vs-fp64 distance
    tol = max(1e-5, 4 * rel_l2(ref, ref64))
    assert rel_l2(with_images, ref64) < tol and rel_l2(plain, ref64) < tol and rel_l2(with_images, plain) < tol
