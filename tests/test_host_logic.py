"""Host-side logic of the drop-in surface (no GPU): sigma grid, per-step tables, configuration
objects, state_dict compatibility, batching / sharding helpers, error behaviour."""
import math
import os

import pytest
import inspect

import torch

import diffsci_amd.models as M
from diffsci_amd.models.karras.steptable import build_step_table
from diffsci_amd.parallel import global_white_noise, shard_rows
from oracle import karras_ref as K
from tests.golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exact_on_same_isa(got, want):
    v, _ = load("schedule")
    if v["cpu_capability"] == torch.backends.cpu.get_cpu_capability():
        assert torch.equal(got, want)
    else:
        torch.testing.assert_close(got, want, rtol=3e-7, atol=0)


@pytest.mark.parametrize("n", [2, 5, 10, 18, 50, 100, 256])
def test_create_steps_bit_exact(n):
    v, _ = load("schedule")
    got = M.EDMScheduler().create_steps(n + 1)
    assert got.dtype == torch.float32 and got.device.type == "cpu"
    _exact_on_same_isa(got, v[f"steps_{n}"])


def test_step_from_time_integer_path():
    v, _ = load("schedule")
    s = M.EDMScheduler()
    for n in (19, 51):
        got = s.step_from_time(v["step_from_time_t"], n)
        assert got.dtype == torch.int32 and torch.equal(got, v[f"step_from_time_{n}"])


@pytest.mark.parametrize("n", [18, 50])
def test_step_table_scalars_match_reference(n):
    """Every scalar handed to the kernels equals what the reference computes on [B]-tensors."""
    v, _ = load("schedule")
    sch = M.EDMScheduler()
    sch.create_steps = lambda k: v[f"steps_{k - 1}"].clone()
    table = build_step_table(sch, M.HeunIntegrator(), n, preconditioner=M.EDMPreconditioner())
    want = v[f"precond_B64_N{n}"]          # rows: c_skip, c_out, c_in, c_noise at t[i]
    t = v[f"steps_{n}"]
    dt = torch.diff(t)
    assert len(table.rows) == n
    for i, row in enumerate(table.rows):
        e = row.first
        got = torch.tensor([e.c_skip, e.c_out, e.c_in, e.c_noise])
        _exact_on_same_isa(got, want[i, :4])
        assert e.sigma == float(t[i]) and e.sigma_sq == float(t[i] ** 2) and e.neg_mult == float(-(t[i] * (1 + 0 * t[i])))
        assert row.dt == float(dt[i])
        if i < n - 1:
            assert row.second is not None and row.second.sigma == float(t[i] + dt[i])   # fl(t+dt), not t[i+1]
        else:
            assert row.second is None                                                   # t+dt == 0: d2 = d1
    assert len(table.evals) == 2 * n - 1


def test_fl_t_plus_dt_differs_from_grid_on_some_steps():
    """at N=18 the corrector's sigma is 1 ulp off t[i+1] on a few steps; the table keeps fl(t+dt)."""
    sch = M.EDMScheduler()
    t = sch.create_steps(19)
    table = build_step_table(sch, M.HeunIntegrator(), 18)
    diffs = [i for i, r in enumerate(table.rows[:-1]) if r.second.sigma != float(t[i + 1])]
    for i in diffs:
        assert abs(table.rows[i].second.sigma - float(t[i + 1])) <= 2e-7 * float(t[i + 1])
    assert all(r.second.sigma == float(t[i] + (t[i + 1] - t[i])) for i, r in enumerate(table.rows[:-1]))


def test_karras_and_em_tables_follow_oracle_scalars():
    sch = M.EDMScheduler()
    sch.langevin_const = 0.7
    t = sch.create_steps(7)
    dt = torch.diff(t)
    tab = build_step_table(sch, M.KarrasIntegrator(), 6, preconditioner=M.EDMPreconditioner())
    for i, row in enumerate(tab.rows):
        back = min(40 / 6, math.sqrt(2) - 1) if 0.05 <= t[i] <= 50 else 0
        sig_hat = t[i] + back * t[i]
        assert row.first.sigma == float(sig_hat)
        assert row.churn_coef == float(torch.sqrt(sig_hat ** 2 - t[i] ** 2) * 1.003)
        assert row.dt == float((t[i] + dt[i]) - sig_hat)
    tab = build_step_table(sch, M.EulerMaruyamaIntegrator(), 6)
    for i, row in enumerate(tab.rows):
        assert row.first.stochastic and row.first.neg_lang == float(-K.langevin_factor(t[i], 0.7))
        assert row.noise_coef == float(K.noise_injection(t[i], langevin_const=0.7))
        assert row.sqrt_abs_dt == float(torch.sqrt(torch.abs(dt[i])))
    sch.langevin_interval = (0.1, 10.0)
    tab = build_step_table(sch, M.EulerMaruyamaIntegrator(), 6)
    assert [r.noise_coef > 0 for r in tab.rows] == [bool(0.1 < float(x) < 10.0) for x in t[:-1]]


def test_forward_table_skips_first_level():
    sch = M.EDMScheduler()
    tab = build_step_table(sch, M.EulerIntegrator(), 10, backward=False)
    t = sch.create_steps(11).flip(0)
    assert len(tab.rows) == 9 and tab.rows[0].first.sigma == float(t[1]) and tab.rows[0].dt > 0


def test_negative_time_is_rejected():
    sch = M.EDMScheduler()
    sch.create_steps = lambda n: torch.tensor([1.0, 0.5, -0.1])
    with pytest.raises(ValueError, match="t\\+dt < 0"):
        build_step_table(sch, M.HeunIntegrator(), 2)


def test_integrator_names_and_defaults():
    assert isinstance(M.name_to_integrator("euler"), M.EulerIntegrator)
    assert isinstance(M.name_to_integrator("heun"), M.HeunIntegrator)
    assert isinstance(M.name_to_integrator("euler-maruyama"), M.EulerMaruyamaIntegrator)
    k = M.name_to_integrator("karras")
    assert (k.s_schurn, k.s_tmin, k.s_tmax, k.s_noise) == (40, 0.05, 50, 1.003) and k.need_fns
    with pytest.raises(ValueError, match="Unknown integrator"):
        M.name_to_integrator("rk4")
    s = M.EDMScheduler()
    assert isinstance(s.integrator, M.HeunIntegrator) and s.maximum_scale == 80.0
    s.set_temporary_integrator("euler")
    assert isinstance(s.integrator, M.EulerIntegrator)
    s.unset_temporary_integrator()
    assert isinstance(s.integrator, M.HeunIntegrator)
    assert s.langevin_const == 1.0 and s.langevin_interval is None


def test_config_factory_and_module_surface():
    cfg = M.KarrasModuleConfig.from_edm(sigma_data=0.5)
    assert isinstance(cfg.preconditioner, M.EDMPreconditioner)
    assert isinstance(cfg.noisescheduler, M.EDMScheduler) and cfg.tag == "edm"
    assert float(cfg.preconditioner.noise_conditioner(torch.tensor(80.0))) == float(0.5 * torch.log(torch.tensor(80.0)))
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    module = M.KarrasModule(net, cfg)
    # like the reference, only the network is in the module tree
    assert all(k.startswith("model.") for k in module.state_dict())
    assert module.device.type == "cpu" and module.norm == 1.0
    for name in ("sample", "propagate_white_noise", "propagate_toward_sample", "get_score", "get_denoiser",
                 "encode", "decode"):
        assert callable(getattr(module, name))
    with pytest.raises(RuntimeError, match="no CPU path"):
        module.propagate_white_noise(torch.randn(1, 1, 32, 32), nsteps=2)
    vp, ve = M.KarrasModuleConfig.from_vp(), M.KarrasModuleConfig.from_ve()
    assert isinstance(vp.noisescheduler, M.VPScheduler) and isinstance(vp.preconditioner, M.VPPreconditioner) and vp.tag == "vp"
    assert isinstance(ve.noisescheduler, M.VEScheduler) and isinstance(ve.preconditioner, M.VEPreconditioner) and ve.tag == "ve"
    assert not vp.noisescheduler.scheduler_fns.constant_scaling_fn and ve.noisescheduler.scheduler_fns.constant_scaling_fn
    assert ve.noisescheduler.create_steps(5)[0] == 100.0 ** 2 and vp.noisescheduler.create_steps(5)[0] == 1.0


def test_punetg_state_dict_keys_match_reference():
    _, sd = load("punetg8_forward")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    mine = net.state_dict()
    assert set(mine) == set(sd)
    assert all(tuple(mine[k].shape) == tuple(sd[k].shape) for k in sd)
    net.load_state_dict(sd)            # strict
    full = M.PUNetG(M.PUNetGConfig())
    assert sum(p.numel() for p in full.parameters()) == 11_314_689
    assert len(full.state_dict()) == 213


def test_punetg_config_roundtrip_and_unsupported_options():
    c = M.PUNetGConfig(model_channels=32, channel_expansion=[2, 2, 4])
    d = c.export_description()
    assert d["model_channels"] == 32 and d["first_resblock_norm"] == "GroupLN" and d["time_projection_scale"] == 30.0
    c2 = M.PUNetGConfig.from_description(d)
    assert c2.export_description() == d and c2.extended_channel_expansion == [1, 2, 2, 4]
    with pytest.raises(NotImplementedError, match="convolution_type"):
        M.PUNetG(M.PUNetGConfig(convolution_type="spherical"))
    # the layer variants keep the reference's state_dict keys and shapes (checkpoints load strictly)
    for tag, over in (("mp", dict(convolution_type="mp")),
                      ("pix_ln", dict(first_resblock_norm="GroupPix", second_resblock_norm="GroupLN")),
                      ("none_rms_noaffine", dict(first_resblock_norm="none", second_resblock_norm="GroupRMS", affine_norm=False)),
                      ("cosine", dict(attn_type="cosine"))):
        _, sd = load("punetg8_" + tag)
        net = M.PUNetG(M.PUNetGConfig(model_channels=8, **over))
        mine = net.state_dict()
        assert set(mine) == set(sd) and all(tuple(mine[k].shape) == tuple(sd[k].shape) for k in sd), tag
        net.load_state_dict(sd)
    # checkpoints trained with dropout / condition dropping load too (both are the identity under eval())
    dn = M.PUNetG(M.PUNetGConfig(model_channels=8, dropout=0.1, cond_dropout=0.2, cond_drop=0.3))
    assert tuple(dn.state_dict()["cond_drop.null_embedding"].shape) == (1, 8)
    assert "cond_drop.null_embedding" not in M.PUNetG(M.PUNetGConfig(model_channels=8)).state_dict()
    mp = M.PUNetG(M.PUNetGConfig(model_channels=8, convolution_type="mp"))
    assert "attn_block.0.mhattn.q_proj_matrix" in mp.state_dict() and float(mp.convin.weight.std()) > 0.5   # N(0,1) init
    circ = M.PUNetG(M.PUNetGConfig(model_channels=8, convolution_type="circular"))
    assert "convin.conv.weight" in circ.state_dict() and "downsamplers.0.conv.conv.bias" in circ.state_dict()
    with pytest.raises(NotImplementedError, match="dimension"):
        M.PUNetG(M.PUNetGConfig(dimension=1))
    vol = M.PUNetG(M.PUNetGConfig(model_channels=8, dimension=3))
    assert tuple(vol.convin.weight.shape) == (8, 1, 3, 3, 3) and len(vol.state_dict()) == 213
    with pytest.raises(TypeError):
        M.PUNetGConfig(not_an_option=1)


def test_mlp_state_dict_keys_match_reference():
    _, sd = load("mlp_cfg1")
    m = M.MLPUncond(2, [20])
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd)


def test_minibatch_sizes_and_row_shards():
    from diffsci_amd.models.karras.karrasmodule import get_minibatch_sizes
    assert get_minibatch_sizes(10, 4) == [4, 4, 2] and get_minibatch_sizes(8, 4) == [4, 4]
    assert get_minibatch_sizes(3, 5) == [3]
    for total, world in ((512, 8), (10, 4), (3, 8), (64, 1)):
        spans = [shard_rows(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_global_noise_is_the_reference_draw():
    """Rank slices reproduce torch.manual_seed(s); torch.randn(G, *shape) of karrasmodule.py:837."""
    torch.manual_seed(123)
    want = torch.randn(6, 1, 4, 4)
    state = torch.random.get_rng_state()
    got = torch.cat([global_white_noise(6, [1, 4, 4], 123, rows=shard_rows(6, 4, r)) for r in range(4)])
    assert torch.equal(got, want)
    assert torch.equal(torch.random.get_rng_state(), state)      # global generator untouched


def test_config_descriptions_round_trip():
    """KarrasModuleConfig.export_description / load_from_description_with_tag (karrasmodule.py:343-366) and
    KarrasModule.export_description (:462-474)."""
    for ctor, kw in ((M.KarrasModuleConfig.from_edm, dict(sigma_data=0.7)), (M.KarrasModuleConfig.from_ve, dict(sigma_max=50.0)),
                     (M.KarrasModuleConfig.from_vp, dict(beta_min=0.2))):
        c = ctor(**kw)
        d = c.export_description()
        c2 = M.KarrasModuleConfig.load_from_description_with_tag(d)
        assert c2.export_description() == d and type(c2.noisescheduler) is type(c.noisescheduler)
    with pytest.raises(ValueError, match="custom"):
        M.KarrasModuleConfig.load_from_description_with_tag(dict(tag="custom", extra_args={}))
    with pytest.raises(ValueError, match="Unknown tag"):
        M.KarrasModuleConfig.load_from_description_with_tag(dict(tag="nope", extra_args={}))
    module = M.KarrasModule(M.MLPUncond(2, [20]), M.KarrasModuleConfig.from_edm(), conditional=True)
    d = module.export_description()
    assert d["conditional"] is True and d["autoencoder"] is False and d["config_description"]["tag"] == "edm"
    assert not module.config.has_dynamic_loss_weight
    from diffsci_amd.models.karras import UniformNoiseSampler, LatentSpaceAutoregressive
    assert issubclass(M.KarrasModule, LatentSpaceAutoregressive)
    torch.manual_seed(0)
    s = UniformNoiseSampler(t=0.5, T=2.0).sample([1000])
    assert float(s.min()) >= 0.5 and float(s.max()) <= 2.0


def test_global_noise_rows_without_the_full_tensor():
    """A rank draws only its rows (chunked skip of the lower ranks' rows): identical to slicing the full draw."""
    import diffsci_amd.parallel as P
    old = P._CHUNK_FLOATS
    P._CHUNK_FLOATS = 1 << 10
    try:
        for total, shape in ((8, [1, 4, 4]), (5, [1, 4, 4]), (7, [3, 8, 8]), (4, [1, 3, 3]), (6, [2, 16, 16]), (1, [1, 4, 4])):
            torch.manual_seed(7)
            ref = torch.randn(total, *shape)
            for world in (1, 2, 3, 4):
                for rank in range(world):
                    lo, hi = P.shard_rows(total, world, rank)
                    assert torch.equal(P.global_white_noise(total, shape, 7, rows=(lo, hi)), ref[lo:hi])
    finally:
        P._CHUNK_FLOATS = old


def test_plan_keys_depend_on_condition_structure_not_values():
    """one-hot labels 3 and 7 (sum = |sum| = 1) and a fresh tensor in a recycled block must not be
    told apart -- or confused -- by a plan key: the key is structural, the values are refreshed before every replay."""
    from diffsci_amd.models.karras.engine import condition_signature
    a, b = torch.zeros(80), torch.zeros(80)
    a[3], b[7] = 1.0, 1.0
    assert condition_signature(a) == condition_signature(b)
    assert condition_signature({"y": a, "z": torch.zeros(2, 3)}) == condition_signature({"z": torch.ones(2, 3), "y": b})
    assert condition_signature(a) != condition_signature(torch.zeros(81))
    assert condition_signature({"y": a}) != condition_signature({"w": a})
    assert condition_signature(None) is None


def test_plan_keys_and_plan_copies_follow_lists_tuples_and_scalars():
    """with capture_eager the plan owns a copy of the condition that every replay rewrites.  Tensors inside
    a list or tuple must be cloned and rewritten like those inside a dict, and a leaf the graph cannot rewrite -- a Python
    scalar or string -- has to be part of the key: a different value is a different plan, never a stale replay."""
    from diffsci_amd.models.karras.engine import clone_condition, condition_signature, copy_condition
    t1, t2 = torch.arange(4.0), torch.arange(4.0) + 10
    y1 = {"fields": [t1, (t1 * 2, 3)], "scale": 0.5, "mode": "a"}
    y2 = {"fields": [t2, (t2 * 2, 3)], "scale": 0.5, "mode": "a"}
    assert condition_signature(y1) == condition_signature(y2)                       # tensor values only: same plan
    assert condition_signature(y1) != condition_signature(dict(y1, scale=0.25))     # scalar leaf: in the key
    assert condition_signature(y1) != condition_signature(dict(y1, mode="b"))
    assert condition_signature({"f": [t1]}) != condition_signature({"f": (t1,)})    # container kind
    assert condition_signature([t1, t1]) != condition_signature([t1])
    own = clone_condition(y1)
    assert own["fields"][0] is not t1 and own["fields"][1][0] is not y1["fields"][1][0] and isinstance(own["fields"][1], tuple)
    copy_condition(own, y2)                                                          # what refresh() does before a replay
    assert torch.equal(own["fields"][0], t2) and torch.equal(own["fields"][1][0], t2 * 2) and own["fields"][1][1] == 3
    assert torch.equal(y1["fields"][0], t1)                                          # the caller's tensors are untouched


def test_plan_cache_evicts_the_least_recently_used():
    """a hit protects a plan (the cache was FIFO)."""
    from diffsci_amd.models.karras import engine
    src = inspect.getsource(engine.PlanCache.run)
    assert "self.plans.pop(key, None)" in src and "self.plans[key] = plan" in src
    plans = {}
    for k in "abc":
        plans[k] = k
    hit = plans.pop("a")
    plans[hit] = hit                                     # the move-to-back the cache performs on a hit
    assert next(iter(plans)) == "b"


def test_bench_self_launch_command():
    """`python bench.py --gpus N` with no launcher around it starts N ranks through torch.distributed.run on
    127.0.0.1 and hands its own arguments on."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "2"], 4, 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")


def test_bench_self_launch_relays_the_rank_zero_line_and_the_exit_code(tmp_path, capfd):
    """bench.py --gpus N without a launcher starts its own N ranks (torch.distributed.run as a child process, before anything
    touches the GPU) and relays rank 0's JSON line and the ranks' exit status -- driven here end to end with a stub script in
    place of bench.py (two CPU ranks)."""
    import argparse
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import json, os, sys\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'\n"
        "if rank == 0:\n"
        "    print(json.dumps({'metric': 'stub', 'n_gpus': world, 'argv': sys.argv[1:]}), flush=True)\n"
        "sys.exit(int(sys.argv[sys.argv.index('--exit') + 1]) if (rank == 1 and '--exit' in sys.argv) else 0)\n")
    args = argparse.Namespace(gpus=2)
    rc = bench.self_launch(args, argv=["--gpus", "2", "--steps", "1"], script=str(stub))
    out = capfd.readouterr().out
    line = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert rc == 0 and len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["argv"] == ["--gpus", "2", "--steps", "1"]
    rc = bench.self_launch(args, argv=["--gpus", "2", "--exit", "3"], script=str(stub))
    assert rc != 0                                             # a failing rank fails the run
    cmd = bench.launcher_command(["--gpus", "4"], 4, 1234)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[-3].endswith("bench.py")


def test_global_noise_skip_ahead_is_self_checked(monkeypatch):
    """The chunked per-rank draw leans on a detail of torch's CPU normal fill; the first use in a process checks it on a small
    case, and a torch that fails the check gets the full draw (same values) instead of silently different noise."""
    from diffsci_amd import parallel
    monkeypatch.setattr(parallel, "_SKIP_AHEAD_OK", None)
    assert parallel.skip_ahead_matches_full_draw() is True and parallel._SKIP_AHEAD_OK is True     # this torch passes
    full = parallel.global_white_noise(6, [2, 4, 4], seed=3)
    assert torch.equal(parallel.global_white_noise(6, [2, 4, 4], seed=3, rows=(2, 5)), full[2:5])
    calls = []
    real = torch.randn
    monkeypatch.setattr(torch, "randn", lambda *a, **k: (calls.append(a), real(*a, **k))[1])
    parallel.global_white_noise(6, [2, 4, 4], seed=3, rows=(2, 5))
    assert all(a[0] < 6 for a in calls)                        # chunked: never the whole tensor
    monkeypatch.setattr(parallel, "_SKIP_AHEAD_OK", False)     # a torch whose generator does not skip ahead like that
    calls.clear()
    assert torch.equal(parallel.global_white_noise(6, [2, 4, 4], seed=3, rows=(2, 5)), full[2:5])
    assert calls[0][0] == 6                                     # the full draw


def test_model_signature_follows_the_module_tree():
    """engine.model_signature caches the walk over the module tree (the plan key is built once per run); the cache must see an
    in-place update, a parameter assigned anew, and a submodule replaced, added or removed."""
    import copy
    import diffsci_amd.models as M
    from diffsci_amd.models.karras import engine
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    s0 = engine.model_signature(net)
    assert engine.model_signature(net) == s0 and len(s0[0]) == len(list(net.parameters())) + len(list(net.buffers()))
    with torch.no_grad():
        net.time_projection.W.mul_(1.5)                    # a buffer the plan's time-shift tables are computed from
    assert engine.model_signature(net) != s0
    s0 = engine.model_signature(net)
    with torch.no_grad():
        net.convin.weight.add_(1.0)
    s1 = engine.model_signature(net)
    assert s1 != s0
    net.convin.weight = torch.nn.Parameter(net.convin.weight.detach().clone())
    s2 = engine.model_signature(net)
    assert s2 != s1
    net.convin = copy.deepcopy(net.convin)
    s3 = engine.model_signature(net)
    assert s3 != s2
    net.extra_thing = torch.nn.Linear(3, 3)
    s4 = engine.model_signature(net)
    assert len(s4[0]) == len(s3[0]) + 2
    del net.extra_thing
    assert engine.model_signature(net) == s3
    net.conv_precision = "fp32"
    assert engine.model_signature(net) != s3


def test_condition_copies_keep_structure_and_follow_values():
    """engine.clone_condition / copy_condition: the plan-owned copy of a condition (dicts of tensors, nested, with non-tensor
    entries) that a captured evaluated-as-given run reads and every replay rewrites."""
    from diffsci_amd.models.karras.engine import clone_condition, condition_signature, copy_condition
    y = {"label": torch.tensor([1.0, 2.0]), "nested": {"field": torch.ones(2, 3)}, "tag": "abc", "n": 3}
    c = clone_condition(y)
    assert condition_signature(c) == condition_signature(y)
    assert c["label"] is not y["label"] and torch.equal(c["label"], y["label"]) and c["tag"] == "abc" and c["n"] == 3
    keep = (c["label"].data_ptr(), c["nested"]["field"].data_ptr())
    y2 = {"label": torch.tensor([5.0, 6.0]), "nested": {"field": torch.full((2, 3), 7.0)}, "tag": "abc", "n": 3}
    copy_condition(c, y2)
    assert torch.equal(c["label"], y2["label"]) and torch.equal(c["nested"]["field"], y2["nested"]["field"])
    assert keep == (c["label"].data_ptr(), c["nested"]["field"].data_ptr())          # same buffers: a captured graph reads them
    t = torch.arange(4.0)
    ct = clone_condition(t)
    copy_condition(ct, t * 2)
    assert torch.equal(ct, t * 2) and clone_condition(None) is None
