"""The Python surface of diffsci_amd.models against the reference's, mechanically ("same names, signatures, defaults").

tests/golden/api_surface.json is generated from the imported reference by oracle/tools/make_golden.py (api_surface): for every
class the hot path names, every public method / property it defines itself (torch / Lightning machinery excluded) with its
ordered parameters and defaults, plus the module-level factories and which names each package namespace re-exports.

Rule for a method: our signature STARTS with the reference's parameters -- same names, same kinds, same defaults, same order --
so every positional or keyword call a reference user writes binds the same way.  Parameters after those are extensions; each
must have a default and is listed in EXTENSIONS below with what it is for.

NOT_BUILT lists what is absent on purpose, by category.  Everything else must be there.
"""
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "api_surface.json")) as f:
    SURFACE = json.load(f)

# ---- training: losses, optimisers, Lightning hooks (out of scope for the sampling path) ----
TRAINING = {
    "training_step", "validation_step", "configure_optimizers", "loss_fn", "set_optimizer_and_scheduler", "set_loss_metric",
    "autoregressive_loss_fn", "has_autoregressive_loss", "log_autoregressive_step_losses", "start_dynamic_loss_weight",
    "select_batch", "sample_timestep", "get_loss_summary", "update_loss_metric", "set_loss_metric_module", "set_loss_weighting",
}
# ---- partial forwards and layer factories INSIDE one ADM residual block (adm.py:315-453): they return / chain the reference's
#      torch layer objects (GroupNorm, AvgPool2d, Upsample, Conv2d instances) between the steps of ADMBaseBlock.forward.  Here a
#      block's forward is a fixed sequence of fused HIP launches (norm + SiLU + pooling in one kernel, FiLM + SiLU in another):
#      there is no tensor "after norm2 but before FiLM" to hand out, and no torch layer object to return.  The block's public
#      protocol -- constructor, forward(x, te, skip), state_dict keys -- is mirrored and pinned by goldens (adm_blocks). ----
ADM_BLOCK_INTERNALS = {"first_block", "second_block", "embed_block", "residual_block", "make_image_sample", "make_downsample",
                       "make_upsample", "make_norm_layers", "make_attn_layer", "conv_fn", "get_channels_in_modified"}
# ---- SIModuleConfig's post-construction setters mutate training-time members (loss weighting, metric); its two sampling-time
#      ones are constructor arguments here as there ----
NOT_BUILT = {
    "karras.flowfield.SIModuleConfig": {"set_preconditioner", "set_scheduling_functions"},
    "nets.adm.ADMBaseBlock": ADM_BLOCK_INTERNALS,
    "nets.adm.ADMEncoderBlock": ADM_BLOCK_INTERNALS,
    "nets.adm.ADMDecoderBlock": ADM_BLOCK_INTERNALS,
}

# (class, method) -> extension parameters (all keyword-with-default, after the reference's own)
NOISE = "injected noise draws instead of the device generator: what the parity tests replay the reference's recorded draws through"
EXTENSIONS = {
    ("*Scheduler", "propagate"): {"eps": NOISE}, ("*Scheduler", "propagate_backward"): {"eps": NOISE},
    ("*Scheduler", "propagate_forward"): {"eps": NOISE}, ("*Scheduler", "propagate_partial"): {"eps": NOISE},
    ("*Scheduler", "renoise"): {"noise": NOISE}, ("*Scheduler", "repaint"): {"noise": NOISE},
    ("KarrasModule", "propagate_white_noise"): {"eps": NOISE},
    ("KarrasModule", "propagate_toward_sample"): {"eps": NOISE, "_scale": "private: the scheduler's maximum scale for the latent path"},
    ("KarrasModule", "propagate_toward_noise"): {"eps": NOISE},
    ("KarrasModule", "propagate_repaint_toward_sample"): {"noise": NOISE},
    ("KarrasModule", "propagate_partial_toward_sample"): {"guidance": "classifier-free guidance on a partial run", "eps": NOISE},
    ("SIModule", "inpaint"): {"noise": NOISE},
}


def _default_repr(v):
    if v is inspect.Parameter.empty:
        return "<required>"
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if isinstance(v, (list, tuple)) and all(x is None or isinstance(x, (bool, int, float, str)) for x in v):
        return list(v)
    return "<" + type(v).__name__ + ">"


def _signature(fn, drop_cls=False):
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        if name == "self":
            continue
        out.append([name, p.kind.name, _default_repr(p.default)])
    if drop_cls and out and out[0][0] == "cls":
        out = out[1:]
    return out


def _extensions(cname, mname):
    for (c, m), ext in EXTENSIONS.items():
        if m == mname and (c == cname or (c.startswith("*") and cname.endswith(c[1:]))):
            return ext
    return {}


def _cases():
    for key, info in sorted(SURFACE["classes"].items()):
        for mname, minfo in sorted(info["methods"].items()):
            yield key, mname, minfo


def test_every_class_exists_where_the_reference_has_it():
    for key in SURFACE["classes"]:
        modpath, cname = key.rsplit(".", 1)
        mod = importlib.import_module("diffsci_amd.models." + modpath)
        assert inspect.isclass(getattr(mod, cname, None)), f"diffsci_amd.models.{key} is missing"


def test_package_namespaces_reexport_the_same_names():
    for ns, names in SURFACE["exports"].items():
        mod = importlib.import_module("diffsci_amd.models" + ("." + ns if ns else ""))
        missing = [n for n in names if not hasattr(mod, n)]
        assert not missing, f"diffsci_amd.models{'.' + ns if ns else ''} does not export {missing}"


def test_factory_functions():
    for key, ref in SURFACE["functions"].items():
        modpath, fname = key.rsplit(".", 1)
        fn = getattr(importlib.import_module("diffsci_amd.models." + modpath), fname)
        assert _signature(fn) == ref, key


@pytest.mark.parametrize("key,mname,minfo", list(_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_method_signature(key, mname, minfo):
    modpath, cname = key.rsplit(".", 1)
    cls = getattr(importlib.import_module("diffsci_amd.models." + modpath), cname)
    if mname in TRAINING or mname in NOT_BUILT.get(key, ()):
        pytest.skip("not built: see TRAINING / NOT_BUILT at the top of this file")
    obj = inspect.getattr_static(cls, mname, None)
    assert obj is not None, f"{key}.{mname} is missing"
    if minfo["kind"] == "property":
        assert isinstance(obj, property) or not callable(obj), f"{key}.{mname} should be a property / attribute"
        return
    assert isinstance(obj, staticmethod) == (minfo["kind"] == "staticmethod"), f"{key}.{mname}: staticmethod mismatch"
    assert isinstance(obj, classmethod) == (minfo["kind"] == "classmethod"), f"{key}.{mname}: classmethod mismatch"
    fn = obj.__func__ if isinstance(obj, (staticmethod, classmethod)) else obj
    got = _signature(fn, drop_cls=isinstance(obj, classmethod))
    ref = minfo["params"]
    assert got[:len(ref)] == ref, f"{key}.{mname}\n  reference: {ref}\n  ours:      {got}"
    extra = got[len(ref):]
    allowed = _extensions(cname, mname)
    for name, kind, default in extra:
        assert name in allowed, f"{key}.{mname}: parameter '{name}' is neither the reference's nor a listed extension"
        assert default != "<required>" and kind in ("POSITIONAL_OR_KEYWORD", "KEYWORD_ONLY"), f"{key}.{mname}: extension '{name}' needs a default"


def test_allow_lists_name_only_things_the_reference_has():
    """An entry that matches nothing would hide a typo: every allow-listed name must occur in the reference's surface."""
    all_methods = {m for info in SURFACE["classes"].values() for m in info["methods"]}
    assert TRAINING <= all_methods, sorted(TRAINING - all_methods)
    for key, names in NOT_BUILT.items():
        assert set(names) <= set(SURFACE["classes"][key]["methods"]), key
    for (c, m), _ in EXTENSIONS.items():
        assert m in all_methods, (c, m)


def test_positional_configs_like_the_reference():
    import diffsci_amd.models as M
    c = M.PUNetGConfig(1, 1, 2, 32, [1, 2])
    assert (c.model_channels, c.channel_expansion) == (32, [1, 2])
    a = M.ADMConfig(3, 3, 2, 128, 128, 512, [1, 2, 4, 4])
    assert (a.model_channels, a.time_embed_dim, a.output_embed_dim, a.middle_channel) == (128, 128, 512, 512)
    k = M.KarrasModuleConfig.from_edm(0.5, -1.2, 1.2, True)
    assert k.has_edm_batch_norm and k.extra_args["autoregressive_loss_steps"] == 1 and k.extra_args["focus_radius"] is None
    assert M.KarrasModuleConfig.load_from_description_with_tag(k.export_description()).tag == "edm"


def test_receptive_field_matches_the_reference_formula():
    import diffsci_amd.models as M
    net = M.PUNetG(M.PUNetGConfig(model_channels=8, number_resnet_attn_block=1))
    r = net.calculate_receptive_field()
    # convin 2 | two levels: 2 blocks x 2 convs x 2 s, pool s, conv 2 (2s) | bottom 5 blocks at stride 4 | up | convout
    rf, s = 1 + 2, 1
    for _ in range(2):
        rf += 2 * 2 * 2 * s + s
        s *= 2
        rf += 2 * s
    rf += (2 + 1 + 2) * 2 * 2 * s
    for _ in range(2):
        s //= 2
        rf += 2 * s + 2 * 2 * 2 * s
    rf += 2
    assert r["rf"] == rf and not r["has_attention"] and r["downsampling_factor"] == 4 and r["feasible_chunking"]
    assert M.PUNetG(M.PUNetGConfig(model_channels=8)).calculate_receptive_field()["rf"] == float("inf")


def test_lightning_checkpoint_loads_on_the_host():
    """karrasmodule.py:410-429: the reference's .ckpt layout (fixture written from a reference module by make_golden.py)."""
    import numpy as np
    import torch
    import diffsci_amd.models as M
    path = os.path.join(HERE, "golden", "ckpt8_lightning.ckpt")
    net = M.PUNetG(M.PUNetGConfig(model_channels=8))
    cfg = M.KarrasModuleConfig.from_edm(has_edm_batch_norm=True)
    module = M.KarrasModule.load_from_checkpoint(path, model=net, config=cfg)
    keys = sorted(np.load(os.path.join(HERE, "golden", "ckpt8.npz"))["keys"].tolist())
    assert sorted(module.state_dict()) == keys
    assert module.model is not net and module.config is not cfg                   # deep copies, as in the reference
    assert float(module.edm_batch_norm.running_var) == pytest.approx(1.7)
    sd = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
    assert all(torch.equal(module.state_dict()[k], sd[k]) for k in keys)
    with pytest.raises(KeyError):
        torch.save({"weights": {}}, "/tmp/not_lightning.ckpt")
        M.KarrasModule.load_from_checkpoint("/tmp/not_lightning.ckpt", model=net, config=cfg)
