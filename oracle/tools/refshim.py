"""Container-only loader for the read-only reference at /root/reference.

TEST INFRASTRUCTURE. Used solely by oracle/tools/make_golden.py to emit the
fixtures under tests/golden/. Never imported by diffsci_amd, never shipped to
the GPU box's run-time path (the reference does not exist there).

The reference imports four packages that are absent from this image and that
contribute no arithmetic to the sampling path:
  jaxtyping  - annotations only (Float[Tensor, "..."])
  lightning  - LightningModule used only as the nn.Module base class
  diffusers / torchvision - imported by off-path wrappers
This module pre-seeds sys.modules with annotation/base-class placeholders for
them so that `import diffsci.models` succeeds; no reference code is modified
or copied.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


class _Ann:
    def __class_getitem__(cls, item):
        return cls


def _mk(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install():
    if "diffsci" in sys.modules:
        return sys.modules["diffsci"]
    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Bool", "Shaped", "Int", "Integer"):
        setattr(jt, n, type(n, (_Ann,), {}))
    sys.modules["jaxtyping"] = jt

    L = _mk("lightning")

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def save_hyperparameters(self, *a, **k):
            pass

    class _CB:
        def __init__(self, *a, **k):
            pass

    L.LightningModule = LightningModule
    L.Trainer = object
    L.Callback = _CB
    Lp = _mk("lightning.pytorch")
    L.pytorch = Lp
    Lc = _mk("lightning.pytorch.callbacks")
    Lp.callbacks = Lc
    Lc.Callback = _CB
    Lc.StochasticWeightAveraging = _CB
    _mk("diffusers")
    _mk("torchvision")
    sys.path.insert(0, REFERENCE_ROOT)
    import diffsci.models  # noqa: F401
    import diffsci.data  # noqa: F401
    return sys.modules["diffsci"]
