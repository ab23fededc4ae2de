"""diffsci_amd -- MI355X-native implementation of DiffSci's Karras-EDM sampling path.

Drop-in for ``diffsci.models`` on that path: ``diffsci_amd.models`` exposes KarrasModule,
KarrasModuleConfig, EDMScheduler, the integrators / preconditioners and the PUNetG score
network with the reference's names, signatures and state_dict keys.  All tensor work runs in
hand-written HIP kernels (libdiffsci_hip.so, include/diffsci_hip.h); there is no CPU path.
"""
from . import _native, ops  # noqa: F401
from . import models  # noqa: F401

__version__ = "0.1.0"
