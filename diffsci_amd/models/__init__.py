"""Namespace mirroring ``diffsci.models`` for the Karras-EDM sampling path."""
from . import karras, nets  # noqa: F401
from .karras import *  # noqa: F401,F403
from .nets import PUNetG, PUNetGCond, PUNetGConfig, MLPUncond, MLPCond, ADM, ADMConfig, PorosityEmbedder  # noqa: F401
