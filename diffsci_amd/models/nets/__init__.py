from .punetg_config import PUNetGConfig  # noqa: F401
from .punetg import PUNetG, PUNetGCond  # noqa: F401
from .mlp import MLPUncond  # noqa: F401
from .adm import ADM, ADMBaseBlock, ADMConfig, ADMDecoderBlock, ADMEncoderBlock  # noqa: F401
from .embedder import PorosityEmbedder  # noqa: F401
