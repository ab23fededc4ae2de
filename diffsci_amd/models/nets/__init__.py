from .punetg_config import PUNetGConfig  # noqa: F401
from .punetg import PUNetG, PUNetGCond  # noqa: F401
from .mlp import MLPCond, MLPUncond  # noqa: F401
from .adm import (ADM, ADMBaseBlock, ADMConfig, ADMDecoder, ADMDecoderBlock, ADMEncoder, ADMEncoderBlock,  # noqa: F401
                  ADMMiddleBlock, ADMTimeEmbedding)
from .embedder import PorosityEmbedder  # noqa: F401
