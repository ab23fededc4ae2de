"""CPU oracle for the conditional embedders on the path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PorosityEmbedder (diffsci/models/nets/embedder.py:198-229): y = {'porosity': [B, 1]} ->
GaussianFourierProjection(dembed)(porosity) -> Linear(d,4d) SiLU Linear(4d,4d) SiLU Linear(4d,d).
"""
import torch.nn.functional as F

from .punetg_ref import fourier_features


def porosity_embed(sd, prefix, y):
    x = y["porosity"].squeeze(-1)
    h = fourier_features(x, sd[prefix + "gaussian_proj.W"])
    h = F.silu(F.linear(h, sd[prefix + "net.0.weight"], sd[prefix + "net.0.bias"]))
    h = F.silu(F.linear(h, sd[prefix + "net.2.weight"], sd[prefix + "net.2.bias"]))
    return F.linear(h, sd[prefix + "net.4.weight"], sd[prefix + "net.4.bias"])
