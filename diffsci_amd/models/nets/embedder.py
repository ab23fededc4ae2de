"""Conditional embedders of the sampling path on HIP kernels.

PorosityEmbedder: same constructor, ``forward({'porosity': [B, 1]})`` protocol and state_dict keys
(``gaussian_proj.W``, ``net.{0,2,4}.{weight,bias}``) as the reference (diffsci/models/nets/embedder.py:198-229);
the layers are parameter containers, the arithmetic is ds_fourier_features + ds_linear.
"""
import torch

from ... import ops
from .punetg import _Fourier


class PorosityEmbedder(torch.nn.Module):
    def __init__(self, dembed, scale=30.0):
        super().__init__()
        self.dembed = dembed
        self.scale = scale
        self.gaussian_proj = _Fourier(dembed, scale)
        self.net = torch.nn.Sequential(
            torch.nn.Linear(dembed, 4 * dembed), torch.nn.Identity(),
            torch.nn.Linear(4 * dembed, 4 * dembed), torch.nn.Identity(),
            torch.nn.Linear(4 * dembed, dembed))

    def forward(self, x):
        p = x["porosity"].squeeze(-1).reshape(-1).to(torch.float32).contiguous()      # [nbatch]
        ops.require_device(p, "y['porosity']")
        h = ops.fourier_features(p, self.gaussian_proj.W)
        h = ops.linear(h, self.net[0].weight, self.net[0].bias, act=1)
        h = ops.linear(h, self.net[2].weight, self.net[2].bias, act=1)
        return ops.linear(h, self.net[4].weight, self.net[4].bias, act=0)

    def export_description(self):
        return {"dembed": self.dembed, "scale": self.scale}
