"""DimensionAgnosticBatchNorm -- the data-space <-> network-space affine map the reference wraps around
its sampling loop when ``KarrasModuleConfig.has_edm_batch_norm`` (diffsci/models/karras/edmbatchnorm.py ->
diffsci/models/aux_scripts/batchnorm.py:86-170; used at karrasmodule.py:1209-1210,1225-1226,1236-1241).

Sampling only needs the eval-mode arithmetic (running statistics); both directions are one launch of
``ds_batchnorm_eval``.  Training-mode statistics updates belong to the reference's training loop and raise here.
"""
import torch

from ... import ops


class DimensionAgnosticBatchNorm(torch.nn.Module):
    def __init__(self, num_channels: int | None = None, eps: float = 1e-5, affine: bool = False,
                 momentum: float = 0.1, sigma: float = 1.0):
        super().__init__()
        self.num_channels = num_channels
        self.nc = num_channels if num_channels is not None else 1       # 1 broadcasts over the channels
        self.eps = eps
        self.affine = affine
        self.momentum = momentum
        self.sigma = sigma
        if affine:
            self.weight = torch.nn.Parameter(torch.ones(self.nc))
            self.bias = torch.nn.Parameter(torch.zeros(self.nc))
        self.register_buffer("running_mean", torch.zeros(self.nc))
        self.register_buffer("running_var", torch.ones(self.nc))

    def _apply_kernel(self, x, inverse):
        if self.training:
            raise NotImplementedError("DimensionAgnosticBatchNorm: batch statistics (training mode) are outside the "
                                      "HIP sampling path; call .eval()")
        if x.dim() < 2:
            raise ValueError("expected x of shape (N, C, *spatial)")
        w, b = (self.weight, self.bias) if self.affine else (None, None)
        return ops.batchnorm_eval(x.contiguous(), self.running_mean, self.running_var, w, b, eps=self.eps,
                                  sigma=self.sigma, inverse=inverse)

    def forward(self, x):
        return self._apply_kernel(x, False)

    def normalize(self, x):
        return self(x)

    def unnorm(self, x):
        return self._apply_kernel(x, True)

    def unnormalize(self, x):
        return self.unnorm(x)


class ConstantBatchNorm(torch.nn.Module):
    """aux_scripts/batchnorm.py:172-187: x / sigma and back."""

    def __init__(self, sigma: float = 1.0):
        super().__init__()
        self.sigma = sigma

    def forward(self, x):
        return ops.div_scalar(x.contiguous(), self.sigma)

    def unnorm(self, x):
        return ops.scale(x.contiguous(), self.sigma)

    normalize = forward
    unnormalize = unnorm


class IdentityBatchNorm(torch.nn.Module):
    """aux_scripts/batchnorm.py:190-204."""

    def forward(self, x):
        return x

    def unnorm(self, x):
        return x

    normalize = forward
    unnormalize = unnorm
