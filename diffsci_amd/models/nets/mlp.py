"""MLPUncond (toy score net of BASELINE config 1) and MLPCond on ds_linear.
Reference: diffsci/models/nets/mlp.py:4-121; same state_dict keys (net.{0,2,..}.{weight,bias})."""
import torch

from ... import ops


def _mlp_stack(in_dim, out_dim, hidden_dims, nonlinearity, dropout):
    if not isinstance(nonlinearity, torch.nn.ReLU):
        raise NotImplementedError("only ReLU hidden activations are implemented on the HIP path")
    if dropout > 0:
        raise NotImplementedError("dropout > 0 is a training feature; not on the sampling path")
    layers = []
    for h in hidden_dims:
        layers += [torch.nn.Linear(in_dim, h), torch.nn.Identity()]       # Identity keeps the key numbering
        in_dim = h
    layers.append(torch.nn.Linear(in_dim, out_dim))
    return torch.nn.Sequential(*layers)


def _mlp_forward(net, h):
    lin = [m for m in net if isinstance(m, torch.nn.Linear)]
    for i, m in enumerate(lin):
        h = ops.linear(h, m.weight, m.bias, act=2 if i < len(lin) - 1 else 0)
    return h


class MLPUncond(torch.nn.Module):
    def __init__(self, dim, hidden_dims=[10], nonlinearity=torch.nn.ReLU(), dropout=0.0):
        super().__init__()
        self.dim = dim
        self.net = _mlp_stack(dim + 1, dim, hidden_dims, nonlinearity, dropout)

    @ops.device_guard
    def forward(self, x, t):
        ops.require_device(x, "x")
        h = torch.cat([x, t.to(x)[..., None]], dim=-1).contiguous()       # mlp.py:55-57 (pure data movement)
        return _mlp_forward(self.net, h)


class MLPCond(torch.nn.Module):
    """mlp.py:61-121: the same stack on [x, t, y]; state_dict keys net.{0,2,..}.{weight,bias}."""

    def __init__(self, dim, ydim, hidden_dims=[10], nonlinearity=torch.nn.ReLU(), dropout=0.0):
        super().__init__()
        self.dim = dim
        self.ydim = ydim
        self.net = _mlp_stack(dim + 1 + ydim, dim, hidden_dims, nonlinearity, dropout)

    @ops.device_guard
    def forward(self, x, t, y):
        ops.require_device(x, "x")
        if y.shape[0] != x.shape[0]:
            y = y.expand(x.shape[0], *y.shape[1:])                         # a [1, ydim] condition serves the whole batch
        h = torch.cat([x, t.to(x)[..., None], y.to(x)], dim=-1).contiguous()   # mlp.py:119-120 (pure data movement)
        return _mlp_forward(self.net, h)
