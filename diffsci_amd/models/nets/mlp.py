"""MLPUncond (toy score net of BASELINE config 1) on ds_linear.
Reference: diffsci/models/nets/mlp.py:4-58; same state_dict keys (net.{0,2,..}.{weight,bias})."""
import torch

from ... import ops


class MLPUncond(torch.nn.Module):
    def __init__(self, dim, hidden_dims=[10], nonlinearity=torch.nn.ReLU(), dropout=0.0):
        super().__init__()
        if not isinstance(nonlinearity, torch.nn.ReLU):
            raise NotImplementedError("only ReLU hidden activations are implemented on the HIP path")
        if dropout > 0:
            raise NotImplementedError("dropout > 0 is a training feature; not on the sampling path")
        self.dim = dim
        layers, in_dim = [], dim + 1
        for h in hidden_dims:
            layers += [torch.nn.Linear(in_dim, h), torch.nn.Identity()]   # Identity keeps the key numbering
            in_dim = h
        layers.append(torch.nn.Linear(in_dim, dim))
        self.net = torch.nn.Sequential(*layers)

    @ops.device_guard
    def forward(self, x, t):
        ops.require_device(x, "x")
        h = torch.cat([x, t.to(x)[..., None]], dim=-1).contiguous()       # mlp.py:55-57 (pure data movement)
        lin = [m for m in self.net if isinstance(m, torch.nn.Linear)]
        for i, m in enumerate(lin):
            h = ops.linear(h, m.weight, m.bias, act=2 if i < len(lin) - 1 else 0)
        return h
