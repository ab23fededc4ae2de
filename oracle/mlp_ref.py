"""CPU oracle for the toy MLP score net (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates diffsci/models/nets/mlp.py:4-58 (MLPUncond): Linear(dim+1,h)-ReLU-...-Linear(h,dim)
applied to cat[x, t[:,None]], driven by a state_dict with the reference's key names
(``net.{0,2,...}.{weight,bias}``).
"""
import torch
import torch.nn.functional as F


def mlp_uncond_forward(sd, x, t):
    h = torch.cat([x, t[..., None]], dim=-1)
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("net.")})
    for n, i in enumerate(idx):
        h = F.linear(h, sd[f"net.{i}.weight"], sd[f"net.{i}.bias"])
        if n < len(idx) - 1:
            h = F.relu(h)
    return h


def make_net(sd):
    def net(x, t, y=None):
        return mlp_uncond_forward(sd, x, t)
    return net
