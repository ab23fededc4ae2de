"""CPU oracle for the SIModule sampler (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates, in the reference's operation order, diffsci/models/karras/flowfield.py:
  :21-111   SIScheduler.linear / cosine / edm (finterpolation)
  :153-169  Preconditioner.edm (and identity: v = model(x, t, y=y))
  :441-458  get_flow_field with classifier-free guidance on the flow fields
  :460-481  get_score_field
  :503-544  sample (linspace(1, 0, nsteps), x * sigma(t0))
  :704-795  integrate_flow_field / integration_step (Heun, last step Euler)
  :546-702  inpaint (Euler-Maruyama + re-imposed known region + jumps) and the soft mask
"""
import numpy as np
import torch

from .karras_ref import _bcast


def scheduler(name, expoent=7.0, sigma_min=0.02, sigma_max=80.0):
    if name == "linear":
        return dict(alpha=lambda t: 1 - t, sigma=lambda t: t, alpha_dot=lambda t: -1 * torch.ones_like(t),
                    sigma_dot=lambda t: torch.ones_like(t))
    if name == "cosine":
        return dict(alpha=lambda t: torch.cos(t * np.pi / 2), sigma=lambda t: torch.sin(t * np.pi / 2),
                    alpha_dot=lambda t: -1 * torch.pi / 2 * torch.sin(t * np.pi / 2),
                    sigma_dot=lambda t: torch.pi / 2 * torch.cos(t * np.pi / 2))
    if name == "edm":
        f = lambda x: x ** expoent                      # noqa: E731
        finv = lambda x: x ** (1 / expoent)             # noqa: E731
        fdot = lambda x: expoent * x ** (expoent - 1)   # noqa: E731
        return dict(alpha=lambda t: 0.0 * t + 1.0, alpha_dot=lambda t: 0.0 * t,
                    sigma=lambda t: f((1 - t) * finv(sigma_min) + t * finv(sigma_max)),
                    sigma_dot=lambda t: fdot((1 - t) * finv(sigma_min) + t * finv(sigma_max)) * (finv(sigma_max) - finv(sigma_min)))
    raise ValueError(name)


def precondition(sch, kind, model, x, t, y=None, sigma_data=0.5):
    if callable(kind):                                   # user precondition callable (flowfield.py:140-144)
        return kind(model, x, t, y=y)
    if kind == "auto_identity":                          # autonomous flows: the network sees no time (:147-150, 162-164)
        return model(x, y=y)
    if kind == "auto_edm":
        sigma = _bcast(sch["sigma"](t), x)
        cin = 1 / torch.sqrt(sigma_data ** 2 + sigma ** 2)
        cout = sigma * sigma_data / torch.sqrt(sigma_data ** 2 + sigma ** 2)
        cskip = sigma_data ** 2 / (sigma_data ** 2 + sigma ** 2)
        return cskip * x + cout * model(x / cin, y=y)
    if kind == "identity":
        return model(x, t, y=y)
    sigma = _bcast(sch["sigma"](t), x)
    sigma_dot = _bcast(sch["sigma_dot"](t), x)
    cin = 1 / torch.sqrt(sigma_data ** 2 + sigma ** 2)
    cout = sigma * sigma_data / torch.sqrt(sigma_data ** 2 + sigma ** 2)
    cskip = sigma_data ** 2 / (sigma_data ** 2 + sigma ** 2)
    cnoise = 0.5 * torch.log(sch["sigma"](t))
    denoiser = cskip * x + cout * model(cin * x, cnoise, y=y)
    return sigma_dot / sigma * (x - denoiser)


def flow_field(sch, kind, model, x, t, y=None, guidance=1.0):
    if guidance == 1.0 or y is None:
        return precondition(sch, kind, model, x, t, y=y)
    v = precondition(sch, kind, model, x, t, y=y)
    vu = precondition(sch, kind, model, x, t, y=None)
    return guidance * v + (1 - guidance) * vu


def score_field(sch, kind, model, x, t, y=None, guidance=1.0):
    v = flow_field(sch, kind, model, x, t, y, guidance)
    alpha, sigma = _bcast(sch["alpha"](t), x), _bcast(sch["sigma"](t), x)
    alpha_dot, sigma_dot = _bcast(sch["alpha_dot"](t), x), _bcast(sch["sigma_dot"](t), x)
    return (alpha * v - alpha_dot * x) / (sigma * (alpha_dot * sigma - alpha * sigma_dot))


def integrate(sch, kind, model, x, time_schedule, y=None, guidance=1.0, return_history=False):
    hist = [x]
    n = len(time_schedule)
    for i in range(n - 1):
        t_curr = time_schedule[i] * torch.ones(x.shape[0]).to(x)
        t_next = time_schedule[i + 1] * torch.ones(x.shape[0]).to(x)
        dt = _bcast(t_next - t_curr, x)
        v1 = flow_field(sch, kind, model, x, t_curr, y, guidance)
        if i == n - 2:
            x = x + dt * v1
        else:
            v2 = flow_field(sch, kind, model, x + dt * v1, t_next, y, guidance)
            x = x + dt * (v1 + v2) / 2
        hist.append(x)
    return torch.stack(hist) if return_history else x


def sample(sch, kind, model, noise, nsteps, y=None, guidance=1.0, norm_sigma=None):
    ts = torch.linspace(1, 0, nsteps).to(noise)
    x = integrate(sch, kind, model, noise * sch["sigma"](ts[0]), ts, y, guidance)
    return x if norm_sigma is None else x * norm_sigma


def integration_step(sch, kind, model, x, t_curr, t_next, method="euler", y=None, guidance=1.0):
    """flowfield.py:749-781 (deterministic methods, integrate_on_sigma=False)."""
    dt = _bcast(t_next - t_curr, x)
    v1 = flow_field(sch, kind, model, x, t_curr, y, guidance)
    if method == "euler":
        return x + dt * v1
    v2 = flow_field(sch, kind, model, x + dt * v1, t_next, y, guidance)
    return x + dt * (v1 + v2) / 2


def score_from_flow(sch, v, x, t):
    """flowfield.py:483-501."""
    alpha, sigma = _bcast(sch["alpha"](t), x), _bcast(sch["sigma"](t), x)
    alpha_dot, sigma_dot = _bcast(sch["alpha_dot"](t), x), _bcast(sch["sigma_dot"](t), x)
    return (alpha * v - alpha_dot * x) / (sigma * (alpha_dot * sigma - alpha * sigma_dot))


def soft_mask(mask, falloff):
    """flowfield.py:643-702 (2-D)."""
    import torch.nn.functional as F
    m = mask.unsqueeze(0).float()
    k, p = 2 * falloff + 1, falloff
    dil = F.avg_pool2d(m, kernel_size=k, stride=1, padding=p)
    ero = F.avg_pool2d(1 - m, kernel_size=k, stride=1, padding=p)
    s = dil / (dil + ero + 1e-8)
    return ((1 - torch.cos(s * np.pi)) / 2).squeeze(0)


def inpaint(sch, kind, model, x_orig, mask, orig_noise, nsteps, draws, norm_sigma=None, mask_falloff=0,
            resample_steps=0, jump_length=1, mask_start_t=1.0):
    """flowfield.py:546-641 with the loop's randn_like draws supplied in order (`draws`)."""
    it = iter(draws)
    sm = soft_mask(mask, mask_falloff) if mask_falloff > 0 else mask
    xo = x_orig.unsqueeze(0)
    if norm_sigma is not None:
        xo = xo / norm_sigma
    ts = torch.linspace(1, 0, nsteps).to(orig_noise)
    x = orig_noise * sch["sigma"](ts[0])
    for i in range(nsteps - 1):
        t_curr = ts[i] * torch.ones(x.shape[0]).to(x)
        t_next = ts[i + 1] * torch.ones(x.shape[0]).to(x)
        for r in range(resample_steps + 1):
            dt = _bcast(t_next - t_curr, x)
            v = flow_field(sch, kind, model, x, t_curr)
            sc = score_from_flow(sch, v, x, t_curr)
            omega = _bcast(sch["sigma"](t_curr), x)
            x = x + dt * (v - 0.5 * omega * sc)
            x = x + torch.sqrt(omega * torch.abs(dt)) * next(it)
            if ts[i + 1].item() <= mask_start_t:
                sigma, alpha = _bcast(sch["sigma"](t_next), xo), _bcast(sch["alpha"](t_next), xo)
                x = (1 - sm) * x + sm * (alpha * xo + sigma * next(it))
                if r < resample_steps and i + jump_length < nsteps - 1:
                    sj, aj = _bcast(sch["sigma"](ts[i]), x), _bcast(sch["alpha"](ts[i]), x)
                    x = aj * x + sj * next(it)
                    x = (1 - sm) * x + sm * (aj * xo + sj * next(it))
    return x if norm_sigma is None else x * norm_sigma
