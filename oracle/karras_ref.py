"""CPU oracle for the Karras-EDM stepper (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Functional restatement, in the reference's own operation order, of
  diffsci/models/karras/schedulers.py      (sigma grid, rhs, N-step loop)
  diffsci/models/karras/integrators.py     (Euler, Heun, Euler-Maruyama, Karras churn)
  diffsci/models/karras/preconditioners.py (EDM / Null c_skip, c_out, c_in, c_noise)
  diffsci/models/karras/karrasmodule.py    (denoiser, score, CFG, white-noise propagation)
Every elementwise operation is written in the order the reference evaluates it,
so that on the same host this code is bit-identical to the reference in fp32 and
fp64 (checked by tests/test_oracle_golden.py against tests/golden/).

Only the EDM scheduling functions (s(t)=1, sigma(t)=t) are restated: that is the
"constant scaling" branch of Scheduler.rhs, the one BASELINE.json's path uses.
"""
import math

import torch


# --------------------------------------------------------------------------- sigma grid
def edm_sigma_grid(n, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    """n noise levels, last one 0.  schedulers.py:377-385 (EDMScheduler.create_steps).

    Always evaluated in fp32 on the CPU, exactly like the reference (its buffers
    never leave the CPU); callers cast with ``.to(x)``.
    The torch op sequence is kept verbatim because tensor**tensor pow on the CPU
    is not position-independent in the last ulp (vector body vs scalar tail).
    """
    smin = torch.tensor(sigma_min)
    smax = torch.tensor(sigma_max)
    rho_t = torch.tensor(rho)
    s = torch.arange(n - 1).to(rho_t) / (n - 2)
    start = smax ** (1 / rho_t)
    end = smin ** (1 / rho_t)
    steps = (start + s * (end - start)) ** rho_t
    return torch.cat([steps, torch.zeros([1]).to(steps)])


def edm_step_from_time(t, n, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    """Integer step index of noise level t.  schedulers.py:387-390 (note n-1, not n-2)."""
    smin = torch.tensor(sigma_min)
    smax = torch.tensor(sigma_max)
    exp = 1 / torch.tensor(rho)
    step = (n - 1) * (t ** exp - smax ** exp) / (smin ** exp - smax ** exp)
    return torch.round(step).int()


# --------------------------------------------------------------------------- preconditioners
def edm_precond(sigma, sigma_data=0.5):
    """(c_skip, c_out, c_in, c_noise) of preconditioners.py:35-53.

    ``sigma_data`` is an fp32 0-dim CPU buffer in the reference (it is not moved
    by module.double()), hence torch.tensor(sigma_data) here.
    """
    sd = torch.tensor(sigma_data)
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / torch.sqrt(sigma ** 2 + sd ** 2)
    c_in = 1 / torch.sqrt(sigma ** 2 + sd ** 2)
    c_noise = 0.5 * torch.log(sigma)
    return c_skip, c_out, c_in, c_noise


def null_precond(sigma):
    """preconditioners.py:139-161: D = F, network sees (x, sigma)."""
    return 0.0 * sigma, 1.0 + 0.0 * sigma, 1.0 + 0.0 * sigma, sigma


def _bcast(t, x):
    # torchutils.py:4-40 broadcast_from_below
    return t.view(t.shape + (1,) * (x.ndim - t.ndim)).to(x)


# --------------------------------------------------------------------------- denoiser / score
def denoiser(net, x, sigma, precond=edm_precond, y=None, guidance=1.0, conditional=False):
    """karrasmodule.py:673-719.  net(x_scaled, c_noise[, y]) -> F."""
    c_skip, c_out, c_in, c_noise = precond(sigma)
    c_in = _bcast(c_in, x)
    c_out = _bcast(c_out, x)
    c_skip = _bcast(c_skip, x)
    xin = c_in * x
    if conditional and guidance != 0.0:
        F = net(xin, c_noise, y)
        if guidance != 1.0:
            Fu = net(xin, c_noise)
            F = (1 - guidance) * Fu + guidance * F
    else:
        F = net(xin, c_noise)
    return c_out * F + c_skip * x


def score(net, x, sigma, **kw):
    """karrasmodule.py:721-733."""
    D = denoiser(net, x, sigma, **kw)
    return (D - x) / (_bcast(sigma, x) ** 2)


# --------------------------------------------------------------------------- rhs
def langevin_factor(t, langevin_const=1.0, langevin_interval=None):
    """schedulers.py:219-240 with the EDM functions of schedulingfunctions.py:41-63."""
    standard = (1 + 0 * t) ** 2 * (1 + 0 * t) * (1 * t)
    if langevin_interval is not None:
        t_ = t[0] if len(t.shape) > 0 else t
        if t_ > langevin_interval[0] and t_ < langevin_interval[1]:
            return langevin_const * standard + 0 * t
        return 0 * t
    return langevin_const * standard + 0 * t


def noise_injection(t, **kw):
    """schedulers.py:242-245."""
    return torch.sqrt(2 * langevin_factor(t, **kw))


def rhs_edm(x, ti, score_fn, stochastic=False, langevin_const=1.0, langevin_interval=None, backward=True):
    """Drift.  schedulers.py:247-274 (constant_scaling_fn branch); forward mode flips the Langevin term."""
    t = ti * torch.ones(x.shape[0]).to(x)
    t_ = _bcast(t, x)
    sigma = 1 * t
    sigma_deriv = 1 + 0 * t
    multiplier = _bcast(sigma, x) * _bcast(sigma_deriv, x)
    sc = score_fn(x, sigma)
    res = -multiplier * sc
    if stochastic:
        stochastic_factor = -(langevin_factor(t_, langevin_const, langevin_interval) * sc)
        if not backward:
            stochastic_factor = -stochastic_factor
        res += stochastic_factor
    return res


# --------------------------------------------------------------------------- integrators
def step_euler(x, t, dt, rhs):
    """integrators.py:29-35."""
    return x + dt * rhs(x, t)


def step_heun(x, t, dt, rhs):
    """integrators.py:38-54.  The corrector's noise level is fl(t+dt), not t[i+1]."""
    d1 = rhs(x, t)
    if (t + dt) > 0:
        xe = x + dt * d1
        d2 = rhs(xe, t + dt)
    elif (t + dt) == 0:
        d2 = d1
    else:
        raise ValueError("t+dt < 0 is not supported")
    return x + 0.5 * (d1 + d2) * dt


def step_euler_maruyama(x, t, dt, rhs, eps, **lv):
    """integrators.py:57-69; eps replaces torch.randn_like(x)."""
    return x + rhs(x, t) * dt + (noise_injection(t, **lv) * eps * torch.sqrt(torch.abs(dt)))


def step_karras(x, t, dt, rhs, eps, nsteps, s_churn=40, s_tmin=0.05, s_tmax=50, s_noise=1.003):
    """integrators.py:87-113 (EDM stochastic sampler); eps replaces randn_like(x)."""
    back = min(s_churn / nsteps, math.sqrt(2) - 1)
    # the reference uses np.sqrt(2)-1 (a numpy float64); math.sqrt gives the same double
    if s_tmin is not None:
        if not s_tmin <= t <= s_tmax:
            back = 0
    sigma = 1 * t
    sigma_hat = sigma + back * sigma
    t_hat = 1 * sigma_hat
    scale = 1 + 0 * t
    scale_hat = 1 + 0 * t_hat
    std = scale_hat * torch.sqrt(sigma_hat ** 2 - sigma ** 2)
    x_hat = (scale_hat / scale) * x + std * s_noise * eps
    d1 = rhs(x_hat, t_hat)
    dt_hat = (t + dt) - t_hat
    x = x_hat + dt_hat * d1
    if (t + dt) > 0:
        d2 = rhs(x, t + dt)
        x = x_hat + 0.5 * (d1 + d2) * dt_hat
    return x


# --------------------------------------------------------------------------- N-step loop
def propagate_backward(x, score_fn, nsteps, integrator="heun", record_history=False,
                       eps=None, sigma_grid=None, langevin_const=1.0,
                       langevin_interval=None, karras_kwargs=None):
    """schedulers.py:48-89 (Scheduler.propagate, backward=True).

    integrator: "euler" | "heun" | "euler-maruyama" | "karras".
    eps: [nsteps, *x.shape] injected noise for the stochastic integrators (one draw
         per step, in step order -- the reference calls randn_like once per step).
    sigma_grid: optional precomputed fp32 grid (nsteps+1 values); default = EDM grid.
    """
    t = (edm_sigma_grid(nsteps + 1) if sigma_grid is None else sigma_grid).to(x)
    dt = torch.diff(t)
    stochastic = integrator == "euler-maruyama"
    lv = dict(langevin_const=langevin_const, langevin_interval=langevin_interval)

    def rhs(xx, tt):
        return rhs_edm(xx, tt, score_fn, stochastic=stochastic, **lv)

    if record_history:
        history = torch.zeros([nsteps + 1] + list(x.shape)).to(x)
        history[0] = x
    for i in range(nsteps):
        if integrator == "euler":
            x = step_euler(x, t[i], dt[i], rhs)
        elif integrator == "heun":
            x = step_heun(x, t[i], dt[i], rhs)
        elif integrator == "euler-maruyama":
            x = step_euler_maruyama(x, t[i], dt[i], rhs, eps[i].to(x), **lv)
        elif integrator == "karras":
            x = step_karras(x, t[i], dt[i], rhs, eps[i].to(x), nsteps, **(karras_kwargs or {}))
        else:
            raise ValueError(f"Unknown integrator: {integrator}")
        if record_history:
            history[i + 1] = x
    return history if record_history else x


def _one_step(x, t, dt, rhs, integrator, eps, lv):
    if integrator == "euler":
        return step_euler(x, t, dt, rhs)
    if integrator == "heun":
        return step_heun(x, t, dt, rhs)
    if integrator == "euler-maruyama":
        return step_euler_maruyama(x, t, dt, rhs, eps, **lv)
    raise ValueError(f"Unknown integrator: {integrator}")


def propagate_forward(x, score_fn, nsteps, integrator="heun", record_history=False, eps=None,
                      langevin_const=1.0, langevin_interval=None):
    """schedulers.py:48-89 with backward=False: the grid is flipped (0, sigma_min, ..., sigma_max),
    the step from t=0 is skipped, history[0] stays zero.  eps: one draw per executed step."""
    t = edm_sigma_grid(nsteps + 1).to(x).flip(0)
    dt = torch.diff(t)
    lv = dict(langevin_const=langevin_const, langevin_interval=langevin_interval)

    def rhs(xx, tt):
        return rhs_edm(xx, tt, score_fn, stochastic=integrator == "euler-maruyama", backward=False, **lv)
    if record_history:
        history = torch.zeros([nsteps + 1] + list(x.shape)).to(x)
        history[1] = x
    for i in range(nsteps - 1):
        x = _one_step(x, t[i + 1], dt[i + 1], rhs, integrator, None if eps is None else eps[i].to(x), lv)
        if record_history:
            history[i + 2] = x
    return history if record_history else x


def propagate_partial(x, score_fn, nsteps, initial_step, final_step, integrator="heun"):
    """schedulers.py:178-217 (backward, deterministic integrators)."""
    t = edm_sigma_grid(nsteps + 1).to(x)
    dt = torch.diff(t)

    def rhs(xx, tt):
        return rhs_edm(xx, tt, score_fn)
    for i in range(initial_step, final_step):
        x = _one_step(x, t[i], dt[i], rhs, integrator, None, {})
    return x


def inpaint(x, y, mask, score_fn, nsteps, integrator="heun", record_history=False):
    """Scheduler.inpaint, schedulers.py:91-121."""
    if record_history:
        history = torch.zeros([nsteps + 1] + list(x.shape)).to(x)
        history[0] = x
    x = x * (1 - mask) + y[-1] * mask
    for i in range(nsteps):
        x = propagate_partial(x, score_fn, nsteps, i, i + 1, integrator)
        x = x * (1 - mask) + y[-i - 2] * mask
        if record_history:
            history[i + 1] = x
    return history if record_history else x


def renoise(x, t, t_noise, eps):
    """Scheduler.renoise, schedulers.py:166-176 (EDM: scale = 1)."""
    std = (1 + 0 * t_noise) * torch.sqrt((1 * t_noise) ** 2 - (1 * t) ** 2)
    return ((1 + 0 * t_noise) / (1 + 0 * t)) * x + std * eps


def repaint(x, y, mask, score_fn, nsteps, rsteps, nresamples, eps, integrator="heun", record_history=False):
    """Scheduler.repaint, schedulers.py:123-164.  eps: the renoise draws in order."""
    assert nsteps % rsteps == 0
    t = edm_sigma_grid(nsteps + 1).to(x)
    draws = iter(eps)
    if record_history:
        history = torch.zeros([int(nresamples * (nsteps / rsteps - 1)) + 2] + list(x.shape)).to(x)
        history[0] = x
    x = x * (1 - mask) + y[-1] * mask
    step, fstep = 0, rsteps
    x = propagate_partial(x, score_fn, nsteps, step, fstep, integrator)
    step, fstep = fstep, fstep + rsteps
    level = 0
    while fstep <= nsteps:
        x = propagate_partial(x, score_fn, nsteps, step, fstep, integrator)
        for i in range(nresamples):
            x = x * (1 - mask) + y[-fstep - 1] * mask
            if record_history:
                history[level + i + 1] = x
            x = renoise(x, t[fstep], t[step], next(draws).to(x))
            x = propagate_partial(x, score_fn, nsteps, step, fstep, integrator)
        step, fstep = fstep, fstep + rsteps
        level = level + nresamples
    if record_history:
        history[level + 1] = x
        return history
    return x


def linear_interpolation(x1, x2, n):
    """torchutils.py:64-65."""
    return torch.stack([x1 + (x2 - x1) * i / (n - 1) for i in range(n)])


def propagate_white_noise(net, white_noise, nsteps, integrator="heun", precond=edm_precond,
                          y=None, guidance=1.0, conditional=False, sigma_max=80.0, **kw):
    """karrasmodule.py:867-931 for a non-latent module (decode = identity, norm = 1.0).

    y follows the reference protocol: un-batched tensor or dict of tensors that gets
    unsqueeze(0) (karrasmodule.py:916-917).
    """
    x = white_noise * sigma_max
    if y is not None:
        y = ({k: v.unsqueeze(0) for k, v in y.items()} if isinstance(y, dict)
             else y.unsqueeze(0))

    def score_fn(xx, sigma):
        return score(net, xx, sigma, precond=precond, y=y, guidance=guidance,
                     conditional=conditional)

    with torch.inference_mode():
        out = propagate_backward(x, score_fn, nsteps, integrator=integrator, **kw)
        return out * 1.0  # decode(): x * self.norm with norm = 1.0 (karrasmodule.py:1224)


def batchnorm_eval(x, mean, var, sigma=1.0, eps=1e-5, weight=None, bias=None, inverse=False):
    """DimensionAgnosticBatchNorm.forward / .unnorm with running statistics (aux_scripts/batchnorm.py:123-170);
    mean / var / weight / bias hold 1 or C entries."""
    shape = [1, mean.numel()] + [1] * (x.dim() - 2)
    mean, var = mean.view(shape), var.view(shape)
    if inverse:
        x = x / sigma
        if weight is not None:
            x = (x - bias.view(shape)) / weight.view(shape)
        return x * torch.sqrt(var + eps) + mean
    x = (x - mean) / torch.sqrt(var + eps)
    if weight is not None:
        x = x * weight.view(shape) + bias.view(shape)
    return x * sigma


def encode(x, autoencoder=None, bn=None, norm=1.0):
    """KarrasModule.encode (karrasmodule.py:1192-1214) without y-encoding: autoencoder.encode -> batch-norm
    normalize -> / norm.  bn = dict(mean=, var=, sigma=) or None."""
    if autoencoder is not None:
        x = autoencoder.encode(x)
    if bn is not None:
        x = batchnorm_eval(x, **bn)
    return x / norm


def decode(x, autoencoder=None, bn=None, norm=1.0, record_history=False):
    """KarrasModule.decode (karrasmodule.py:1216-1234): * norm -> batch-norm unnormalize -> autoencoder.decode;
    histories are decoded slice by slice."""
    if record_history:
        return torch.stack([decode(xx, autoencoder, bn, norm) for xx in x], dim=0)
    x = x * norm
    if bn is not None:
        x = batchnorm_eval(x, inverse=True, **bn)
    if autoencoder is not None:
        x = autoencoder.decode(x)
    return x


def autoregressive_forecast(sample_fn, y0, nsamples, latent_shape, nsteps_forecast, cond_time):
    """LatentSpaceAutoregressive.autoregressive_sample in latent space (autoregressivesample.py:83-183), written
    the way the reference runs it: a prediction buffer, and a condition dict whose 'y' entry is overwritten in place
    and read back on the next step.  sample_fn(nsamples, y_dict) -> [nsamples, C, h, w].  Returns the stacked
    forecasts [nsteps_forecast, nsamples, C, h, w]."""
    C, h, w = latent_shape
    y = {"y": y0}
    buf = torch.zeros((max(cond_time, nsteps_forecast + 1), nsamples, C, h, w))
    out = [sample_fn(nsamples, y)]
    buf[0] = out[0]
    for step in range(nsteps_forecast - 1):
        count = step + 1
        if count >= cond_time:
            y["y"] = buf[count - cond_time:count][:, 0].reshape(cond_time * C, h, w)      # sample 0 of each frame
        else:
            initial = y["y"].reshape(cond_time, C, h, w)              # the window written on the previous step
            need = cond_time - count
            combined = torch.zeros((cond_time, C, h, w))
            combined[:need] = initial[cond_time - need:]
            combined[need:] = buf[:count, 0]
            y["y"] = combined.reshape(cond_time * C, h, w)
        pred = sample_fn(nsamples, y)
        buf[(step + 1) % buf.shape[0]] = pred
        out.append(pred)
    return torch.stack(out, dim=0)


# --------------------------------------------------------------------------- closed-form KATs
def gaussian_target_score(scale):
    """grad log p(x; sigma) for data ~ N(0, scale^2 I).  data/toy_datasets.py:259-279."""
    def fn(x, sigma):
        s = _bcast(sigma, x)
        sigma_mod = torch.sqrt(s ** 2 + scale ** 2)
        return -(x - 0.0) / (sigma_mod ** 2)
    return fn


def point_target_score(x0=0.0):
    """Point-mass target (ZeroDataset when x0 = 0).  data/toy_datasets.py:170-189,282-287."""
    def fn(x, sigma):
        s = _bcast(sigma, x)
        return -(x - x0) / (s ** 2)
    return fn


def heun_gain_product(sigma_grid, scale):
    """Closed form for the Heun trajectory of the linear ODE with a N(0, scale^2) target:
    x_N = x_0 * prod_i g_i, evaluated in float64 python arithmetic."""
    t = [float(v) for v in sigma_grid.double()]

    def a(tt):
        return -tt / (tt * tt + scale * scale)  # dx/dt = -t*score = t*x/(t^2+s^2); backward dt<0

    g = 1.0
    for i in range(len(t) - 1):
        dt = t[i + 1] - t[i]
        a1 = -a(t[i])
        if t[i + 1] > 0:
            a2 = -a(t[i + 1])
            g *= 1.0 + 0.5 * dt * (a1 + a2 * (1.0 + dt * a1))
        else:
            g *= 1.0 + dt * a1
    return g
