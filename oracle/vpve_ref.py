"""CPU oracle for the VP / VE parameterisations (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates, in the reference's operation order,
  diffsci/models/karras/schedulingfunctions.py:66-149   VP / VE functions s(t), sigma(t), ...
  diffsci/models/karras/schedulers.py:247-294           Scheduler.rhs, both branches
  diffsci/models/karras/schedulers.py:393-448           VPScheduler / VEScheduler.create_steps
  diffsci/models/karras/preconditioners.py:56-105       VP / VE preconditioners
  diffsci/models/karras/integrators.py:29-113           the four integrators with general s(t), sigma(t)
"""
import math

import torch

from .karras_ref import _bcast


class VP:
    constant_scaling_fn = False
    has_pf_score_multiplier = False

    def __init__(self, beta_data=19.9, beta_min=0.1):
        self.beta_data, self.beta_min = beta_data, beta_min

    def _e(self, t):
        return 0.5 * self.beta_data * t ** 2 + self.beta_min * t

    def scaling_fn(self, t):
        return torch.exp(-self._e(t) / 2)

    def scaling_fn_deriv(self, t):
        expoent_deriv = self.beta_data * t + self.beta_min
        return -expoent_deriv / 2 * torch.exp(-self._e(t) / 2)

    def noise_fn(self, t):
        return torch.sqrt(torch.exp(self._e(t)) - 1)

    def inverse_noise_fn(self, t):
        y = torch.log(t ** 2 + 1)
        delta = self.beta_min ** 2 + 2 * self.beta_data * y
        return (-self.beta_min + torch.sqrt(delta)) / self.beta_data

    def noise_fn_deriv(self, t):
        expoent_deriv = self.beta_data * t + self.beta_min
        exponentiated = torch.exp(self._e(t))
        return (expoent_deriv * exponentiated) / (2 * torch.sqrt(exponentiated - 1))


class VE:
    constant_scaling_fn = True
    has_pf_score_multiplier = True

    def scaling_fn(self, t):
        return 1 + 0 * t

    def scaling_fn_deriv(self, t):
        return 0 * t

    def noise_fn(self, t):
        return torch.sqrt(t)

    def inverse_noise_fn(self, t):
        return t ** 2

    def noise_fn_deriv(self, t):
        return 0.5 / torch.sqrt(t)

    def pf_score_multiplier(self, t):
        return 0.5 + 0 * t


def vp_steps(n, epsilon_min=0.001):
    eps = torch.tensor(epsilon_min)
    s = torch.arange(n).to(eps) / (n - 1)
    return 1 + s * (eps - 1)


def ve_steps(n, sigma_min=0.02, sigma_max=100):
    smin, smax = torch.tensor(sigma_min), torch.tensor(float(sigma_max))
    s = torch.arange(n).to(smin) / (n - 1)
    return smax ** 2 * (smin ** 2 / smax ** 2) ** s


def vp_precond(fns, M=1000):
    def f(sigma):
        return 1 + 0.0 * sigma, -sigma, 1 / torch.sqrt(sigma ** 2 + 1.0), (M - 1) * fns.inverse_noise_fn(sigma)
    return f


def ve_precond(sigma):
    return 1 + 0.0 * sigma, sigma, 1 + 0.0 * sigma, torch.log(0.5 * sigma)


def langevin_factor(fns, t, langevin_const=1.0):
    return langevin_const * (fns.scaling_fn(t) ** 2 * fns.noise_fn_deriv(t) * fns.noise_fn(t)) + 0 * t


def rhs(fns, x, ti, score_fn, backward=True, stochastic=False):
    """Scheduler.rhs, schedulers.py:247-294."""
    t = ti * torch.ones(x.shape[0]).to(x)
    t_ = _bcast(t, x)
    sigma = fns.noise_fn(t)
    sigma_ = _bcast(sigma, x)
    sigma_deriv_ = _bcast(fns.noise_fn_deriv(t), x)
    if fns.constant_scaling_fn:
        multiplier = fns.pf_score_multiplier(t_) if fns.has_pf_score_multiplier else sigma_ * sigma_deriv_
        sc = score_fn(x, sigma)
        res = -multiplier * sc
        if stochastic:
            sf = -(langevin_factor(fns, t_) * sc)
            res += sf if backward else -sf
        return res
    s = fns.scaling_fn(t_)
    scale_multiplier = fns.scaling_fn_deriv(t_) / s
    multiplier = s * (fns.noise_fn_deriv(t_) * fns.noise_fn(t_))
    sc = score_fn(x / s, sigma)
    res = scale_multiplier * x - multiplier * sc
    if stochastic:
        sf = -(langevin_factor(fns, t_) * 1 / s * sc)
        res += sf if backward else -sf
    return res


def step(fns, integrator, x, t, dt, f, eps=None, nsteps=None, s_churn=40, s_tmin=0.05, s_tmax=50, s_noise=1.003):
    """integrators.py:29-113; f(x, t) is the drift."""
    if integrator == "euler":
        return x + dt * f(x, t)
    if integrator == "heun":
        d1 = f(x, t)
        if (t + dt) > 0:
            d2 = f(x + dt * d1, t + dt)
        elif (t + dt) == 0:
            d2 = d1
        else:
            raise ValueError("t+dt < 0 is not supported")
        return x + 0.5 * (d1 + d2) * dt
    if integrator == "euler-maruyama":
        return x + f(x, t) * dt + (torch.sqrt(2 * langevin_factor(fns, t)) * eps * torch.sqrt(torch.abs(dt)))
    if integrator == "karras":
        back = min(s_churn / nsteps, math.sqrt(2) - 1)
        if s_tmin is not None and not s_tmin <= t <= s_tmax:
            back = 0
        sigma = fns.noise_fn(t)
        sigma_noise = sigma + back * sigma
        t_noise = fns.inverse_noise_fn(sigma_noise)
        scale, scale_noise = fns.scaling_fn(t), fns.scaling_fn(t_noise)
        std = scale_noise * torch.sqrt(sigma_noise ** 2 - sigma ** 2)
        x_noise = (scale_noise / scale) * x + std * s_noise * eps
        d1 = f(x_noise, t_noise)
        dt_noise = (t + dt) - t_noise
        x = x_noise + dt_noise * d1
        if (t + dt) > 0:
            d2 = f(x, t + dt)
            x = x_noise + 0.5 * (d1 + d2) * dt_noise
        return x
    raise ValueError(integrator)


def propagate(fns, grid, x, score_fn, integrator="heun", backward=True, record_history=False, eps=None):
    """Scheduler.propagate, schedulers.py:48-89, on a given time grid (nsteps+1 values)."""
    nsteps = grid.numel() - 1
    t = grid.to(x)
    skip = 0
    if not backward:
        t, skip = t.flip(0), 1
    dt = torch.diff(t)
    stochastic = integrator == "euler-maruyama"

    def f(xx, tt):
        return rhs(fns, xx, tt, score_fn, backward=backward, stochastic=stochastic)
    if record_history:
        history = torch.zeros([nsteps + 1] + list(x.shape)).to(x)
        history[0 + skip] = x
    for i in range(nsteps - skip):
        x = step(fns, integrator, x, t[i + skip], dt[i + skip], f, eps=None if eps is None else eps[i].to(x),
                 nsteps=nsteps)
        if record_history:
            history[i + 1 + skip] = x
    return history if record_history else x
