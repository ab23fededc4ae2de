"""Integrators with the reference's class names, constructor arguments and ``step`` protocol
(diffsci/models/karras/integrators.py:17-126).

``Scheduler.propagate`` recognises these built-in classes and runs them through the fused HIP
stepper (engine.py); ``step`` itself is also available and does the same arithmetic one
reference operation group at a time on drift tensors -- each group one HIP launch."""
from typing import Any

import numpy as np
import torch

from ... import ops
from ..._native import DS_IN_DRIFT, EvalCoef


def _drift_coef():
    return EvalCoef(c_out=1.0, c_skip=0.0, sigma_sq=1.0, neg_mult=1.0, neg_lang=0.0, guidance=1.0,
                    one_minus_guidance=0.0, input_kind=DS_IN_DRIFT, stochastic=0)


class Integrator(torch.nn.Module):
    stochastic = False
    need_fns = False

    def step(self, x, t, dt, rhs, noise_strength: None | Any = None):
        raise NotImplementedError


class EulerIntegrator(Integrator):
    def step(self, x, t, dt, rhs, noise_strength: None | Any = None):
        """x + dt*rhs(x, t)   (integrators.py:35)."""
        d = rhs(x, t)
        return ops.euler(x, d, _drift_coef(), float(dt), x_out=torch.empty_like(x))


class HeunIntegrator(Integrator):
    def step(self, x, t, dt, rhs, noise_strength: None | Any = None):
        """integrators.py:44-54."""
        d1 = rhs(x, t)
        k = _drift_coef()
        if (t + dt) > 0:
            xe = ops.euler(x, d1, k, float(dt), x_out=torch.empty_like(x))
            d2 = rhs(xe, t + dt)
        elif (t + dt) == 0:
            d2 = d1
        else:
            raise ValueError("t+dt < 0 is not supported")
        return ops.heun(x, d1, k, d2, k, float(dt), x_out=torch.empty_like(x))


class EulerMaruyamaIntegrator(Integrator):
    stochastic = True

    def step(self, x, t, dt, rhs, noise_strength: None | Any = None):
        """x + rhs*dt + (noise_strength(t)*randn_like(x))*sqrt(|dt|)   (integrators.py:66-69)."""
        assert noise_strength is not None
        d = rhs(x, t)
        eps = torch.randn_like(x)
        return ops.euler(x, d, _drift_coef(), float(dt), x_out=torch.empty_like(x), eps=eps,
                         noise_coef=float(noise_strength(t)),
                         sqrt_abs_dt=float(torch.sqrt(torch.abs(torch.as_tensor(dt)))))


class KarrasIntegrator(Integrator):
    stochastic = False                  # the integration step is from the ODE
    need_fns = True

    def __init__(self, s_schurn: float = 40, s_tmin: float = 0.05, s_tmax: float = 50,
                 s_noise: float = 1.003) -> None:
        super().__init__()
        self.s_schurn = s_schurn
        self.s_tmin = s_tmin
        self.s_tmax = s_tmax
        self.s_noise = s_noise

    def step(self, x, t, dt, rhs, scheduler_fns, noise_strength: None | Any = None, nsteps: int = 100):
        """integrators.py:94-113 (EDM stochastic sampler)."""
        backstep = min(self.s_schurn / nsteps, np.sqrt(2) - 1)
        if self.s_tmin is not None:
            if not self.s_tmin <= t <= self.s_tmax:
                backstep = 0
        sigma = scheduler_fns.noise_fn(t)
        sigma_noise = sigma + backstep * sigma
        t_noise = scheduler_fns.inverse_noise_fn(sigma_noise)
        scale = scheduler_fns.scaling_fn(t)
        scale_noise = scheduler_fns.scaling_fn(t_noise)
        std = scale_noise * torch.sqrt(sigma_noise ** 2 - sigma ** 2)
        ratio = float(scale_noise / scale)
        if ratio == 1.0:                                   # EDM / VE: x + (std*S_noise)*eps in one pass
            x_noise = ops.churn(x, torch.randn_like(x), float(std * self.s_noise), xhat_out=torch.empty_like(x))
        else:                                              # VP: (s(t^)/s(t))*x + (std*S_noise)*eps, integrators.py:103-104
            x_noise = ops.axpby(x.contiguous(), ratio, torch.randn_like(x), float(std * self.s_noise))
        d1 = rhs(x_noise, t_noise)
        dt_noise = (t + dt) - t_noise
        k = _drift_coef()
        x = ops.euler(x_noise, d1, k, float(dt_noise), x_out=torch.empty_like(x))
        if (t + dt) > 0:
            d2 = rhs(x, t + dt)
            x = ops.heun(x_noise, d1, k, d2, k, float(dt_noise), x_out=x)
        return x


def name_to_integrator(name: str) -> Integrator:
    if name == "euler":
        return EulerIntegrator()
    elif name == "heun":
        return HeunIntegrator()
    elif name == "euler-maruyama":
        return EulerMaruyamaIntegrator()
    elif name == "karras":
        return KarrasIntegrator()
    else:
        raise ValueError(f"Unknown integrator: {name}")
