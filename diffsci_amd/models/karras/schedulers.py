"""Scheduler / EDMScheduler with the reference's public surface
(diffsci/models/karras/schedulers.py:27-390): create_steps, propagate / propagate_backward /
propagate_forward / propagate_partial, rhs, langevin_factor, noise_injection, apply_noise, renoise,
inpaint, repaint, set/unset_temporary_integrator, the runtime knobs maximum_scale / langevin_const /
langevin_interval.

The sigma grid and every per-step scalar are computed on the CPU in fp32 with the reference's
operation sequence; all tensor-sized work goes through the HIP stepper (engine.py)."""
import functools

import torch

from ... import ops
from ..._native import DS_IN_SCORE
from . import integrators, schedulingfunctions
from .engine import ScoreFnSource, run_table
from .steptable import build_step_table, make_eval_row

_BUILTIN = (integrators.EulerIntegrator, integrators.HeunIntegrator,
            integrators.EulerMaruyamaIntegrator, integrators.KarrasIntegrator)


def _is_builtin(integrator):
    return type(integrator) in _BUILTIN


class Scheduler(torch.nn.Module):
    def __init__(self, scheduler_fns, integrator, maximum_scale, stochastic_integrator=None):
        super().__init__()
        self.scheduler_fns = scheduler_fns
        self._integrator = integrator
        self.maximum_scale = maximum_scale
        if stochastic_integrator is None:
            stochastic_integrator = integrators.EulerMaruyamaIntegrator()
        else:
            assert stochastic_integrator.stochastic is True
        self.stochastic_integrator = stochastic_integrator
        self._temporary_integrator = None
        self.langevin_const = 1.0
        self.langevin_interval = None

    # -------------------------------------------------------------- N-step loop
    @ops.device_guard
    def propagate(self, x, score_fn, nsteps: int = 100, record_history: bool = False,
                  backward: bool = True, stochastic: bool = False, eps=None):
        """schedulers.py:48-89.  ``eps`` (extension): [nsteps, *x.shape] injected noise for the
        stochastic integrators instead of device-generator draws."""
        integrator = self.integrator if not stochastic else self.stochastic_integrator
        if _is_builtin(integrator):                    # both branches of rhs (constant scaling; s(t): VP) are tabulated
            table = build_step_table(self, integrator, nsteps, backward=backward)
            src = ScoreFnSource(score_fn, x.shape[0], x)
            out = run_table(table, src, x, record_history=record_history, eps=eps)
            if record_history and not backward:
                # forward mode fills history[1:] (skip = 1, schedulers.py:63-70,82-85)
                hist = torch.zeros((nsteps + 1,) + tuple(x.shape), dtype=x.dtype, device=x.device)
                hist[1:] = out
                return hist
            return out
        return self._propagate_custom(x, score_fn, integrator, nsteps, record_history, backward,
                                      0, None)

    @ops.device_guard
    def propagate_partial(self, x, score_fn, nsteps: int = 100, initial_step: int = 0,
                          final_step: int = 100, record_history: bool = False,
                          backward: bool = True, stochastic: bool = False, eps=None):
        """schedulers.py:178-217."""
        integrator = self.integrator if not stochastic else self.stochastic_integrator
        if not backward:
            raise NotImplementedError
        if _is_builtin(integrator):                    # both branches of rhs (constant scaling; s(t): VP) are tabulated
            table = build_step_table(self, integrator, nsteps, backward=True,
                                     initial_step=initial_step, final_step=final_step)
            src = ScoreFnSource(score_fn, x.shape[0], x)
            return run_table(table, src, x, record_history=record_history, eps=eps)
        return self._propagate_custom(x, score_fn, integrator, nsteps, record_history, True,
                                      initial_step, final_step)

    # -------------------------------------------------------------- inpainting loops
    @ops.device_guard
    def inpaint(self, x, y, mask, score_fn, nsteps: int = 100, record_history: bool = False):
        """schedulers.py:91-121: one integrator step, then re-impose the known region from the
        noised original ``y`` ([nsteps+1, B, *shape], most noised last); mask [*shape], 1 = known."""
        ops.require_device(x, "x")
        mask = mask.to(x).contiguous()
        if record_history:
            history = torch.zeros((nsteps + 1,) + tuple(x.shape), dtype=x.dtype, device=x.device)
            history[0] = x
        x = ops.mask_blend(x.contiguous(), y[-1].contiguous(), mask)
        for i in range(nsteps):
            x = self.propagate_partial(x, score_fn, nsteps, i, i + 1)
            x = ops.mask_blend(x, y[-i - 2].contiguous(), mask, out=x)
            if record_history:
                history[i + 1] = x
        return history if record_history else x

    @ops.device_guard
    def repaint(self, x, y, mask, score_fn, nsteps: int = 100, rsteps: int = 10, nresamples: int = 10,
                record_history: bool = False, noise=None):
        """schedulers.py:123-164 (RePaint resampling).  ``noise`` (extension): an iterable of tensors
        shaped like x used by the renoise draws in order, instead of the device generator."""
        if not (nsteps % rsteps) == 0:
            raise ValueError("rsteps should divide nsteps")
        ops.require_device(x, "x")
        mask = mask.to(x).contiguous()
        draws = iter(noise) if noise is not None else None
        t = self.create_steps(nsteps + 1)
        if record_history:
            history = torch.zeros((int(nresamples * (nsteps / rsteps - 1)) + 2,) + tuple(x.shape),
                                  dtype=x.dtype, device=x.device)
            history[0] = x
        x = ops.mask_blend(x.contiguous(), y[-1].contiguous(), mask)
        step, fstep = 0, rsteps
        x = self.propagate_partial(x, score_fn, nsteps, step, fstep)
        step, fstep = fstep, fstep + rsteps
        level = 0
        while fstep <= nsteps:
            x = self.propagate_partial(x, score_fn, nsteps, step, fstep)
            for i in range(nresamples):
                x = ops.mask_blend(x, y[-fstep - 1].contiguous(), mask, out=x)
                if record_history:
                    history[level + i + 1] = x
                x = self.renoise(x, t[fstep], t[step], noise=None if draws is None else next(draws))
                x = self.propagate_partial(x, score_fn, nsteps, step, fstep)
            step, fstep = fstep, fstep + rsteps
            level = level + nresamples
        if not step == nsteps:
            raise ValueError('Wrong counting')
        if record_history:
            history[level + 1] = x
            return history
        return x

    def _propagate_custom(self, x, score_fn, integrator, nsteps, record_history, backward, i0, i1):
        """User-defined Integrator subclasses: the reference's own loop, step by step."""
        t = self.create_steps(nsteps + 1).to(torch.float32).cpu()
        skip = 0
        if not backward:
            t, skip = t.flip(0), 1
        dt = torch.diff(t)
        i1 = nsteps - skip if i1 is None else i1
        if record_history:
            history = torch.zeros((i1 - i0 + 1 + (skip if i0 == 0 else 0),) + tuple(x.shape),
                                  dtype=x.dtype, device=x.device)
            history[0 + skip] = x
        rhs = functools.partial(self.rhs, score_fn=score_fn, backward=backward,
                                stochastic=integrator.stochastic)
        step = integrator.step
        if integrator.need_fns:
            step = functools.partial(step, scheduler_fns=self.scheduler_fns, nsteps=nsteps)
        for i in range(i0, i1):
            x = step(x, t[i + skip], dt[i + skip], rhs, noise_strength=self.noise_injection)
            if record_history:
                history[i - i0 + 1 + skip] = x
        return history if record_history else x

    def propagate_backward(self, x, score_fn, nsteps: int = 100, record_history: bool = False,
                           stochastic: bool = False, eps=None):
        return self.propagate(x, score_fn, nsteps, record_history, backward=True,
                              stochastic=stochastic, eps=eps)

    def propagate_forward(self, x, score_fn, nsteps: int = 100, record_history: bool = False,
                          stochastic: bool = False, eps=None):
        return self.propagate(x, score_fn, nsteps, record_history, backward=False,
                              stochastic=stochastic, eps=eps)

    # -------------------------------------------------------------- scalar functions of t
    def langevin_factor(self, t, type: str = 'const'):
        """schedulers.py:219-240."""
        standard_factor = (self.scheduler_fns.scaling_fn(t) ** 2 *
                           self.scheduler_fns.noise_fn_deriv(t) *
                           self.scheduler_fns.noise_fn(t))
        if type != 'const':
            raise NotImplementedError
        if self.langevin_interval is not None:
            t_ = t[0] if len(t.shape) > 0 else t
            if t_ > self.langevin_interval[0] and t_ < self.langevin_interval[1]:
                return self.langevin_const * standard_factor + 0 * t
            return 0 * t
        return self.langevin_const * standard_factor + 0 * t

    def noise_injection(self, t):
        """schedulers.py:242-245."""
        return torch.sqrt(2 * self.langevin_factor(t))

    @ops.device_guard
    def rhs(self, x, ti, score_fn, backward: bool = True, stochastic: bool = False):
        """Drift of the backward ODE/SDE at time ti (schedulers.py:247-274): one score_fn call,
        one HIP pass.  ti: python float or 0-dim tensor (host)."""
        ops.require_device(x, "x")
        tt = torch.as_tensor(ti, dtype=torch.float32).cpu().reshape(())
        fns = self.scheduler_fns
        if fns.constant_scaling_fn:
            row = make_eval_row(tt, self, stochastic=stochastic, backward=backward)
            sigma = torch.full((x.shape[0],), row.sigma, dtype=torch.float32, device=x.device)
            s = score_fn(x, sigma)
            return ops.drift(None, s.contiguous(), row.coef(DS_IN_SCORE))
        # non-constant scaling (VP), schedulers.py:275-293: scalars on the host in the reference's order
        sig = fns.noise_fn(tt)
        s = fns.scaling_fn(tt)
        scale_multiplier = fns.scaling_fn_deriv(tt) / s
        if fns.has_pf_score_multiplier:
            multiplier = fns.pf_score_multiplier(tt)
        else:
            multiplier = s * (fns.noise_fn_deriv(tt) * fns.noise_fn(tt))
        sigma = torch.full((x.shape[0],), float(sig), dtype=torch.float32, device=x.device)
        score = score_fn(ops.div_scalar(x.contiguous(), float(s)), sigma).contiguous()
        res = ops.axpby(x.contiguous(), float(scale_multiplier), score, -float(multiplier))
        if stochastic:
            k = self.langevin_factor(tt) * 1 / s                      # -(lambda * 1/s * score), sign flipped forward
            res = ops.axpby(res, 1.0, score, -float(k) if backward else float(k), out=res)
        return res

    # -------------------------------------------------------------- noise application
    def create_steps(self, n: int):
        raise NotImplementedError

    @ops.device_guard
    def apply_noise(self, x, nsteps: int = 100, step: int = 0):
        """x_noised = s*x + s*sigma*noise (schedulers.py:327-340); EDM: s = 1."""
        if step > nsteps:
            raise ValueError("Step larger than num of steps:{step}>{nsteps}")
        t = self.create_steps(nsteps + 1)
        t_step = t[step]
        sigma = self.scheduler_fns.noise_fn(t_step)
        scale = self.scheduler_fns.scaling_fn(t_step)
        noise = torch.randn(x.shape).to(x)
        if float(scale) != 1.0:
            return ops.axpby(x.contiguous(), float(scale), noise, float(scale * sigma))
        return ops.churn(x.contiguous(), noise, float(scale * sigma), xhat_out=torch.empty_like(x))

    @ops.device_guard
    def renoise(self, x, t: float, t_noise: float, noise=None):
        """schedulers.py:166-176.  noise (extension): the draw to use instead of randn_like(x)."""
        t = torch.as_tensor(t, dtype=torch.float32)
        t_noise = torch.as_tensor(t_noise, dtype=torch.float32)
        sigma = self.scheduler_fns.noise_fn(t)
        sigma_noise = self.scheduler_fns.noise_fn(t_noise)
        scale = self.scheduler_fns.scaling_fn(t)
        scale_noise = self.scheduler_fns.scaling_fn(t_noise)
        std = scale_noise * torch.sqrt(sigma_noise ** 2 - sigma ** 2)
        noise = torch.randn_like(x) if noise is None else noise.to(x).contiguous()
        if float(scale_noise / scale) != 1.0:
            return ops.axpby(x.contiguous(), float(scale_noise / scale), noise, float(std))
        return ops.churn(x.contiguous(), noise, float(std), xhat_out=torch.empty_like(x))

    # -------------------------------------------------------------- integrator selection
    def unset_temporary_integrator(self):
        self._temporary_integrator = None

    def set_temporary_integrator(self, integrator):
        if type(integrator) is str:
            integrator = integrators.name_to_integrator(integrator)
        self._temporary_integrator = integrator

    @property
    def integrator(self):
        if self._temporary_integrator is not None:
            return self._temporary_integrator
        return self._integrator


class EDMScheduler(Scheduler):
    def __init__(self, sigma_min: float = 0.002, sigma_max: float = 80.0, expoent_steps: float = 7.0,
                 scheduler_fns="EDM"):
        if type(scheduler_fns) is str:
            scheduler_fns = schedulingfunctions.name_to_scheduling_functions(scheduler_fns)
        super().__init__(scheduler_fns, integrators.HeunIntegrator(), sigma_max)
        self.register_buffer("sigma_min", torch.tensor(sigma_min))
        self.register_buffer("sigma_max", torch.tensor(sigma_max))
        self.register_buffer("expoent_steps", torch.tensor(expoent_steps))

    def create_steps(self, n: int):
        """rho-spaced noise levels followed by 0 (schedulers.py:377-385).

        Always evaluated in fp32 on the CPU -- in the reference these buffers never leave the CPU
        either (the config is not a registered submodule) -- with the reference's exact torch op
        sequence: the CPU tensor**tensor pow is position dependent in the last ulp, so any other
        formulation (numpy, scalar loops) would not reproduce the grid bit for bit."""
        rho = self.expoent_steps.detach().to("cpu", torch.float32)
        smax = self.sigma_max.detach().to("cpu", torch.float32)
        smin = self.sigma_min.detach().to("cpu", torch.float32)
        s = torch.arange(n - 1).to(rho) / (n - 2)
        start = smax ** (1 / rho)
        end = smin ** (1 / rho)
        steps = (start + s * (end - start)) ** rho
        if not self.scheduler_fns.identity_noise_fn:
            steps = self.scheduler_fns.inverse_noise_fn(steps)
        return torch.cat([steps, torch.zeros([1]).to(steps)])

    def step_from_time(self, t, n: int):
        """Integer step index (schedulers.py:387-390; n-1 where create_steps uses n-2 -- kept)."""
        rho = self.expoent_steps.detach().to("cpu", torch.float32)
        smax = self.sigma_max.detach().to("cpu", torch.float32)
        smin = self.sigma_min.detach().to("cpu", torch.float32)
        exp = 1 / rho
        t = torch.as_tensor(t).detach().to("cpu")
        step = (n - 1) * (t ** exp - smax ** exp) / (smin ** exp - smax ** exp)
        return torch.round(step).int()


class VPScheduler(Scheduler):
    """schedulers.py:393-418: t runs linearly from 1 to epsilon_min (no trailing zero)."""

    def __init__(self, epsilon_min: float = 0.001, scheduler_fns="VP", *args, **kwargs):
        if type(scheduler_fns) is str:
            scheduler_fns = schedulingfunctions.name_to_scheduling_functions(scheduler_fns, *args, **kwargs)
        sigma_max = (scheduler_fns.noise_fn(torch.ones([1])) * scheduler_fns.scaling_fn(torch.ones([1]))).item()
        super().__init__(scheduler_fns, integrators.HeunIntegrator(), sigma_max)
        self.register_buffer("epsilon_min", torch.tensor(epsilon_min))

    def create_steps(self, n: int):
        eps = self.epsilon_min.detach().to("cpu", torch.float32)
        s = torch.arange(n).to(eps) / (n - 1)
        return 1 + s * (eps - 1)

    def step_from_time(self, t, n: int):
        eps = self.epsilon_min.detach().to("cpu", torch.float32)
        step = (n - 1) * (torch.as_tensor(t).detach().cpu() - 1) / (eps - 1)
        return torch.round(step).int()


class VEScheduler(Scheduler):
    """schedulers.py:421-448: t = sigma^2, geometric from sigma_max^2 to sigma_min^2."""

    def __init__(self, sigma_min: float = 0.02, sigma_max: float = 100, scheduler_fns="VE", *args, **kwargs):
        if type(scheduler_fns) is str:
            scheduler_fns = schedulingfunctions.name_to_scheduling_functions(scheduler_fns, *args, **kwargs)
        super().__init__(scheduler_fns, integrators.HeunIntegrator(), sigma_max)
        self.register_buffer("sigma_min", torch.tensor(sigma_min))
        self.register_buffer("sigma_max", torch.tensor(sigma_max))

    def create_steps(self, n: int):
        smin = self.sigma_min.detach().to("cpu", torch.float32)
        smax = self.sigma_max.detach().to("cpu", torch.float32)
        s = torch.arange(n).to(smin) / (n - 1)
        return smax ** 2 * (smin ** 2 / smax ** 2) ** s

    def step_from_time(self, t, n: int):
        smin = self.sigma_min.detach().to("cpu", torch.float32)
        smax = self.sigma_max.detach().to("cpu", torch.float32)
        t = torch.as_tensor(t).detach().cpu()
        step = (n - 1) * (torch.log(t) - torch.log(smax ** 2)) / (torch.log(smin ** 2) - torch.log(smax ** 2))
        return torch.round(step).int()
