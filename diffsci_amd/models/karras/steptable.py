"""Host-built per-step tables for the Karras stepper.

During one sampling run every sample of the batch sits at the same noise level
(reference schedulers.py:254: ``t = ti*torch.ones(B)``), so the noise levels, step sizes,
preconditioner values and drift multipliers are per-step scalars.  They are computed here once,
on the CPU, in fp32, with the reference's own torch operation sequence (same ops, same order,
0-dim tensors), so the numbers handed to the kernels are bit-identical to the ones the
reference's CPU path computes.  Nothing here touches the GPU.

Both branches of Scheduler.rhs are tabulated: constant scaling (EDM / VE, schedulers.py:259-274) and a scaling function
s(t) (VP, schedulers.py:275-293: rows carry s, s'/s and the multiplier s*sigma'*sigma, and the kernels divide the state by s
on the way into the network).
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from ..._native import DS_IN_NETWORK, EvalCoef


@dataclass
class EvalRow:
    """One score evaluation at noise level t (0-dim fp32 CPU tensor)."""
    t: torch.Tensor
    sigma: float = 0.0
    sigma_sq: float = 0.0
    neg_mult: float = 0.0
    neg_lang: float = 0.0
    stochastic: bool = False
    # preconditioner values (filled when a preconditioner is given)
    c_skip: float = 0.0
    c_out: float = 1.0
    c_in: float = 1.0
    c_noise: float = 0.0
    # non-constant scaling (VP): x~ = x / scale goes into the score, d = scale_mult*x + neg_mult*score
    scaled: bool = False
    scale: float = 1.0
    scale_mult: float = 0.0

    def coef(self, input_kind=DS_IN_NETWORK, guidance=1.0, next_scale=1.0, xin_copies=1, nonfinite=None):
        """next_scale: s at the evaluation the emitted network input feeds (xin = c_in * (x / s)); xin_copies: 2 when that
        evaluation is a batched-guidance one reading the input twice ([2B, ...]); nonfinite: int32 [1] device word the step
        kernel taking this struct raises when its x_out holds inf / NaN (the run's last step only)."""
        return EvalCoef(c_out=self.c_out, c_skip=self.c_skip, sigma_sq=self.sigma_sq,
                        neg_mult=self.neg_mult, neg_lang=self.neg_lang, guidance=float(guidance),
                        one_minus_guidance=float(1 - guidance), input_kind=int(input_kind),
                        stochastic=int(self.stochastic), scaled=int(self.scaled), scale=self.scale,
                        scale_mult=self.scale_mult, next_scale=float(next_scale), xin_copies=int(xin_copies),
                        nonfinite=None if nonfinite is None else nonfinite.data_ptr())


@dataclass
class StepRow:
    """One integrator step: up to two evaluations plus the scalars of the update."""
    first: EvalRow
    second: Optional[EvalRow]
    dt: float                      # step used by the update (dt, or dt_hat for the churn sampler)
    churn_coef: Optional[float] = None   # std*s_noise (KarrasIntegrator) -- None for other integrators
    churn_ratio: float = 1.0       # s(t_hat)/s(t) (integrators.py:103): 1 under constant scaling
    noise_coef: float = 0.0        # sqrt(2*langevin) (Euler-Maruyama)
    sqrt_abs_dt: float = 0.0


@dataclass
class StepTable:
    kind: str                      # "euler" | "heun" | "euler-maruyama" | "karras"
    t: torch.Tensor                # the fp32 grid, nsteps+1 values
    rows: List[StepRow] = field(default_factory=list)

    @property
    def evals(self):
        out = []
        for r in self.rows:
            out.append(r.first)
            if r.second is not None:
                out.append(r.second)
        return out

    @property
    def needs_noise(self):
        return self.kind in ("euler-maruyama", "karras")

    def digest(self):
        """Every scalar a captured launch sequence bakes in as a kernel argument -- part of the plan key, so a scheduler,
        preconditioner or integrator changed IN PLACE between two runs re-captures instead of replaying stale coefficients."""
        def ev(e):
            return None if e is None else (e.sigma, e.sigma_sq, e.neg_mult, e.neg_lang, e.stochastic, e.c_skip, e.c_out, e.c_in,
                                           e.c_noise, e.scaled, e.scale, e.scale_mult)
        return (self.kind,) + tuple((ev(r.first), ev(r.second), r.dt, r.churn_coef, r.churn_ratio, r.noise_coef, r.sqrt_abs_dt)
                                    for r in self.rows)


def _f(x):
    return float(x)


def make_eval_row(t, scheduler, stochastic=False, backward=True, preconditioner=None):
    """Scalars of Scheduler.rhs at time t (schedulers.py:254-274) and of the preconditioner
    (karrasmodule.py:690-704) -- all as fp32 0-dim tensor arithmetic in the reference's order."""
    fns = scheduler.scheduler_fns
    sigma = fns.noise_fn(t)
    if fns.constant_scaling_fn:
        sigma_deriv = fns.noise_fn_deriv(t)
        if getattr(fns, "has_pf_score_multiplier", False):
            multiplier = fns.pf_score_multiplier(t)
        else:
            multiplier = sigma * sigma_deriv
        row = EvalRow(t=t, sigma=_f(sigma), sigma_sq=_f(sigma ** 2), neg_mult=_f(-multiplier))
        if stochastic:
            lang = scheduler.langevin_factor(t)
            row.stochastic = True
            row.neg_lang = _f(-lang) if backward else _f(lang)     # schedulers.py:269-274
    else:                                                          # schedulers.py:275-293, the same scalar operations in order
        s = fns.scaling_fn(t)
        scale_multiplier = fns.scaling_fn_deriv(t) / s
        if getattr(fns, "has_pf_score_multiplier", False):
            multiplier = fns.pf_score_multiplier(t)
        else:
            multiplier = s * (fns.noise_fn_deriv(t) * fns.noise_fn(t))
        row = EvalRow(t=t, sigma=_f(sigma), sigma_sq=_f(sigma ** 2), neg_mult=_f(-multiplier), scaled=True, scale=_f(s),
                      scale_mult=_f(scale_multiplier))
        if stochastic:
            k = scheduler.langevin_factor(t) * 1 / s               # -(langevin * 1/s * score); sign flipped forward
            row.stochastic = True
            row.neg_lang = _f(-k) if backward else _f(k)
    if preconditioner is not None:
        row.c_skip = _f(preconditioner.skip_scaling(sigma))
        row.c_out = _f(preconditioner.output_scaling(sigma))
        row.c_in = _f(preconditioner.input_scaling(sigma))
        row.c_noise = _f(preconditioner.noise_conditioner(sigma))
    return row


def build_step_table(scheduler, integrator, nsteps, backward=True, initial_step=0, final_step=None,
                     preconditioner=None):
    """Tabulate steps [initial_step, final_step) of Scheduler.propagate (schedulers.py:60-85)."""
    from . import integrators as I
    t = scheduler.create_steps(nsteps + 1).to(torch.float32).cpu()
    skip = 0
    if not backward:
        t = t.flip(0)
        skip = 1
    dt = torch.diff(t)
    if final_step is None:
        final_step = nsteps - skip
    if isinstance(integrator, I.KarrasIntegrator):
        kind = "karras"
    elif isinstance(integrator, I.HeunIntegrator):
        kind = "heun"
    elif isinstance(integrator, I.EulerMaruyamaIntegrator):
        kind = "euler-maruyama"
    elif isinstance(integrator, I.EulerIntegrator):
        kind = "euler"
    else:
        raise TypeError("not a built-in integrator")
    stochastic = bool(integrator.stochastic)
    table = StepTable(kind=kind, t=t)

    def ev(tt):
        return make_eval_row(tt, scheduler, stochastic=stochastic, backward=backward,
                             preconditioner=preconditioner)

    for i in range(initial_step, final_step):
        ti, dti = t[i + skip], dt[i + skip]
        if kind == "euler":
            table.rows.append(StepRow(ev(ti), None, _f(dti)))
        elif kind == "euler-maruyama":
            table.rows.append(StepRow(ev(ti), None, _f(dti),
                                      noise_coef=_f(scheduler.noise_injection(ti)),
                                      sqrt_abs_dt=_f(torch.sqrt(torch.abs(dti)))))
        elif kind == "heun":
            t2 = ti + dti                                       # integrators.py:45-47: fl(t+dt), not t[i+1]
            if t2 > 0:
                table.rows.append(StepRow(ev(ti), ev(t2), _f(dti)))
            elif t2 == 0:
                table.rows.append(StepRow(ev(ti), None, _f(dti)))   # d2 = d1  =>  x + (0.5*(d1+d1))*dt
            else:
                raise ValueError("t+dt < 0 is not supported")
        else:                                                   # integrators.py:94-112
            back = min(integrator.s_schurn / nsteps, math.sqrt(2) - 1)
            if integrator.s_tmin is not None:
                if not integrator.s_tmin <= ti <= integrator.s_tmax:
                    back = 0
            fns = scheduler.scheduler_fns
            sigma = fns.noise_fn(ti)
            sigma_hat = sigma + back * sigma
            t_hat = fns.inverse_noise_fn(sigma_hat)
            scale = fns.scaling_fn(ti)
            scale_hat = fns.scaling_fn(t_hat)
            std = scale_hat * torch.sqrt(sigma_hat ** 2 - sigma ** 2)
            t2 = ti + dti
            dt_hat = t2 - t_hat
            table.rows.append(StepRow(ev(t_hat), ev(t2) if t2 > 0 else None, _f(dt_hat),
                                      churn_coef=_f(std * integrator.s_noise), churn_ratio=_f(scale_hat / scale)))
    return table
