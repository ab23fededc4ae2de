"""Stochastic-interpolant / flow-matching sampler with the reference's surface
(diffsci/models/karras/flowfield.py:21-170, 233-345, 441-544, 704-795): SIScheduler, Preconditioner,
SIModuleConfig, SIModule.sample / integrate_flow_field / integration_step / get_flow_field /
get_score_field.  Sampling only (loss, optimisers, latent autoencoders, batch-norm statistics and
SIModule.inpaint are outside the HIP path and raise).

The Heun / Euler loop over ``linspace(1, 0, nsteps)`` runs on the same fused HIP stepper and hipGraph
machinery as KarrasModule: per evaluation the host tabulates (c_in, c_noise, c_out, c_skip, the
flow multiplier) in fp32 with the reference's operation order, the kernels turn network outputs into
the flow field v, apply classifier-free guidance and the update in one pass.
  precondition 'identity':  v = model(x, t, y=y)                              DS_IN_FLOW
  precondition 'edm':       v = sigma'/sigma * (x - (c_skip x + c_out F))     DS_IN_NETWORK with sigma^2 := 1
"""
import warnings
from typing import Any, Callable

import numpy as np
import torch

from ... import ops
from ..nets import precision
from ..._native import DS_IN_FLOW, DS_IN_NETWORK
from .engine import Loop, ModuleSource, PlanCache, condition_signature, model_signature
from . import edmbatchnorm
from .karrasmodule import dict_to, dict_unsqueeze
from .steptable import EvalRow, StepRow, StepTable


class SIScheduler(object):
    """flowfield.py:21-111 -- scalar functions alpha(t), sigma(t) and derivatives, evaluated on the host."""

    def __init__(self, alpha_fn, sigma_fn, alpha_fn_dot, sigma_fn_dot, sigma_fn_inv):
        self.alpha_fn = alpha_fn
        self.sigma_fn = sigma_fn
        self.alpha_fn_dot = alpha_fn_dot
        self.sigma_fn_dot = sigma_fn_dot
        self.sigma_fn_inv = sigma_fn_inv

    @classmethod
    def linear(cls):
        return cls(alpha_fn=lambda t: 1 - t, sigma_fn=lambda t: t,
                   alpha_fn_dot=lambda t: -1 * torch.ones_like(t), sigma_fn_dot=lambda t: torch.ones_like(t),
                   sigma_fn_inv=lambda s: s)

    @classmethod
    def cosine(cls):
        return cls(alpha_fn=lambda t: torch.cos(t * np.pi / 2), sigma_fn=lambda t: torch.sin(t * np.pi / 2),
                   alpha_fn_dot=lambda t: -1 * torch.pi / 2 * torch.sin(t * np.pi / 2),
                   sigma_fn_dot=lambda t: torch.pi / 2 * torch.cos(t * np.pi / 2),
                   sigma_fn_inv=lambda s: (2 / np.pi) * torch.arcsin(s))

    @classmethod
    def finterpolation(cls, f, finv, fdot, sigma_min: float, sigma_max: float):
        def sigma_fn(t):
            return f((1 - t) * finv(sigma_min) + t * finv(sigma_max))

        def sigma_fn_inv(s):
            return (finv(s) - finv(sigma_min)) / (finv(sigma_max) - finv(sigma_min))

        def sigma_fn_dot(t):
            return fdot((1 - t) * finv(sigma_min) + t * finv(sigma_max)) * (finv(sigma_max) - finv(sigma_min))
        return cls(alpha_fn=lambda t: 0.0 * t + 1.0, sigma_fn=sigma_fn, alpha_fn_dot=lambda t: 0.0 * t,
                   sigma_fn_dot=sigma_fn_dot, sigma_fn_inv=sigma_fn_inv)

    @classmethod
    def edm(cls, expoent: float = 7.0, sigma_min: float = 0.02, sigma_max: float = 80.0):
        return cls.finterpolation(lambda x: x ** expoent, lambda x: x ** (1 / expoent),
                                  lambda x: expoent * x ** (expoent - 1), sigma_min, sigma_max)

    @classmethod
    def get_interpolator(cls, name, *args, **kwargs):
        if name not in cls.named_interpolators():
            raise ValueError(f"Invalid interpolator: {name}")
        return getattr(cls, name)(*args, **kwargs)

    @classmethod
    def named_interpolators(cls):
        return ['linear', 'cosine', 'edm', 'finterpolation']


class Preconditioner(object):
    """flowfield.py:114-169.  The named parameterisations ('identity', 'edm', None) of a time-dependent network run
    on the fused stepper through eval_row; autonomous flows (model(x, y=y), no time input) and user precondition
    callables cannot be tabulated -- they are evaluated step by step (`generic`), the callable as given, the
    arithmetic around the network on the HIP elementwise kernels."""

    def __init__(self, scheduler: SIScheduler, precondition_fn='identity', is_autonomous: bool = False, **kwargs):
        if isinstance(precondition_fn, str) and precondition_fn not in ('identity', 'edm'):
            raise ValueError(f"Invalid condition function: {precondition_fn}")
        self.scheduler = scheduler
        self.precondition_fn = precondition_fn
        self.is_autonomous = is_autonomous
        self.kwargs = kwargs

    @property
    def generic(self):
        return self.is_autonomous or callable(self.precondition_fn)

    def __call__(self, model, x, t=None, y=None):
        """get_flow_field of the reference for the generic cases; t: [B] device tensor (one value)."""
        fn = self.precondition_fn
        if callable(fn):
            return fn(model, x, y=y) if self.is_autonomous else fn(model, x, t, y=y)
        if fn in (None, 'identity'):                                   # autonomous identity
            return model(x, y=y)
        # autonomous 'edm' (flowfield.py:163-164): cskip*x + cout*model(x / cin, y=y) -- as written there, without
        # the sigma'/sigma factor of the time-dependent branch
        sigma_data = self.kwargs.get("sigma_data", 0.5)
        sigma = self.scheduler.sigma_fn(t.reshape(-1)[0].cpu())
        cin = 1 / torch.sqrt(sigma_data ** 2 + sigma ** 2)
        cout = sigma * sigma_data / torch.sqrt(sigma_data ** 2 + sigma ** 2)
        cskip = sigma_data ** 2 / (sigma_data ** 2 + sigma ** 2)
        f = model(ops.div_scalar(x.contiguous(), float(cin)), y=y)
        return ops.axpby(x.contiguous(), float(cskip), f.contiguous(), float(cout))

    @property
    def kind(self):
        return 'edm' if self.precondition_fn == 'edm' else 'identity'

    def eval_row(self, t, integrate_on_sigma=False):
        """Scalars of one flow-field evaluation at time t (0-dim fp32 CPU tensor), reference order."""
        sch = self.scheduler
        sigma_dot = sch.sigma_fn_dot(t)
        if self.kind == 'identity':
            # v = F [/ sigma']: DS_IN_FLOW, d = neg_mult*(F/sigma_sq)
            return EvalRow(t=t, sigma=float(sch.sigma_fn(t)), sigma_sq=float(sigma_dot) if integrate_on_sigma else 1.0,
                           neg_mult=1.0, c_skip=0.0, c_out=1.0, c_in=1.0, c_noise=float(t))
        sigma_data = self.kwargs.get("sigma_data", 0.5)
        sigma = sch.sigma_fn(t)
        cin = 1 / torch.sqrt(sigma_data ** 2 + sigma ** 2)
        cout = sigma * sigma_data / torch.sqrt(sigma_data ** 2 + sigma ** 2)
        cskip = sigma_data ** 2 / (sigma_data ** 2 + sigma ** 2)
        cnoise = 0.5 * torch.log(sch.sigma_fn(t))
        mult = sigma_dot / sigma                                   # v = mult*(x - D) = (-mult)*((D - x)/1)
        if integrate_on_sigma:
            mult = mult / sigma_dot
        return EvalRow(t=t, sigma=float(sigma), sigma_sq=1.0, neg_mult=float(-mult), c_skip=float(cskip),
                       c_out=float(cout), c_in=float(cin), c_noise=float(cnoise))


class SIModuleConfig(torch.nn.Module):
    """flowfield.py:233-286 (sampling-side fields; the loss configuration is stored, not used)."""

    def __init__(self, scheduler: SIScheduler | str = 'linear', scheduler_args: dict[str, Any] = {},
                 num_channels: int | None = None, initial_norm: bool | float = False,
                 autonomous_flow: bool = False, precondition_fn: Callable | str | None = None,
                 loss_weighting='uniform', loss_metric='huber', autoencoder_is_conditional: bool = False,
                 encode_condition: bool = False):
        super().__init__()
        if isinstance(scheduler, str):
            scheduler = SIScheduler.get_interpolator(scheduler, **scheduler_args)
        self.scheduler = scheduler
        self.num_channels = num_channels
        self.initial_norm = initial_norm
        self.autonomous_flow = autonomous_flow
        self.loss_weighting = loss_weighting
        self.loss_metric = loss_metric
        self.precondition_fn = precondition_fn
        self.autoencoder_is_conditional = autoencoder_is_conditional
        self.encode_condition = encode_condition
        self.alpha_fn, self.sigma_fn = scheduler.alpha_fn, scheduler.sigma_fn
        self.alpha_fn_dot, self.sigma_fn_dot = scheduler.alpha_fn_dot, scheduler.sigma_fn_dot
        self.sigma_fn_inv = scheduler.sigma_fn_inv
        self.preconditioner = Preconditioner(scheduler, precondition_fn, autonomous_flow)


class _Adapter:
    """What engine.ModuleSource needs from a module: the network and whether a condition is present."""

    def __init__(self, model, conditional):
        self.model, self.conditional = model, conditional


class SIModule(torch.nn.Module):
    def __init__(self, config: SIModuleConfig, model: torch.nn.Module, autoencoder: torch.nn.Module | None = None):
        super().__init__()
        object.__setattr__(self, "config", config)
        self.model = model
        # latent models (flowfield.py:300-334): the autoencoder is a user module, run as given
        self.autoencoder = autoencoder
        if self.autoencoder:
            self.freeze_autoencoder()
        self.set_initial_norm()
        self.use_graph = True
        self.capture_eager = False      # opt-in capture of runs through a user torch network (see KarrasModule.capture_eager)
        self._plans = PlanCache()

    def freeze_autoencoder(self):
        """flowfield.py:304-310."""
        for param in self.autoencoder.parameters():
            param.requires_grad = False

    def set_initial_norm(self):
        """flowfield.py:336-345: the data <-> network-space map undone after integration."""
        n = self.config.initial_norm
        if isinstance(n, bool):
            self.initial_norm = (edmbatchnorm.DimensionAgnosticBatchNorm(self.config.num_channels) if n
                                 else edmbatchnorm.IdentityBatchNorm())
        elif isinstance(n, (float, int)):
            self.initial_norm = edmbatchnorm.ConstantBatchNorm(n)
        else:
            raise ValueError(f"Invalid initial norm: {n}")

    def encode(self, x, y=None):
        """flowfield.py:312-325."""
        if not self.autoencoder:
            return x, y
        c = self.config
        if not c.autoencoder_is_conditional and not c.encode_condition:
            x = self.autoencoder.encode(x)
        elif c.autoencoder_is_conditional and not c.encode_condition:
            x = self.autoencoder.encode(x, y)
        elif not c.autoencoder_is_conditional and c.encode_condition:
            raise ValueError("Cannot encode condition if autoencoder is not conditional")
        else:
            x, y = self.autoencoder.encode(x, y)
        if isinstance(x, dict):
            x = x['zsample']
        return x, y

    def decode(self, x, y=None):
        """flowfield.py:327-334."""
        if not self.autoencoder:
            return x, y
        x = self.autoencoder.decode(x, y) if self.config.autoencoder_is_conditional else self.autoencoder.decode(x)
        return x, y

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def _apply(self, fn, *a, **k):
        self._plans.clear()
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ fields
    def _eval_scalars(self, t):
        tt = torch.as_tensor(t, dtype=torch.float32).detach().cpu()
        if tt.numel() != 1 and not bool((tt == tt.reshape(-1)[0]).all()):
            raise NotImplementedError("per-sample times in get_flow_field (the sampler uses one time per step)")
        return tt.reshape(-1)[0].reshape(())

    @ops.device_guard
    def get_flow_field(self, x_noised, t, guidance: float = 1.0, y=None, integrate_on_sigma: bool = False):
        """flowfield.py:441-458.  t: a scalar or [B]; samples at different times are evaluated group by group."""
        ops.require_device(x_noised, "x_noised")
        tt = torch.as_tensor(t, dtype=torch.float32).detach().cpu().reshape(-1)
        if tt.numel() > 1 and bool((tt != tt[0]).any()):
            if tt.numel() != x_noised.shape[0]:
                raise ValueError("t must be a scalar or have one entry per sample")
            out = torch.empty_like(x_noised)
            for u in torch.unique(tt):
                idx = torch.nonzero(tt == u).reshape(-1).to(x_noised.device)
                out[idx] = self.get_flow_field(x_noised[idx].contiguous(), u, guidance, y, integrate_on_sigma)
            return out
        if self.config.preconditioner.generic:
            return self._generic_flow(x_noised.contiguous(), tt[0], guidance, y, integrate_on_sigma)
        row = self.config.preconditioner.eval_row(self._eval_scalars(t), integrate_on_sigma)
        src = self._source(y, guidance, x_noised)
        table = StepTable(kind="euler", t=torch.zeros(2), rows=[StepRow(row, None, 0.0)])
        src.prepare(table)
        xin = ops.scale(x_noised.contiguous(), row.c_in)
        f, fu = src.evaluate(x_noised, xin, row, 0, 0)
        return ops.drift(x_noised.contiguous(), f, row.coef(src.input_kind, src.guidance), fu=fu)

    def _generic_flow(self, x, t, guidance, y, integrate_on_sigma):
        """Autonomous flows / user precondition callables: the preconditioner is called as the reference calls it;
        guidance blend and the optional 1/sigma' on the HIP elementwise kernels (flowfield.py:449-457)."""
        pre = self.config.preconditioner
        tb = torch.full((x.shape[0],), float(t), device=x.device)
        v = pre(self.model, x, tb, y=y)
        if not (guidance == 1.0 or y is None):
            vu = pre(self.model, x, tb, y=None)
            v = ops.axpby(v.contiguous(), float(guidance), vu.contiguous(), float(1 - guidance))
        if integrate_on_sigma:
            v = ops.div_scalar(v.contiguous(), float(self.config.sigma_fn_dot(t)))
        return v

    @ops.device_guard
    def get_score_field_from_flow_field(self, flow_field, x_noised, t):
        """flowfield.py:483-501: (alpha v - alpha' x) / (sigma (alpha' sigma - alpha sigma'))."""
        tv = torch.as_tensor(t, dtype=torch.float32).detach().cpu().reshape(-1)
        if tv.numel() > 1 and bool((tv != tv[0]).any()):                 # per-sample times: group by group
            out = torch.empty_like(flow_field)
            for u in torch.unique(tv):
                idx = torch.nonzero(tv == u).reshape(-1).to(flow_field.device)
                out[idx] = self.get_score_field_from_flow_field(flow_field[idx].contiguous(), x_noised[idx].contiguous(), u)
            return out
        tt = self._eval_scalars(t)
        c = self.config
        alpha, sigma, alpha_dot, sigma_dot = c.alpha_fn(tt), c.sigma_fn(tt), c.alpha_fn_dot(tt), c.sigma_fn_dot(tt)
        den = sigma * (alpha_dot * sigma - alpha * sigma_dot)
        num = ops.axpby(flow_field.contiguous(), float(alpha), x_noised.contiguous(), -float(alpha_dot))
        return ops.div_scalar(num, float(den), out=num)

    def get_score_field(self, x_noised, t, y=None, guidance: float = 1.0, integrate_on_sigma: bool = False):
        """flowfield.py:460-481."""
        v = self.get_flow_field(x_noised, t, y=y, guidance=guidance, integrate_on_sigma=integrate_on_sigma)
        return self.get_score_field_from_flow_field(v, x_noised, t)

    # ------------------------------------------------------------------ sampling
    def _source(self, y, guidance, like):
        src = ModuleSource(_Adapter(self.model, y is not None), y, guidance, like.shape[0], like)
        # guidance == 0 with a condition: the reference forms 0*v_c + 1*v_u; ModuleSource evaluates the
        # unconditional branch alone, which is the same field
        src.input_kind = DS_IN_FLOW if self.config.preconditioner.kind == 'identity' else DS_IN_NETWORK
        return src

    def _table(self, time_schedule, integrate_on_sigma):
        ts = torch.as_tensor(time_schedule, dtype=torch.float32).detach().cpu()
        pc = self.config.preconditioner
        n = ts.numel()
        rows = []
        for i in range(n - 1):
            t_curr, t_next = ts[i], ts[i + 1]
            dt = (self.config.sigma_fn(t_next) - self.config.sigma_fn(t_curr)) if integrate_on_sigma else (t_next - t_curr)
            first = pc.eval_row(t_curr, integrate_on_sigma)
            second = None if i == n - 2 else pc.eval_row(t_next, integrate_on_sigma)   # last step: Euler (flowfield.py:724)
            rows.append(StepRow(first, second, float(dt)))
        return StepTable(kind="heun", t=ts, rows=rows)

    def sample(self, nsamples: int, shape: list[int], y=None, guidance: float = 1.0, nsteps: int = 30,
               is_latent_shape: bool = False, integrate_on_sigma: bool = False, noise_injection: bool = False,
               return_latents: bool = False, orig_noise=None):
        """flowfield.py:503-544."""
        with torch.inference_mode():
            if orig_noise is None:
                x = torch.randn(nsamples, *shape).to(self.device)
            else:
                assert orig_noise.shape[0] == nsamples, "Number of samples must match"
                assert list(orig_noise.shape[1:]) == list(shape), "Shape of noise must match"
                x = orig_noise.to(self.device)
            if y is not None:
                warnings.warn("Moving y to device: {}".format(self.device))
                y = dict_to(y, self.device)
            if not is_latent_shape and self.autoencoder:
                x, _ = self.encode(x, y)                 # only to learn the latent shape (flowfield.py:525-528)
                x = torch.randn_like(x)
            if y is not None:
                y = dict_unsqueeze(y, 0)
            time_schedule = torch.linspace(1, 0, nsteps)
            sigma_init = self.config.sigma_fn(time_schedule[0])
            x = self._integrate(x, time_schedule, y, guidance, False, integrate_on_sigma, noise_injection,
                                scale=float(sigma_init))
            if not return_latents:
                x, _ = self.decode(x, y)
            return x

    def integrate_flow_field(self, x, time_schedule, y=None, guidance: float = 1.0, return_history: bool = False,
                             integrate_on_sigma: bool = False, noise_injection: bool = False):
        """flowfield.py:704-747: Heun steps, the last one Euler; history = [(t_i, x_i)]."""
        with torch.inference_mode():
            return self._integrate(x, time_schedule, y, guidance, return_history, integrate_on_sigma, noise_injection)

    @ops.device_guard
    def _integrate(self, x, time_schedule, y, guidance, return_history, integrate_on_sigma, noise_injection, scale=None):
        ops.require_device(x, "x")
        if noise_injection:
            return self._integrate_em(x, time_schedule, y, guidance, return_history, integrate_on_sigma, scale)
        if self.config.preconditioner.generic:
            return self._integrate_generic(x, time_schedule, y, guidance, return_history, integrate_on_sigma, scale)
        table = self._table(time_schedule, integrate_on_sigma)

        checked = [False]

        def run():
            src = self._source(y, guidance, x)
            checked[0] = src.nonfinite_word is not None and len(table.rows) > 0      # the last step kernel looks at the result
            if self.use_graph and x.is_cuda and (src.planned or self.capture_eager):
                src.static_condition = not src.planned
                return self._run_planned(table, src, x, y, guidance, return_history, integrate_on_sigma, scale)
            loop = Loop(table, src, x, return_history)
            loop.load(x, scale)
            loop.launch()
            return loop.result()

        out = run()
        if precision.needs_escalation(self.model, out, x, result_checked=checked[0]):     # an activation left the fp16x3 range: nets/precision.py
            precision.escalate(self.model)
            out = run()
        if return_history:                                                        # initial_norm.unnorm, flowfield.py:742-747
            return [(table.t[i].to(x.device), self.initial_norm.unnorm(out[i])) for i in range(out.shape[0])]
        return self.initial_norm.unnorm(out)

    def _run_planned(self, table, src, x, y, guidance, return_history, integrate_on_sigma, scale):
        """Capture the whole run once per (shape, schedule, guidance, condition structure) and replay it; what depends
        on the condition's values is refreshed in plan-owned buffers before every replay (engine.PlanCache)."""
        key = (src.planned, tuple(x.shape), table.digest(), float(guidance), condition_signature(y), bool(return_history),
               bool(integrate_on_sigma), str(x.device), model_signature(self.model))
        return self._plans.run(key, lambda: Loop(table, src, x, return_history), x, y=y, scale=scale, torch_graph=not src.planned)

    def _integrate_generic(self, x, time_schedule, y, guidance, return_history, integrate_on_sigma, scale):
        """integrate_flow_field for preconditioners that cannot be tabulated: Heun steps, the last one Euler
        (flowfield.py:719-747), each through integration_step."""
        ts = torch.as_tensor(time_schedule, dtype=torch.float32).detach().cpu()
        x = ops.scale(x.contiguous(), scale) if scale is not None else x.contiguous()
        history = [(ts[0], x)] if return_history else None
        n = ts.numel()
        for i in range(n - 1):
            x = self.integration_step(x, ts[i], ts[i + 1], y, guidance, method='euler' if i == n - 2 else 'heun',
                                      integrate_on_sigma=integrate_on_sigma)
            if return_history:
                history.append((ts[i + 1], x))
        if return_history:
            return [(t, self.initial_norm.unnorm(h)) for t, h in history]
        return self.initial_norm.unnorm(x)

    def _integrate_em(self, x, time_schedule, y, guidance, return_history, integrate_on_sigma, scale):
        """Euler-Maruyama with noise injection (flowfield.py:783-793), one HIP pass per operation group."""
        ts = torch.as_tensor(time_schedule, dtype=torch.float32).detach().cpu()
        c = self.config
        x = ops.scale(x.contiguous(), scale) if scale is not None else x.contiguous()
        history = [(ts[0], x)] if return_history else None
        for i in range(ts.numel() - 1):
            t_curr, t_next = ts[i], ts[i + 1]
            dt = (c.sigma_fn(t_next) - c.sigma_fn(t_curr)) if integrate_on_sigma else (t_next - t_curr)
            v = self.get_flow_field(x, t_curr, y=y, guidance=guidance, integrate_on_sigma=integrate_on_sigma)
            score = self.get_score_field_from_flow_field(v, x, t_curr)
            omega = c.sigma_fn(t_curr)
            d = ops.axpby(v, 1.0, score, -float(0.5 * omega))            # v - 0.5*omega*score
            x = ops.axpby(x, 1.0, d, float(dt))                           # x + dt*(...)
            x = ops.axpby(x, 1.0, torch.randn_like(x), float(torch.sqrt(omega * torch.abs(dt))), out=x)
            if return_history:
                history.append((ts[i + 1], x))
        if return_history:
            return [(t, self.initial_norm.unnorm(h)) for t, h in history]
        return self.initial_norm.unnorm(x)

    @ops.device_guard
    def integration_step(self, x, t_curr, t_next, y=None, guidance: float = 1.0, method: str = 'euler',
                         integrate_on_sigma: bool = False, noise_injection: bool = False):
        """flowfield.py:749-795: one Euler / Heun / Euler-Maruyama step between two times (every sample of the
        batch at the same time, as in the reference's own loop).  The loops above run whole schedules through the fused
        stepper; this is the single-step entry point the reference exposes."""
        ops.require_device(x, "x")
        tc, tn = (torch.as_tensor(t, dtype=torch.float32).detach().cpu().reshape(-1) for t in (t_curr, t_next))
        if bool((tc != tc[0]).any()) or bool((tn != tn[0]).any()):
            raise NotImplementedError("integration_step: per-sample times (the sampler uses one time per step)")
        tc, tn = tc[0], tn[0]
        c = self.config
        dt = float((c.sigma_fn(tn) - c.sigma_fn(tc)) if integrate_on_sigma else (tn - tc))
        if method in ('euler', 'heun'):
            assert not noise_injection, "Noise injection is not supported for Euler and Heun methods"
        kw = dict(y=y, guidance=guidance, integrate_on_sigma=integrate_on_sigma)
        x = x.contiguous()
        if method == 'euler':
            return ops.axpby(x, 1.0, self.get_flow_field(x, tc, **kw), dt)
        if method == 'heun':
            v1 = self.get_flow_field(x, tc, **kw)
            v2 = self.get_flow_field(ops.axpby(x, 1.0, v1, dt), tn, **kw)
            vs = ops.add(v1, v2)                                                  # x + dt*(v1 + v2)/2
            return ops.axpby(x, 1.0, ops.div_scalar(ops.scale(vs, dt), 2.0), 1.0)
        if method == 'euler_maruyama':
            if not noise_injection:
                raise ValueError("Noise injection is required for Euler-Maruyama method")
            return self._em_step(x, tc, tn, y, guidance, integrate_on_sigma, torch.randn_like)
        raise ValueError(f"Invalid integration method: {method}")

    def inpaint(self, x_orig, mask, nsamples: int = 1, y=None, guidance: float = 1.0, nsteps: int = 30,
                integrate_on_sigma: bool = False, noise_injection: bool = False, orig_noise=None,
                mask_falloff: int = 0, resample_steps: int = 0, jump_length: int = 1, mask_start_t: float = 1.0,
                noise=None):
        """flowfield.py:546-641: Euler-Maruyama steps; after each one the known region (mask = 1) is replaced by a
        freshly noised copy of x_orig at the step's noise level, optionally with RePaint-style jumps back.
        noise (extension, for reproducibility): an iterator of the standard-normal draws the loop consumes, in the
        reference's order -- per inner iteration: the step's [B, *shape], the patch's [1, *shape], and for a jump the
        re-noising [B, *shape] and its patch [1, *shape]."""
        warnings.warn("We are assuming we are in latent space for inpainting")
        draws = iter(noise) if noise is not None else None

        def randn_like(t):
            if draws is None:
                return torch.randn_like(t)
            e = next(draws).to(t)
            if tuple(e.shape) != tuple(t.shape):
                raise ValueError(f"injected noise has shape {tuple(e.shape)}, expected {tuple(t.shape)}")
            return e.contiguous()

        with torch.inference_mode():
            if y is not None:
                warnings.warn("Moving y to device: {}".format(self.device))
                y = dict_to(y, self.device)
            x_orig = x_orig.to(self.device)
            mask = mask.to(self.device)
            shape = x_orig.shape
            soft_mask = (self._create_soft_mask(mask, mask_falloff) if mask_falloff > 0 else mask).to(torch.float32).contiguous()
            x_orig = self.initial_norm(x_orig.unsqueeze(0).contiguous())
            if orig_noise is None:
                x = torch.randn(nsamples, *shape).to(self.device)
            else:
                assert orig_noise.shape[0] == nsamples, "Number of samples must match"
                assert orig_noise.shape[1:] == shape, "Shape of noise must match"
                x = orig_noise.to(self.device)
            c = self.config
            ts = torch.linspace(1, 0, nsteps)
            x = ops.scale(x.contiguous(), float(c.sigma_fn(ts[0])))
            B = x.shape[0]

            def blend(x, t):                              # (1 - m)*x + m*(alpha(t)*x_orig + sigma(t)*eps)
                patch = ops.axpby(x_orig, float(c.alpha_fn(t)), randn_like(x_orig), float(c.sigma_fn(t)))
                return ops.mask_blend(x, patch.expand(B, *shape).contiguous(), soft_mask)

            for i in range(nsteps - 1):
                t_curr, t_next = ts[i], ts[i + 1]
                for r in range(resample_steps + 1):
                    x = self._em_step(x, t_curr, t_next, y, guidance, integrate_on_sigma, randn_like)
                    if t_next.item() <= mask_start_t:
                        x = blend(x, t_next)
                        if r < resample_steps and i + jump_length < nsteps - 1:
                            # jump back to the current level: re-noise the sample, re-impose the known region
                            x = ops.axpby(x, float(c.alpha_fn(t_curr)), randn_like(x), float(c.sigma_fn(t_curr)))
                            x = blend(x, t_curr)
            return self.initial_norm.unnorm(x)

    def _em_step(self, x, tc, tn, y, guidance, integrate_on_sigma, randn_like):
        """The Euler-Maruyama branch of integration_step (flowfield.py:783-793) with an injectable noise source."""
        c = self.config
        dt = float((c.sigma_fn(tn) - c.sigma_fn(tc)) if integrate_on_sigma else (tn - tc))
        v = self.get_flow_field(x, tc, y=y, guidance=guidance, integrate_on_sigma=integrate_on_sigma)
        score = self.get_score_field_from_flow_field(v, x, tc)
        omega = c.sigma_fn(tc)
        d = ops.axpby(v, 1.0, score, -float(0.5 * omega))
        x = ops.axpby(x, 1.0, d, dt)
        return ops.axpby(x, 1.0, randn_like(x), float(torch.sqrt(omega * abs(dt))))

    @staticmethod
    def _create_soft_mask(mask, falloff: int):
        """flowfield.py:643-702: box-filtered mask / (itself + box-filtered complement), cosine-smoothed.  One-off
        preprocessing of the mask with torch ops (2-D and 3-D masks; others are returned unchanged)."""
        if falloff <= 0:
            return mask
        import numpy as np
        import torch.nn.functional as F
        ndim = mask.dim() - 1
        m = mask.unsqueeze(0).float()
        pool = {2: F.avg_pool2d, 3: F.avg_pool3d}.get(ndim)
        if pool is None:
            return mask
        k, p = 2 * falloff + 1, falloff
        m_dilated = pool(m, kernel_size=k, stride=1, padding=p)
        m_eroded = pool(1 - m, kernel_size=k, stride=1, padding=p)
        soft = m_dilated / (m_dilated + m_eroded + 1e-8)
        soft = (1 - torch.cos(soft * np.pi)) / 2
        return soft.squeeze(0)
