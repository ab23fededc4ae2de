"""KarrasModuleConfig / KarrasModule: the sampling surface of the reference
(diffsci/models/karras/karrasmodule.py:29-401, 431-455, 673-931, 1192-1234) on the HIP stepper.

Kept: ``KarrasModuleConfig.from_edm`` (+ the plain constructor), ``KarrasModule(model, config,
conditional=...)`` with ``sample``, ``propagate_white_noise``, ``propagate_toward_sample``,
``propagate_partial_toward_sample``, ``get_denoiser``, ``get_score``, ``encode`` / ``decode``
(``norm``, the EDM batch-norm map of ``has_edm_batch_norm`` -- one HIP launch each way -- and a user
autoencoder run as given), ``device``, ``.to()`` / ``.eval()``, and the model protocol
``model(c_in*x, c_noise[, y])``.  Training (loss_fn, training_step, optimisers) is outside this path.

For a HIP-native score network (``PUNetG``) the whole N-step loop -- ~85 launches per network
evaluation -- is captured once per (batch shape, nsteps, integrator, guidance) into a hipGraph
and replayed; an arbitrary ``model`` runs eagerly through the same step kernels."""
from typing import Any

import torch

from ... import ops
from ..nets import precision
from ..._native import DS_IN_NETWORK
from . import edmbatchnorm, noisesamplers, preconditioners, schedulers
from .autoregressivesample import LatentSpaceAutoregressive
from .engine import Loop, ModuleSource, PlanCache, condition_signature, model_signature
from .steptable import build_step_table


def get_minibatch_sizes(n: int, b: int) -> list[int]:
    """diffsci/utils.py:5-11."""
    return [b] * (n // b) + ([n % b] if n % b else [])


def dict_map(func, d):
    return {k: dict_map(func, v) for k, v in d.items()} if isinstance(d, dict) else func(d)


def dict_unsqueeze(d, dim):
    """diffsci/torchutils.py:75-77."""
    return dict_map(lambda x: torch.unsqueeze(x, dim), d)


def dict_to(d, device):
    """diffsci/torchutils.py:85-87."""
    return dict_map(lambda x: x.to(device), d)


_AR = ("autoregressive_loss_steps", "autoregressive_loss_diffusion_steps", "autoregressive_loss_guidance",
       "autoregressive_loss_weights", "autoregressive_loss_maximum_batch_size", "autoregressive_loss_integrator",
       "spatial_shape", "focus_radius")


class KarrasModuleConfig(object):
    # positional order and defaults of the reference (karrasmodule.py:40-94); the autoregressive_loss_* / spatial_shape /
    # focus_radius arguments configure training losses: they are stored and exported, the sampling path does not read them
    def __init__(self, preconditioner, noisesampler, noisescheduler, loss_metric="huber",
                 tag: str = "custom", has_edm_batch_norm: bool = False,
                 dynamic_loss_weight: int | None = None, extra_args: None | dict[str, Any] = None,
                 autoregressive_loss_steps: int = 1, autoregressive_loss_diffusion_steps: int = 100,
                 autoregressive_loss_guidance: float = 1.0, autoregressive_loss_weights=None,
                 autoregressive_loss_maximum_batch_size=None, autoregressive_loss_integrator=None,
                 spatial_shape=None, focus_radius=None):
        given = locals()
        self.preconditioner = preconditioner
        self.noisesampler = noisesampler
        self.noisescheduler = noisescheduler
        self.loss_metric = loss_metric
        self.tag = tag
        self.has_edm_batch_norm = has_edm_batch_norm
        self.dynamic_loss_weight = dynamic_loss_weight
        self.extra_args = dict() if extra_args is None else extra_args
        for k in _AR:
            setattr(self, k, given[k])

    @classmethod
    def from_edm(cls, sigma_data: float = 0.5, prior_mean: float = -1.2, prior_std: float = 1.2,
                 has_edm_batch_norm: bool = False, dynamic_loss_weight: int | None = None,
                 loss_metric="huber", autoregressive_loss_steps: int = 1, autoregressive_loss_diffusion_steps: int = 100,
                 autoregressive_loss_guidance: float = 1.0, autoregressive_loss_weights=None,
                 autoregressive_loss_maximum_batch_size=None, autoregressive_loss_integrator=None,
                 spatial_shape=None, focus_radius=None):
        """karrasmodule.py:96-175."""
        given = locals()
        ar = {k: given[k] for k in _AR}
        extra_args = dict(sigma_data=sigma_data, prior_mean=prior_mean, prior_std=prior_std, loss_metric=loss_metric, **ar)
        return cls(preconditioner=preconditioners.EDMPreconditioner(sigma_data=sigma_data),
                   noisesampler=noisesamplers.EDMNoiseSampler(sigma_data=sigma_data, prior_mean=prior_mean,
                                                              prior_std=prior_std),
                   noisescheduler=schedulers.EDMScheduler(),
                   loss_metric=loss_metric, tag="edm", has_edm_batch_norm=has_edm_batch_norm,
                   dynamic_loss_weight=dynamic_loss_weight, extra_args=extra_args, **ar)

    @classmethod
    def from_vp(cls, beta_data: float = 19.9, beta_min: float = 0.1, epsilon_min: float = 1e-3,
                epsilon_sampler: float = 1e-5, M: int = 1000, loss_metric="huber", autoregressive_loss_steps: int = 1,
                autoregressive_loss_diffusion_steps: int = 100, autoregressive_loss_guidance: float = 1.0,
                autoregressive_loss_weights=None, autoregressive_loss_maximum_batch_size=None,
                autoregressive_loss_integrator=None, spatial_shape=None, focus_radius=None):
        """karrasmodule.py:177-237."""
        given = locals()
        ar = {k: given[k] for k in _AR}
        noisescheduler = schedulers.VPScheduler(epsilon_min=epsilon_min, beta_data=beta_data, beta_min=beta_min)
        extra_args = dict(beta_data=beta_data, beta_min=beta_min, epsilon_min=epsilon_min,
                          epsilon_sampler=epsilon_sampler, M=M, loss_metric=loss_metric, **ar)
        return cls(preconditioner=preconditioners.VPPreconditioner(scheduler=noisescheduler, M=M),
                   noisesampler=noisesamplers.VPNoiseSampler(noise_scheduler=noisescheduler, epsilon=epsilon_sampler),
                   noisescheduler=noisescheduler, loss_metric=loss_metric, tag="vp", extra_args=extra_args, **ar)

    @classmethod
    def from_ve(cls, sigma_min: float = 0.02, sigma_max: float = 100, loss_metric="huber", autoregressive_loss_steps: int = 1,
                autoregressive_loss_diffusion_steps: int = 100, autoregressive_loss_guidance: float = 1.0,
                autoregressive_loss_weights=None, autoregressive_loss_maximum_batch_size=None,
                autoregressive_loss_integrator=None, spatial_shape=None, focus_radius=None):
        """karrasmodule.py:239-290."""
        given = locals()
        ar = {k: given[k] for k in _AR}
        extra_args = dict(sigma_min=sigma_min, sigma_max=sigma_max, loss_metric=loss_metric, **ar)
        return cls(preconditioner=preconditioners.VEPreconditioner(),
                   noisesampler=noisesamplers.VENoiseSampler(sigma_min=sigma_min, sigma_max=sigma_max),
                   noisescheduler=schedulers.VEScheduler(sigma_min=sigma_min, sigma_max=sigma_max),
                   loss_metric=loss_metric, tag="ve", extra_args=extra_args, **ar)

    @classmethod
    def conditionalSR3(cls, sigma_min: float = 0.02, sigma_max: float = 100, loss_metric="huber",
                       autoregressive_loss_steps: int = 1, autoregressive_loss_diffusion_steps: int = 100,
                       autoregressive_loss_guidance: float = 1.0, autoregressive_loss_weights=None,
                       autoregressive_loss_maximum_batch_size=None, autoregressive_loss_integrator=None,
                       spatial_shape=None, focus_radius=None):
        """karrasmodule.py:292-340.  The reference's own constructor raises (it hands sigma_min / sigma_max to EDMNoiseSampler,
        which takes neither, noisesamplers.py:21-24); the same TypeError is raised here rather than inventing a sampler."""
        raise TypeError("EDMNoiseSampler.__init__() got an unexpected keyword argument 'sigma_min'")

    def export_description(self) -> dict[str, Any]:
        return dict(tag=self.tag, extra_args=self.extra_args)

    @classmethod
    def load_from_description_with_tag(cls, description: dict[str, Any]):
        """karrasmodule.py:348-366: the inverse of export_description for the tagged constructors."""
        tag = description["tag"]
        extra_args = description["extra_args"]
        if tag == "custom":
            raise ValueError("Cannot load from a custom tag")
        if tag == "edm":
            return cls.from_edm(**extra_args)
        if tag == "vp":
            return cls.from_vp(**extra_args)
        if tag == "ve":
            return cls.from_ve(**extra_args)
        if tag == "conditionalSR3":
            raise NotImplementedError("conditionalSR3 is broken in the reference itself (it passes sigma_min / sigma_max to "
                                      "EDMNoiseSampler, karrasmodule.py:310-313) and is not reproduced")
        raise ValueError(f"Unknown tag: {tag}")

    @property
    def has_dynamic_loss_weight(self):
        return self.dynamic_loss_weight is not None


class KarrasModule(torch.nn.Module, LatentSpaceAutoregressive):
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict=None, **kwargs):
        """karrasmodule.py:410-429 + Lightning's loader: a ``.ckpt`` written by the reference's Trainer is a dict whose
        "state_dict" holds this module's keys ("model.*", "edm_batch_norm.*", "autoencoder.*").  The constructor arguments come
        from the caller (``model=``, ``config=``, ...; the reference saves no hyper-parameters, scripts/testing/*.py pass them),
        merged over any "hyper_parameters" the file carries; ``model`` and ``config`` are deep-copied as in the reference.
        ``hparams_file`` is accepted for signature compatibility and must be None (nothing in this path reads one)."""
        import copy
        if hparams_file is not None:
            raise NotImplementedError("hparams_file: the reference's modules save no hyper-parameters to read back")
        ckpt = torch.load(checkpoint_path, map_location="cpu" if map_location is None else map_location, weights_only=False)
        if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
            raise KeyError(f"{checkpoint_path}: not a Lightning checkpoint (no 'state_dict' entry)")
        init = dict(ckpt.get("hyper_parameters") or {})
        init.update(kwargs)
        for k in ("model", "config"):
            if init.get(k) is not None:
                init[k] = copy.deepcopy(init[k])
        module = cls(**init)
        module.load_state_dict(ckpt["state_dict"], strict=True if strict is None else bool(strict))
        return module

    def __init__(self, model: torch.nn.Module, config: KarrasModuleConfig, conditional: bool = False,
                 masked: bool = False, autoencoder: None | torch.nn.Module = None,
                 autoencoder_conditional: bool = False, encode_y: bool = False,
                 decode_original_y: bool = False):
        super().__init__()
        self.model = model
        self.config = config
        self.conditional = conditional
        self.masked = masked
        # latent models (karrasmodule.py:443-450): the autoencoder is a user module, run as given (its encode /
        # decode are ordinary torch code on whatever device it lives on); the loop between them is the HIP path
        self.autoencoder = autoencoder
        self.autoencoder_conditional = autoencoder_conditional
        self.encode_y = encode_y
        self.decode_original_y = decode_original_y
        self.norm = 1.0
        # karrasmodule.py:1236-1241
        self.edm_batch_norm = (edmbatchnorm.DimensionAgnosticBatchNorm(sigma=config.extra_args.get("sigma_data", 0.5))
                               if config.has_edm_batch_norm else None)
        self.use_graph = True          # capture the loop as a hipGraph for HIP-native networks
        # Opt-in: also capture runs whose network is evaluated as given (an `extra_residual` user module, a torch network of the
        # user's) with torch.cuda.CUDAGraph.  The user's modules must then be capture-safe (no host reads, no data-dependent
        # control flow): torch.cuda.graph's contract.  Off: such runs are launched step by step, as the reference does.
        self.capture_eager = False
        # (first element, total elements) of this process' rows inside a batch sampled by several ranks: set by
        # parallel.sample_sharded around a run so that the in-kernel noise of the stochastic integrators is the unsharded run's
        self.noise_shard = None
        self._plans = PlanCache()

    # ---------------------------------------------------------------- bookkeeping
    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    @property
    def latent_model(self):
        return self.autoencoder is not None

    def _apply(self, fn, *a, **k):
        self._plans.clear()
        return super()._apply(fn, *a, **k)

    def export_description(self) -> dict[str, Any]:
        """karrasmodule.py:462-474."""
        return dict(config_description=self.config.export_description(), conditional=self.conditional,
                    masked=self.masked, autoencoder=True if self.autoencoder else False,
                    autoencoder_conditional=self.autoencoder_conditional, encode_y=self.encode_y)

    def start_edm_batch_norm(self):
        """karrasmodule.py:1236-1241 (already done by the constructor; kept for callers that re-initialise it)."""
        self.edm_batch_norm = (edmbatchnorm.DimensionAgnosticBatchNorm(sigma=self.config.extra_args.get("sigma_data", 0.5))
                               if self.config.has_edm_batch_norm else None)

    def freeze_autoencoder(self):
        """karrasmodule.py: the autoencoder takes no gradients (a no-op for sampling, kept for drop-in scripts)."""
        if self.autoencoder is not None:
            for p in self.autoencoder.parameters():
                p.requires_grad = False

    # ---------------------------------------------------------------- denoiser / score (public API)
    def _coefs(self, sigma):
        """Per-sample preconditioner values on the host, in the reference's fp32 op order."""
        s = sigma.detach().to("cpu", torch.float32).reshape(-1)
        p = self.config.preconditioner
        return p.skip_scaling(s), p.output_scaling(s), p.input_scaling(s), p.noise_conditioner(s)

    def _run_model(self, x, sigma, y, guidance):
        """scaled_input = c_in*x; F = model(scaled_input, c_noise[, y]) (karrasmodule.py:690-716)."""
        ops.require_device(x, "x")
        x = x.contiguous()
        c_skip, c_out, c_in, c_noise = self._coefs(sigma)
        if c_in.numel() not in (1, x.shape[0]):
            raise ValueError("sigma must have one entry per sample")
        c_skip, c_out, c_in, c_noise = (c.expand(x.shape[0]).contiguous() for c in (c_skip, c_out, c_in, c_noise))
        xin = torch.empty_like(x)
        if bool((c_in == c_in[0]).all()):
            ops.scale(x, float(c_in[0]), out=xin)
        else:
            for b in range(x.shape[0]):
                ops.scale(x[b], float(c_in[b]), out=xin[b])
        cn = c_noise.to(x.device)
        fu = None
        if self.conditional and guidance != 0.0:
            f = self.model(xin, cn, y)
            if guidance != 1.0:
                fu = self.model(xin, cn)
        else:
            f = self.model(xin, cn)
        return x, f.contiguous(), (None if fu is None else fu.contiguous()), c_skip, c_out, cn

    @ops.device_guard
    def get_denoiser(self, x, sigma, y=None, guidance: float = 1.0):
        """karrasmodule.py:673-719.  sigma: [B] (values may differ per sample)."""
        x, f, fu, c_skip, c_out, cn = self._run_model(x, sigma, y, guidance)
        D = ops.denoiser(x, f, c_out.to(x.device), c_skip.to(x.device), fu=fu, guidance=guidance)
        return D, cn

    @ops.device_guard
    def get_score(self, x, sigma, y=None, guidance: float = 1.0):
        """(D - x)/sigma^2, karrasmodule.py:721-733 -- fused with the denoiser in one pass."""
        from ..._native import EvalCoef
        x, f, fu, c_skip, c_out, _ = self._run_model(x, sigma, y, guidance)
        s2 = (sigma.detach().to("cpu", torch.float32).reshape(-1) ** 2).expand(x.shape[0])
        out = torch.empty_like(x)

        def coef(b):
            return EvalCoef(c_out=float(c_out[b]), c_skip=float(c_skip[b]), sigma_sq=float(s2[b]), neg_mult=0.0,
                            neg_lang=0.0, guidance=float(guidance), one_minus_guidance=float(1 - guidance),
                            input_kind=DS_IN_NETWORK, stochastic=0)
        if bool((s2 == s2[0]).all()):
            ops.score(x, f, coef(0), fu=fu, out=out)
        else:
            for b in range(x.shape[0]):
                ops.score(x[b], f[b], coef(b), fu=None if fu is None else fu[b], out=out[b])
        return out

    # ---------------------------------------------------------------- sampling
    def sample(self, nsamples: int, shape: list[int], y=None, guidance: float = 1.0, nsteps: int = 100,
               record_history: bool = False, maximum_batch_size: None | int = None,
               integrator=None, move_to_cpu: bool = False, is_latent_shape: bool = False,
               squeeze_memory_efficiency: bool = False, return_in_latent_space: bool = False):
        """karrasmodule.py:801-865: CPU-generator white noise, optional minibatching.  Latent models given a
        data-space `shape` encode a dummy batch to learn the latent shape and redraw the noise there, as the
        reference does (karrasmodule.py:842-851)."""
        with torch.inference_mode():
            if maximum_batch_size is not None:
                result = [self.sample(b, shape, y, guidance, nsteps, record_history, maximum_batch_size=None,
                                      integrator=integrator, move_to_cpu=move_to_cpu, is_latent_shape=is_latent_shape,
                                      squeeze_memory_efficiency=squeeze_memory_efficiency,
                                      return_in_latent_space=return_in_latent_space)
                          for b in get_minibatch_sizes(nsamples, maximum_batch_size)]
                return torch.cat(result, dim=1 if record_history else 0)
            white_noise = torch.randn(*([nsamples] + list(shape))).to(self.device)
            if y is not None:
                y = dict_to(y, self.device)
            original_y = None
            if self.latent_model and not is_latent_shape:
                if self.encode_y:
                    if self.decode_original_y:
                        original_y = y.copy()
                    white_noise, y = self.encode(white_noise, y)
                    y['y'] = y['y'].squeeze(0)
                else:
                    white_noise = self.encode(white_noise, y)
                white_noise = torch.randn_like(white_noise)
            return self.propagate_white_noise(white_noise, y, guidance, nsteps, record_history,
                                              integrator=integrator,
                                              original_y=original_y if self.decode_original_y else None,
                                              move_to_cpu=move_to_cpu, latent_shape=is_latent_shape,
                                              squeeze_memory_efficiency=squeeze_memory_efficiency,
                                              return_in_latent_space=return_in_latent_space)

    def sample_and_filter(self, nsamples: int, shape: list[int], filter_fn, y=None, guidance: float = 1.0,
                          nsteps: int = 100, record_history: bool = False, maximum_batch_size: None | int = None,
                          integrator=None, move_to_cpu: bool = False, return_only_positives: bool = False):
        """karrasmodule.py:735-799: sample, then filter_fn(encode(samples)) -> {'samples', 'filter', 'hit_rate'}."""
        if record_history:
            raise ValueError("record_history is not supported for filtering at the moment")
        if maximum_batch_size is not None:
            samples, filters, num_positive = [], [], 0
            for b in get_minibatch_sizes(nsamples, maximum_batch_size):
                r = self.sample_and_filter(b, shape, filter_fn, y, guidance, nsteps, record_history,
                                           maximum_batch_size=None, integrator=integrator,
                                           return_only_positives=return_only_positives, move_to_cpu=move_to_cpu)
                samples.append(r["samples"])
                filters.append(r["filter"])
                num_positive += r["filter"].sum().item()
            return dict(samples=torch.cat(samples, dim=0), filter=torch.cat(filters, dim=0),
                        hit_rate=num_positive / nsamples)
        samples = self.sample(nsamples, shape, y=y, guidance=guidance, nsteps=nsteps, record_history=record_history,
                              maximum_batch_size=maximum_batch_size, integrator=integrator, move_to_cpu=False)
        with torch.inference_mode():
            filt = filter_fn(self.encode(samples, y, record_history))
        if return_only_positives:
            samples = samples[filt]
            filt = filt[filt]
        if move_to_cpu:
            samples = samples.detach().cpu()
        return dict(samples=samples, filter=filt, hit_rate=filt.sum() / nsamples)

    def propagate_white_noise(self, x, y=None, guidance: float = 1.0, nsteps: int = 100,
                              record_history: bool = False, integrator=None, original_y=None,
                              move_to_cpu: bool = False, latent_shape: bool = False,
                              squeeze_memory_efficiency: bool = False, return_in_latent_space: bool = False,
                              eps=None):
        """karrasmodule.py:867-905: x*maximum_scale, the N-step loop, decode (unless return_in_latent_space)."""
        with torch.inference_mode():
            result = self.propagate_toward_sample(x, y, guidance, nsteps, record_history,
                                                  integrator=integrator, eps=eps,
                                                  _scale=self.config.noisescheduler.maximum_scale)
            if not return_in_latent_space:
                result = self.decode(result, original_y if original_y is not None else y, record_history)
        if move_to_cpu:
            result = result.detach().cpu()
        return result

    def propagate_toward_sample(self, x, y=None, guidance: float = 1.0, nsteps: int = 100,
                                record_history: bool = False, integrator=None, eps=None, _scale=None):
        """karrasmodule.py:907-931.  eps (extension): injected per-step noise [nsteps, *x.shape]."""
        return self._propagate(x, y, guidance, nsteps, record_history, integrator, eps, _scale, 0, None)

    def propagate_partial_toward_sample(self, x, initial_step: int, final_step: int = None, y=None,
                                        nsteps: int = 100, record_history: bool = False, integrator=None,
                                        analytical_score=None, interp_fn=None, guidance: float = 1.0, eps=None):
        """karrasmodule.py:933-976.  interp_fn / analytical_score: the score handed to the integrator is
        alpha * trained + (1 - alpha) * analytical_score(x.cpu(), sigma.cpu()) with alpha = interp_fn(sigma), evaluated step by
        step exactly as the reference does (the analytic score is host code).  guidance, eps: extensions (keyword)."""
        final_step = nsteps if final_step is None else final_step
        if interp_fn is None:
            return self._propagate(x, y, guidance, nsteps, record_history, integrator, eps, None,
                                   initial_step, final_step)
        if analytical_score is None:
            raise AssertionError("interp_fn needs analytical_score")
        yy = None if y is None else dict_unsqueeze(y, 0)

        def rhs(xx, sigma):
            trained = self.get_score(xx, sigma, yy, guidance)
            alpha = interp_fn(sigma).unsqueeze(-1).to(trained.device).to(trained.dtype)
            analytic = analytical_score(xx.cpu().detach(), sigma.cpu().detach()).to(trained).contiguous()
            al = alpha.reshape(-1)
            out = torch.empty_like(trained)
            if bool((al == al[0]).all()):
                return ops.axpby(trained, float(al[0]), analytic, 1.0 - float(al[0]), out=out)
            if al.numel() != trained.shape[0]:
                raise ValueError("interp_fn must return one weight per sample")
            for b in range(trained.shape[0]):            # per-sample weights: one launch per sample (a diagnostic path)
                ops.axpby(trained[b], float(al[b]), analytic[b], 1.0 - float(al[b]), out=out[b])
            return out
        sch = self.config.noisescheduler
        if integrator is not None:
            sch.set_temporary_integrator(integrator)
        try:
            return sch.propagate_partial(x, rhs, nsteps, initial_step, final_step, record_history=record_history)
        finally:
            if integrator is not None:
                sch.unset_temporary_integrator()

    @ops.device_guard                              # launches go to x's GPU whatever the caller's current device is
    @torch.inference_mode()                        # sampling never differentiates; plan buffers live in one mode
    def _propagate(self, x, y, guidance, nsteps, record_history, integrator, eps, scale, i0, i1):
        ops.require_device(x, "x")
        sch = self.config.noisescheduler
        if y is not None:
            y = dict_unsqueeze(y, 0)                                     # karrasmodule.py:916-917
        if integrator is not None:
            sch.set_temporary_integrator(integrator)
        try:
            integ = sch.integrator
            if not schedulers._is_builtin(integ):      # user Integrator subclasses: the reference's own loop, step by step
                def rhs(xx, sigma):
                    return self.get_score(xx, sigma, y, guidance)
                xs = x if scale is None else ops.scale(x.contiguous(), scale)
                return sch._propagate_custom(xs, rhs, integ, nsteps, record_history, True, i0, i1)
            table = build_step_table(sch, integ, nsteps, backward=True, initial_step=i0, final_step=i1,
                                     preconditioner=self.config.preconditioner)
        finally:
            if integrator is not None:
                sch.unset_temporary_integrator()
        injected = eps is not None

        checked = [False]

        def run():
            src = ModuleSource(self, y, guidance, x.shape[0], x)
            checked[0] = src.nonfinite_word is not None and len(table.rows) > 0      # the last step kernel looks at the result

            def make_loop():
                return Loop(table, src, x, record_history, injected_noise=injected, noise_shard=self.noise_shard)

            if self.use_graph and x.is_cuda and (src.planned or self.capture_eager):
                key = (src.planned, src.batched_cfg, tuple(x.shape), str(x.device), nsteps, i0, i1, record_history, table.kind, injected,
                       (integ.s_schurn, integ.s_tmin, integ.s_tmax, integ.s_noise) if table.kind == "karras" else None,
                       float(guidance), condition_signature(y), float(sch.langevin_const), repr(sch.langevin_interval), self.noise_shard,
                       table.digest(), model_signature(self.model))
                src.static_condition = not src.planned         # evaluated as given: the captured calls read a plan-owned condition
                return self._plans.run(key, make_loop, x, y=y, scale=scale, eps=eps, torch_graph=not src.planned)
            loop = make_loop()
            loop.load(x, scale)
            loop.set_noise(eps)
            loop.launch()
            return loop.result()

        out = run()
        if precision.needs_escalation(self.model, out, x, result_checked=checked[0]):     # an activation left the fp16x3 range: see nets/precision.py
            precision.escalate(self.model)
            out = run()
        return out

    # ---------------------------------------------------------------- encode / decode (non-latent)
    # ---------------------------------------------------------------- inpainting / interpolation
    def _score_rhs(self, y):
        def rhs(x, sigma):
            with torch.inference_mode():
                return self.get_score(x, sigma, y)
        return rhs

    def propagate_toward_noise(self, x, y=None, nsteps: int = 100, record_history: bool = False,
                               stochastic_integration: bool = False, eps=None):
        """karrasmodule.py:1093-1115: forward propagation (data -> noise) of the probability-flow ODE or,
        with stochastic_integration, of the forward SDE (Euler-Maruyama).  eps: see Scheduler.propagate."""
        if y is not None:
            y = dict_unsqueeze(y, 0)
        with torch.inference_mode():
            return self.config.noisescheduler.propagate_forward(x, self._score_rhs(y), nsteps,
                                                                record_history=record_history,
                                                                stochastic=stochastic_integration, eps=eps)

    def propagate_inpaint_toward_sample(self, x, x_inpaint, mask, y=None, record_history: bool = False):
        """karrasmodule.py:1043-1066."""
        if y is not None:
            y = dict_unsqueeze(y, 0)
        with torch.inference_mode():
            return self.config.noisescheduler.inpaint(x, x_inpaint, mask, self._score_rhs(y),
                                                      x_inpaint.shape[0] - 1, record_history=record_history)

    def propagate_repaint_toward_sample(self, x, x_inpaint, mask, y=None, record_history: bool = False, noise=None):
        """karrasmodule.py:1068-1091."""
        if y is not None:
            y = dict_unsqueeze(y, 0)
        with torch.inference_mode():
            return self.config.noisescheduler.repaint(x, x_inpaint, mask, self._score_rhs(y),
                                                      x_inpaint.shape[0] - 1, record_history=record_history,
                                                      noise=noise)

    def inpaint(self, x_orig, mask, y=None, nsteps: int = 100, record_history: bool = False,
                maximum_batch_size: None | int = None, mode: str = "inpaint"):
        """karrasmodule.py:978-1026: noise the original forward (SDE), start from fresh noise, denoise
        while re-imposing the known region at every noise level."""
        if maximum_batch_size is not None:
            batch_sizes = get_minibatch_sizes(x_orig.shape[0], maximum_batch_size)
            x_chunks, m_chunks = x_orig.chunk(len(batch_sizes)), mask.chunk(len(batch_sizes))
            result = [self.inpaint(x_chunks[i], m_chunks[i], y, nsteps, record_history, maximum_batch_size=None)
                      for i, _ in enumerate(batch_sizes)]
            return torch.cat(result, dim=1 if record_history else 0)
        x_orig_history = self.propagate_toward_noise(x_orig, nsteps=nsteps, y=y, record_history=True,
                                                     stochastic_integration=True)
        noise = ops.scale(torch.randn_like(x_orig), self.config.noisescheduler.maximum_scale)
        fn = self.propagate_inpaint_toward_sample if mode == "inpaint" else self.propagate_repaint_toward_sample
        return fn(noise, x_orig_history, mask, y=y, record_history=record_history)

    def repaint(self, x_orig, mask, y=None, nsteps: int = 100, record_history: bool = False,
                maximum_batch_size: None | int = None):
        """karrasmodule.py:1028-1041."""
        return self.inpaint(x_orig, mask, y, nsteps, record_history, maximum_batch_size, mode="repaint")

    def interpolate_images(self, x1, x2, ninterp: int, jitter: None | float = 1e-2, y=None, nsteps: int = 100,
                           record_history: bool = False):
        """karrasmodule.py:1117-1144: noise both images with the deterministic forward ODE, interpolate
        linearly in noise space, denoise."""
        x = torch.stack([x1, x2], dim=0).contiguous()
        if jitter is not None:
            x = ops.churn(x, torch.randn_like(x), float(jitter), xhat_out=torch.empty_like(x))   # x + jitter*eps
        x_noised = self.propagate_toward_noise(x, y, nsteps)            # unsqueezes y itself
        x_interp = ops.lerp_stack(x_noised[0].contiguous(), x_noised[1].contiguous(), ninterp)
        return self.propagate_toward_sample(x_interp, y=y, nsteps=nsteps, record_history=record_history)

    def encode(self, x, y=None, record_history=False):
        """karrasmodule.py:1192-1214: [autoencoder.encode] -> [edm_batch_norm.normalize] -> / norm."""
        if record_history:
            return torch.stack([self.encode(xx, y, record_history=False) for xx in x], dim=0)
        if self.latent_model:
            if self.autoencoder_conditional:
                if self.encode_y:
                    x, y = self.autoencoder.encode(x, y)
                else:
                    x = self.autoencoder.encode(x, y)
            else:
                x = self.autoencoder.encode(x)
        if self.edm_batch_norm is not None:
            x = self.edm_batch_norm.normalize(x)
        x = x if self.norm == 1.0 else ops.div_scalar(x.contiguous(), self.norm)
        return (x, y) if self.encode_y else x

    def decode(self, x, y=None, record_history=False):
        """karrasmodule.py:1216-1234: * norm (1.0 -> identity) -> [edm_batch_norm.unnormalize] -> [autoencoder.decode]."""
        if record_history:
            return torch.stack([self.decode(xx, y, record_history=False) for xx in x], dim=0)
        x = x if self.norm == 1.0 else ops.scale(x.contiguous(), self.norm)
        if self.edm_batch_norm is not None:
            x = self.edm_batch_norm.unnormalize(x)
        if self.latent_model:
            x = self.autoencoder.decode(x, y) if self.autoencoder_conditional else self.autoencoder.decode(x)
        return x
