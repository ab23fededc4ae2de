"""Scalar schedule functions s(t), sigma(t) (reference: diffsci/models/karras/schedulingfunctions.py).

These are evaluated on the host (fp32 torch-CPU scalars) when the per-step table is built; the
formulas and their evaluation order follow the reference so the table is bit-identical to the
values the reference computes.  Only the EDM family (s = 1, sigma = t) is on the HIP path; the
VP / VE families are the next scope row."""
import torch


class SchedulingFunctions(torch.nn.Module):
    constant_scaling_fn = False
    identity_noise_fn = False
    has_pf_score_multiplier = False
    has_pf_scale_multiplier = False

    def scaling_fn(self, t):
        raise NotImplementedError

    def scaling_fn_deriv(self, t):
        raise NotImplementedError

    def noise_fn(self, t):
        raise NotImplementedError

    def inverse_noise_fn(self, t):
        raise NotImplementedError

    def noise_fn_deriv(self, t):
        raise NotImplementedError


class EDMSchedulingFunctions(SchedulingFunctions):
    """schedulingfunctions.py:41-63."""
    constant_scaling_fn = True
    identity_noise_fn = True

    def scaling_fn(self, t):
        return 1 + 0 * t

    def scaling_fn_deriv(self, t):
        return 0 * t

    def noise_fn(self, t):
        return 1 * t

    def inverse_noise_fn(self, t):
        return 1 * t

    def noise_fn_deriv(self, t):
        return 1 + 0 * t


def name_to_scheduling_functions(name: str, *args, **kwargs) -> SchedulingFunctions:
    if name == "EDM":
        return EDMSchedulingFunctions()
    if name in ("VP", "VE"):
        raise NotImplementedError(f"{name} scheduling functions are not on the HIP path yet (EDM only)")
    raise ValueError(f"Unknown scheduling functions: {name}")
