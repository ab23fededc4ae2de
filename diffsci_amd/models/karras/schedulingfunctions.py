"""Scalar schedule functions s(t), sigma(t) (reference: diffsci/models/karras/schedulingfunctions.py).

These are evaluated on the host (fp32 torch-CPU scalars) when the per-step table is built; the
formulas and their evaluation order follow the reference so the table is bit-identical to the
values the reference computes.  EDM (s = 1, sigma = t) and VE (s = 1, sigma = sqrt(t)) have a
constant scaling and run through the fused stepper; VP (s(t) != 1) runs the reference's general
rhs branch one HIP launch per operation group (Scheduler.rhs)."""
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

    def pf_score_multiplier(self, t):
        raise NotImplementedError

    def pf_scale_multiplier(self, t):
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


class VPSchedulingFunctions(SchedulingFunctions):
    """schedulingfunctions.py:66-121."""
    constant_scaling_fn = False
    has_pf_score_multiplier = False
    has_pf_scale_multiplier = False

    def __init__(self, beta_data: float = 19.9, beta_min: float = 0.1):
        super().__init__()
        self.beta_data = beta_data
        self.beta_min = beta_min

    def scaling_fn(self, t):
        expoent = 0.5 * self.beta_data * t ** 2 + self.beta_min * t
        return torch.exp(-expoent / 2)

    def scaling_fn_deriv(self, t):
        expoent = 0.5 * self.beta_data * t ** 2 + self.beta_min * t
        expoent_deriv = self.beta_data * t + self.beta_min
        return -expoent_deriv / 2 * torch.exp(-expoent / 2)

    def noise_fn(self, t):
        expoent = 0.5 * self.beta_data * t ** 2 + self.beta_min * t
        return torch.sqrt(torch.exp(expoent) - 1)

    def inverse_noise_fn(self, t):
        y = torch.log(t ** 2 + 1)
        delta = self.beta_min ** 2 + 2 * self.beta_data * y
        return (-self.beta_min + torch.sqrt(delta)) / self.beta_data

    def noise_fn_deriv(self, t):
        expoent = 0.5 * self.beta_data * t ** 2 + self.beta_min * t
        expoent_deriv = self.beta_data * t + self.beta_min
        exponentiated = torch.exp(expoent)
        numerator = expoent_deriv * exponentiated
        denominator = 2 * torch.sqrt(exponentiated - 1)
        return numerator / denominator

    def pf_score_multiplier(self, t):
        return 1 / 2 * (self.beta_data * t + self.beta_min)

    def pf_scale_multiplier(self, t):
        return -1 / 2 * (self.beta_data * t + self.beta_min)


class VESchedulingFunctions(SchedulingFunctions):
    """schedulingfunctions.py:124-149."""
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


def name_to_scheduling_functions(name: str, *args, **kwargs) -> SchedulingFunctions:
    """schedulingfunctions.py:152-168."""
    if name == "EDM":
        return EDMSchedulingFunctions()
    if name == "VP":
        return VPSchedulingFunctions(*args, **kwargs)
    if name == "VE":
        return VESchedulingFunctions(*args, **kwargs)
    raise ValueError(f"Unknown scheduling function name: {name}")
