"""Karras preconditioners (reference: diffsci/models/karras/preconditioners.py).

The c_* functions are scalar formulas of sigma.  They are evaluated on the host in fp32 with
the reference's operation order when the step table is built (uniform sigma per step), or on
host copies of a per-sample sigma vector for the public get_denoiser API."""
import torch


class KarrasPreconditioner(torch.nn.Module):
    def skip_scaling(self, sigma):
        raise NotImplementedError

    def output_scaling(self, sigma):
        raise NotImplementedError

    def input_scaling(self, sigma):
        raise NotImplementedError

    def noise_conditioner(self, sigma):
        raise NotImplementedError


class EDMPreconditioner(KarrasPreconditioner):
    """preconditioners.py:30-53.  Note c_noise = 0.5*ln(sigma), as in the reference."""

    def __init__(self, sigma_data: float = 0.5):
        super().__init__()
        self.register_buffer("sigma_data", torch.tensor(sigma_data))

    def skip_scaling(self, sigma):
        return self.sigma_data ** 2 / (sigma ** 2 + self.sigma_data ** 2)

    def output_scaling(self, sigma):
        return sigma * self.sigma_data / torch.sqrt(sigma ** 2 + self.sigma_data ** 2)

    def input_scaling(self, sigma):
        return 1 / torch.sqrt(sigma ** 2 + self.sigma_data ** 2)

    def noise_conditioner(self, sigma):
        return 0.5 * torch.log(sigma)


class VPPreconditioner(KarrasPreconditioner):
    """preconditioners.py:56-83."""

    def __init__(self, scheduler, M: int = 1000):
        super().__init__()
        self.scheduler = scheduler
        self.M = M

    def skip_scaling(self, sigma):
        return 1 + 0.0 * sigma

    def output_scaling(self, sigma):
        return -sigma

    def input_scaling(self, sigma):
        return 1 / torch.sqrt(sigma ** 2 + 1.0)

    def noise_conditioner(self, sigma):
        finv = self.scheduler.scheduler_fns.inverse_noise_fn
        return (self.M - 1) * finv(sigma)


class VEPreconditioner(KarrasPreconditioner):
    """preconditioners.py:86-105."""

    def __init__(self):
        super().__init__()

    def skip_scaling(self, sigma):
        return 1 + 0.0 * sigma

    def output_scaling(self, sigma):
        return sigma

    def input_scaling(self, sigma):
        return 1 + 0.0 * sigma

    def noise_conditioner(self, sigma):
        return torch.log(0.5 * sigma)


class NullPreconditioner(KarrasPreconditioner):
    """preconditioners.py:139-161: D = F(x, sigma)."""

    def __init__(self):
        super().__init__()

    def skip_scaling(self, sigma):
        return 0.0 * sigma

    def output_scaling(self, sigma):
        return 1.0 + 0.0 * sigma

    def input_scaling(self, sigma):
        return 1.0 + 0.0 * sigma

    def noise_conditioner(self, sigma):
        return sigma


class SR3Preconditioner(KarrasPreconditioner):
    """preconditioners.py:108-136."""

    def __init__(self, sigma_data: float = 0.5):
        super().__init__()
        self.register_buffer("sigma_data", torch.tensor(sigma_data))

    def skip_scaling(self, sigma):
        return self.sigma_data ** 2 / (2 * (sigma ** 2 + self.sigma_data ** 2))

    def output_scaling(self, sigma):
        return sigma * self.sigma_data / (2 * torch.sqrt(sigma ** 2 + self.sigma_data ** 2))

    def input_scaling(self, sigma):
        return 1 / torch.sqrt(sigma ** 2 + self.sigma_data ** 2)

    def noise_conditioner(self, sigma):
        return 0.5 * torch.log(sigma)
