"""Training-time noise prior; constructed by KarrasModuleConfig.from_edm but never used while
sampling (reference: diffsci/models/karras/noisesamplers.py:20-41).  Kept so that a config
built with the reference's factory has the same attributes."""
import torch


class NoiseSampler(torch.nn.Module):
    def sample(self, shape):
        raise NotImplementedError

    def loss_weighting(self, sigma):
        raise NotImplementedError


class EDMNoiseSampler(NoiseSampler):
    def __init__(self, sigma_data: float = 0.5, prior_mean: float = -1.2, prior_std: float = 1.2):
        super().__init__()
        self.register_buffer("sigma_data", torch.tensor(sigma_data))
        self.register_buffer("prior_mean", torch.tensor(prior_mean))
        self.register_buffer("prior_std", torch.tensor(prior_std))

    def sample(self, shape):
        return torch.exp(self.prior_mean + self.prior_std * torch.randn(shape))

    def loss_weighting(self, sigma):
        return (sigma ** 2 + self.sigma_data ** 2) / ((sigma * self.sigma_data) ** 2)


class VPNoiseSampler(NoiseSampler):
    """noisesamplers.py:44-63."""

    def __init__(self, noise_scheduler, epsilon: float = 1e-3):
        super().__init__()
        self.noise_scheduler = noise_scheduler
        self.register_buffer("epsilon", torch.tensor(epsilon))

    def loss_weighting(self, sigma):
        return 1 / (sigma ** 2)

    def sample(self, shape):
        t = torch.rand(shape).to(self.epsilon)
        t = t * (1 - self.epsilon) + self.epsilon
        return self.noise_scheduler.scheduler_fns.noise_fn(t)


class VENoiseSampler(NoiseSampler):
    """noisesamplers.py:66-87."""

    def __init__(self, sigma_min: float = 0.02, sigma_max: float = 100):
        super().__init__()
        self.register_buffer("sigma_min", torch.tensor(sigma_min))
        self.register_buffer("sigma_max", torch.tensor(sigma_max))

    def loss_weighting(self, sigma):
        return 1 / (sigma ** 2)

    def sample(self, shape):
        unif = torch.rand(shape).to(self.sigma_min.device)
        logsigma_min, logsigma_max = torch.log(self.sigma_min), torch.log(self.sigma_max)
        return torch.exp(logsigma_min + unif * (logsigma_max - logsigma_min))


class UniformNoiseSampler(NoiseSampler):
    """noisesamplers.py:90-111: sigma ~ U(t, T); EDM loss weighting."""

    def __init__(self, t: float = 0.0, T: float = 1.0, sigma_data: float = 0.5):
        super().__init__()
        self.register_buffer("t", torch.tensor(t))
        self.register_buffer("T", torch.tensor(T))
        self.register_buffer("sigma_data", torch.tensor(sigma_data))

    def loss_weighting(self, sigma):
        return (sigma ** 2 + self.sigma_data ** 2) / ((sigma * self.sigma_data) ** 2)

    def sample(self, shape):
        sigma = torch.rand(shape).to(self.t.device)
        return self.t + sigma * (self.T - self.t)
