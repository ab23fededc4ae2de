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
