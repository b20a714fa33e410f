"""2-D rings target (mirror of ``sde_sampler/distr/rings.py:38-121``; the sample-based diagnostics of :123-148 -- mode counts, entropy,
KL of the mode weights -- belong to the reference's eval/ layer and are not on this path): a Gaussian mixture over the radius times a
uniform angle.  Host-side torch methods only; the simulate path reads (radiuses, mixture probs, scale) through
``engine.dist_desc`` and evaluates log-density and score in HIP."""
from __future__ import annotations

import math

import torch

from .base import Distribution
from .gauss import score_mog


class Rings(Distribution):
    def __init__(self, dim: int = 2, lower_rad: float = 1.0, upper_rad: float = 5.0, num_rad: int = 3, scale: float = 0.1,
                 equilibrated: bool = False, n_reference_samples: int = int(1e6), domain_tol: float = 5.0, **kwargs):
        if dim != 2:
            raise ValueError("The rings should be two-dimensional.")
        super().__init__(dim=dim, log_norm_const=0.0, n_reference_samples=n_reference_samples, **kwargs)
        self.n_mixtures = num_rad
        self.radiuses = torch.linspace(lower_rad, upper_rad, self.n_mixtures)
        weights = torch.ones((self.n_mixtures,)) if equilibrated else self.radiuses / self.radiuses.sum()
        self.radius_dist = torch.distributions.MixtureSameFamily(
            mixture_distribution=torch.distributions.Categorical(weights),
            component_distribution=torch.distributions.Normal(loc=self.radiuses, scale=scale))
        self.angle_dist = torch.distributions.Uniform(low=0.0, high=2 * torch.pi)
        self.domain_tol = domain_tol
        if self.domain is None:
            ext = upper_rad + self.domain_tol * scale
            self.set_domain(torch.tensor([[-ext, ext], [-ext, ext]], dtype=torch.float))

    @staticmethod
    def _polar_to_xy(r, theta):
        return torch.stack([r * torch.cos(theta), r * torch.sin(theta)], dim=-1)

    def sample(self, shape):
        return self._polar_to_xy(self.radius_dist.sample(shape), self.angle_dist.sample(shape))

    def sample_init_points(self, n_points_per_mode):
        r = self.radius_dist.component_distribution.sample((n_points_per_mode,)).flatten()
        return self._polar_to_xy(r, self.angle_dist.sample((r.shape[0],)))

    def unnorm_log_prob(self, value):
        r = torch.linalg.norm(value, dim=-1)
        theta = torch.atan2(value[..., 1], value[..., 0])
        theta = theta + (theta < 0).type_as(value) * (2 * torch.pi)
        rd = self.radius_dist
        comp = torch.distributions.Normal(rd.component_distribution.loc.to(value.device), rd.component_distribution.scale.to(value.device))
        mix = torch.log_softmax(rd.mixture_distribution.logits.to(value.device), dim=-1)
        lr = torch.logsumexp(comp.log_prob(r.unsqueeze(-1)) + mix, dim=-1)
        return (lr - math.log(2 * math.pi) - torch.log(r)).view((-1, 1))

    def score_radius(self, x):
        rd = self.radius_dist
        return score_mog(x, weights=rd.mixture_distribution.probs.to(x.device),
                         means=rd.component_distribution.loc.unsqueeze(-1).to(x.device),
                         variances=rd.component_distribution.variance.unsqueeze(-1).to(x.device))

    def score(self, x, eps=1e-7, **kwargs):
        norm_x = torch.linalg.norm(x, dim=-1, keepdim=True) + eps
        return x * ((self.score_radius(norm_x) / norm_x) - (1.0 / torch.square(norm_x)))
