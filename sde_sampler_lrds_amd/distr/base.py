"""Distribution base class (mirror of ``sde_sampler/distr/base.py:22-157``): parameter holder with a
torch implementation of log-density / score for host-side use (sampling targets, plotting, training with
autograd).  The simulate path never calls these: it reads the parameters through ``engine.dist_desc`` and
evaluates log-densities and scores in HIP (csrc/prep_kernels.hip, csrc/sim_device.hpp)."""
from __future__ import annotations

import torch


# distr/base.py:13-18
EXPECTATION_FNS = {
    "square": lambda x: (x ** 2).sum(dim=-1, keepdims=True),
    "abs": lambda x: x.abs().sum(dim=-1, keepdims=True),
    "sum": lambda x: x.sum(dim=-1, keepdims=True),
    "square_minus_sum": lambda x: (x ** 2 - x).sum(dim=-1, keepdims=True),
}


class Distribution(torch.nn.Module):
    def __init__(self, dim: int, log_norm_const: float | None = None, domain=None, n_reference_samples=None,
                 grid_points=None, **kwargs):
        super().__init__()
        self.dim = dim
        self.log_norm_const = log_norm_const
        self.n_reference_samples = n_reference_samples
        self.grid_points = grid_points
        self.expectations = {}
        self.set_domain(domain)

    def set_domain(self, d=None):
        if d is not None:
            d = d if isinstance(d, torch.Tensor) else torch.tensor(d, dtype=torch.float)
            if d.ndim == 0:
                d = torch.stack([-d, d], dim=-1)
            if d.ndim == 1:
                d = d.unsqueeze(0)
            if d.shape == (1, 2):
                d = d.repeat(self.dim, 1)
            assert d.shape == (self.dim, 2)
        self.register_buffer("domain", d, persistent=False)

    def compute_stats(self):
        """Reference statistics need 1e7 samples / quadrature (distr/base.py:60-118): out of the hot path."""

    def has_entropy(self):
        return False

    def unnorm_log_prob(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        if self.log_norm_const is None:
            raise NotImplementedError
        return self.unnorm_log_prob(x) - self.log_norm_const

    def pdf(self, x):
        return self.log_prob(x).exp()

    def unnorm_pdf(self, x):
        return self.unnorm_log_prob(x).exp()

    def score(self, x: torch.Tensor, create_graph=False) -> torch.Tensor:
        """Autograd score (the reference's default, distr/base.py:146-154)."""
        flag = x.requires_grad
        x.requires_grad_(True)
        with torch.set_grad_enabled(True):
            total = self.unnorm_log_prob(x).sum()
            out = torch.autograd.grad(total, x, create_graph=create_graph)[0]
        x.requires_grad_(flag)
        return out

    def forward(self, x):
        return self.unnorm_log_prob(x)


def sample_uniform(domain: torch.Tensor, batchsize: int = 1) -> torch.Tensor:
    lo, hi = domain[:, 0], domain[:, 1]
    return lo + torch.rand(batchsize, domain.shape[0], device=domain.device) * (hi - lo)
