"""phi^4 lattice field (mirror of ``sde_sampler/distr/phi_four.py:8-96``; 1-D, Dirichlet-0 boundary)."""
from __future__ import annotations

import torch

from .base import Distribution


class PhiFour(Distribution):
    def __init__(self, a, b, dim, dim_phys=1, beta=1, bc=("dirichlet", 0), tilt=None, grid_points=1024, **kwargs):
        if dim_phys != 1 or tuple(bc) != ("dirichlet", 0) or tilt is not None:
            raise NotImplementedError("the engine covers the 1-D Dirichlet-0 untilted lattice (conf/target/phi_four.yaml)")
        self.a, self.b, self.beta, self.dim_grid, self.dim_phys, self.bc, self.tilt = a, b, beta, dim, dim_phys, bc, tilt
        self.coef = a * dim
        super().__init__(dim=dim, grid_points=grid_points, **kwargs)
        self.set_domain(torch.stack([-1.5 * torch.ones((dim,)), 1.5 * torch.ones((dim,))], dim=1))

    def U(self, x):
        padded = torch.nn.functional.pad(x, (1, 1), value=0.0)
        bonds = ((padded[:, 1:] - padded[:, :-1]) ** 2 / 2).sum(1)
        well = ((1 - x ** 2) ** 2 / 4 + self.b * x).sum(1) / self.coef
        return bonds * self.coef + well

    def grad_U(self, x):
        padded = torch.nn.functional.pad(x, (1, 1), value=0.0)
        lap = 2.0 * x - padded[:, 2:] - padded[:, :-2]
        return (self.b - x * (1.0 - torch.square(x))) / self.coef + self.coef * lap

    def unnorm_log_prob(self, x, *args, **kwargs):
        return -self.beta * self.U(x).unsqueeze(-1)

    def score(self, x, *args, **kwargs):
        return -self.beta * self.grad_U(x)
