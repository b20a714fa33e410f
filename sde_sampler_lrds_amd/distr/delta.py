"""Dirac prior (mirror of ``sde_sampler/distr/delta.py:8-31``): a Gauss with a tiny scale whose ``sample``
repeats ``loc``."""
from __future__ import annotations

import torch

from .gauss import Gauss


class Delta(Gauss):
    def __init__(self, dim=1, loc: torch.Tensor | float = 0.0, approx_scale=1e-3, domain_scale=10, **kwargs):
        super().__init__(dim=dim, loc=loc, scale=approx_scale, domain_scale=domain_scale, **kwargs)

    def sample(self, shape=None):
        return self.loc.repeat(*(shape or ()), 1)
