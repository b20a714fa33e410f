"""Gaussian family (mirror of ``sde_sampler/distr/gauss.py``): GMM :138, TwoModes :422, ManyModes :569,
Gauss :597, GaussFull :632, IsotropicGauss :720.  Parameters are stored under the reference's buffer
names (``loc``, ``scale``, ``mixture_weights``, ``cov``, ``prec``) so the engine's descriptor compiler
treats reference objects and these mirrors alike."""
from __future__ import annotations

import math
from numbers import Number

import torch
from torch import distributions as D

from .base import Distribution


def log_prob_gaussian(x, mean, variance):
    """[B,K] component log-densities, diagonal covariances (reference :67-73)."""
    quad = torch.sum(torch.square(x.unsqueeze(1) - mean.unsqueeze(0)) / variance.unsqueeze(0), dim=-1)
    return -0.5 * quad - 0.5 * mean.shape[-1] * math.log(2.0 * math.pi) - 0.5 * torch.log(variance).sum(dim=-1).unsqueeze(0)


def score_mog(x, weights, means, variances):
    """Mixture score (reference :97-107; the caller's weights are left untouched here)."""
    w = weights / weights.sum()
    resp = torch.softmax(torch.log(w.unsqueeze(0)) + log_prob_gaussian(x, means, variances), dim=-1)
    return -torch.sum(resp.unsqueeze(-1) * (x.unsqueeze(1) - means.unsqueeze(0)) / variances.unsqueeze(0), dim=1)


def score_gauss(x, means, variances):
    return -(x - means) / variances


def log_prob_gaussian_full(x, means, covariances, precisions=None, covariances_log_det=None, return_precision_times_diff=False):
    """distr/gauss.py:75-94 (host-side torch; the step loop uses the HIP kernel)."""
    diff = x.unsqueeze(1) - means.unsqueeze(0)
    if precisions is None:
        ptd = torch.linalg.solve(covariances.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)
    else:
        ptd = torch.matmul(precisions.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)
    lp = -0.5 * torch.sum(diff * ptd, dim=-1) - 0.5 * means.shape[-1] * math.log(2.0 * math.pi)
    lp = lp - 0.5 * (torch.logdet(covariances) if covariances_log_det is None else covariances_log_det).unsqueeze(0)
    return (lp, ptd) if return_precision_times_diff else lp


def score_gauss_full(x, means, covariances, precisions=None):
    """distr/gauss.py:129-135."""
    diff = x - means.unsqueeze(0)
    if precisions is None:
        return -torch.linalg.solve(covariances.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)
    return -torch.matmul(precisions.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)


def score_mog_full(x, weights, means, covariances, precisions=None, covariances_log_det=None):
    """distr/gauss.py:110-121."""
    weights = weights / weights.sum()
    lp, ptd = log_prob_gaussian_full(x, means, covariances, precisions, covariances_log_det, return_precision_times_diff=True)
    p = torch.softmax(torch.log(weights.unsqueeze(0)) + lp, dim=-1)
    return -torch.sum(p.unsqueeze(-1) * ptd, dim=1)


class GMMFull(Distribution):
    """Mixture of full-covariance Gaussians (distr/gauss.py GMMFull): what a full-covariance reference is at t = 0.  Only the
    log-density and score are needed on the simulate path (terminal cost); they are host-side torch."""

    def __init__(self, dim, loc, cov=None, prec=None, cov_log_det=None, mixture_weights=None, **kwargs):
        super().__init__(dim=dim, log_norm_const=0.0, **{k: v for k, v in kwargs.items() if k in ("n_reference_samples",)})
        self.register_buffer("loc", loc, persistent=False)
        if cov is not None:
            self.register_buffer("cov", cov, persistent=False)
            prec, cov_log_det = torch.linalg.inv(cov), torch.logdet(cov)
        self.register_buffer("prec", prec, persistent=False)
        self.register_buffer("cov_log_det", cov_log_det, persistent=False)
        w = mixture_weights if mixture_weights is not None else torch.ones(loc.shape[0])
        self.register_buffer("mixture_weights", w / w.sum(), persistent=False)

    def unnorm_log_prob(self, x):
        lp = log_prob_gaussian_full(x, self.loc, None, precisions=self.prec, covariances_log_det=self.cov_log_det)
        return torch.logsumexp(torch.log(self.mixture_weights).unsqueeze(0) + lp, dim=-1, keepdim=True)

    def score(self, x, *args, **kwargs):
        return score_mog_full(x, self.mixture_weights, self.loc, None, precisions=self.prec, covariances_log_det=self.cov_log_det)

    def sample(self, shape=None):
        comp = D.Categorical(self.mixture_weights).sample(torch.Size(shape or ()))
        cov = getattr(self, "cov", None)
        tril = torch.linalg.cholesky(cov if cov is not None else torch.linalg.inv(self.prec))
        z = torch.randn(*comp.shape, self.dim, device=self.loc.device)
        return self.loc[comp] + torch.matmul(tril[comp], z.unsqueeze(-1)).squeeze(-1)


class TwoModesFull(GMMFull):
    """(2/3) N(-a 1, C) + (1/3) N(+a 1, C) with one full covariance C = Q diag(0.05 logspace) Q^T, Q from the QR factorisation of a
    seeded random matrix (reference :469-505).  A target for the solvers whose control does not evaluate the target score inside
    the step loop (RDS with a ClippedCtrl): its log-density enters the terminal cost only."""

    def __init__(self, dim=2, a=1.0, centered=False, ill_conditioned="medium", rand_factor=5.0, seed_q=42, **kwargs):
        assert ill_conditioned in ["medium", "hard"]
        loc = torch.stack([-a * torch.ones((dim,)), a * torch.ones((dim,))])
        if centered:
            loc += (a / 3.0) * torch.ones((dim,))
        q = torch.linalg.qr(rand_factor * torch.rand((dim, dim), generator=torch.Generator().manual_seed(seed_q)), mode="complete").Q
        diag = 0.05 * torch.logspace(-2.0 if ill_conditioned == "hard" else -1.0, 0.0, dim)
        cov = torch.matmul(q, torch.matmul(torch.diag(diag), q.T))
        super().__init__(dim=dim, loc=loc, cov=torch.stack([cov, cov.clone()]), mixture_weights=torch.FloatTensor([2.0, 1.0]), **kwargs)


class GMM(Distribution):
    def __init__(self, dim=2, loc=None, scale=None, mixture_weights=None, n_reference_samples=int(1e7), name=None,
                 domain_scale=5, domain_tol=1e-5, **kwargs):
        super().__init__(dim=dim, log_norm_const=0.0, n_reference_samples=n_reference_samples, **kwargs)
        if name is not None:
            raise NotImplementedError("named mixtures (gmm_params) are not part of the engine")
        self.n_mixtures = loc.shape[0]
        if not (loc.shape == scale.shape == (self.n_mixtures, self.dim)):
            raise ValueError("Shape missmatch between loc and scale.")
        if mixture_weights is None and self.n_mixtures > 1:
            raise ValueError("Require mixture weights.")
        if not (mixture_weights is None or mixture_weights.shape == (self.n_mixtures,)):
            raise ValueError("Shape missmatch for the mixture weights.")
        self.register_buffer("loc", loc, persistent=False)
        self.register_buffer("scale", scale, persistent=False)
        self.register_buffer("mixture_weights", mixture_weights, persistent=False)
        if self.domain is None:
            mean, std = self._moments()
            self.set_domain(torch.stack([mean - domain_scale * std, mean + domain_scale * std], dim=1))

    def _torch_distr(self):
        if self.mixture_weights is None:
            return D.Independent(D.Normal(self.loc.squeeze(0), self.scale.squeeze(0)), 1)
        return D.MixtureSameFamily(D.Categorical(self.mixture_weights), D.Independent(D.Normal(self.loc, self.scale), 1))

    @property
    def distr(self):
        return self._torch_distr()

    def _moments(self):
        dist = self._torch_distr()
        return dist.mean, dist.stddev

    @property
    def stddevs(self):
        return self._torch_distr().variance.sqrt()

    def unnorm_log_prob(self, x):
        return self._torch_distr().log_prob(x).unsqueeze(-1)

    def sample(self, shape=None):
        return self._torch_distr().sample(torch.Size(shape or ()))

    def score(self, x, *args, **kwargs):
        return score_mog(x, self.mixture_weights, self.loc, torch.square(self.scale))

    def has_entropy(self):
        return self.n_mixtures > 1


class TwoModes(GMM):
    """(2/3) N(-a 1, C) + (1/3) N(+a 1, C), diagonal C (reference :422-466)."""

    def __init__(self, dim=2, a=1.0, centered=False, ill_conditioned="not", **kwargs):
        assert ill_conditioned in ["not", "medium", "hard"]
        loc = torch.stack([-a * torch.ones((dim,)), a * torch.ones((dim,))])
        if centered:
            loc += (a / 3.0) * torch.ones((dim,))
        if ill_conditioned == "not":
            scale = torch.sqrt(0.05 * torch.ones_like(loc))
        else:
            lo = -1 if ill_conditioned == "medium" else -2.0
            scale = torch.sqrt(0.05 * torch.logspace(lo, 0.0, dim)).unsqueeze(0).expand(2, -1)
        super().__init__(dim=dim, loc=loc, scale=scale, mixture_weights=torch.FloatTensor([2.0, 1.0]), **kwargs)


class BracketTwoModes(GMM):
    """(2/3) N(-a 1, C_1) + (1/3) N(+a 1, C_2) with (C_1)_i = (C_2)_(dim-i) on a linear variance ladder (reference :522-553)."""

    def __init__(self, dim=2, a=0.75, equilibrated=False, var_min=0.01, var_max=0.2, **kwargs):
        loc = torch.stack([-a * torch.ones((dim,)), a * torch.ones((dim,))])
        ladder = torch.linspace(var_min, var_max, dim)
        scale = torch.sqrt(torch.stack([ladder, torch.flip(ladder, dims=(0,))]))
        weights = torch.ones((2,)) / 2.0 if equilibrated else torch.FloatTensor([2, 1]) / 2.0
        super().__init__(dim=dim, loc=loc, scale=scale, mixture_weights=weights, **kwargs)


class ManyModes(GMM):
    """n_modes isotropic components, means U[-n, n]^d from a seeded generator, weights
    logspace(0, 1, n, base=factor) (reference :569-594)."""

    def __init__(self, n_modes=3, dim=2, seed_loc=42, mixture_weight_factor=3.0, var=0.1, **kwargs):
        gen = torch.Generator()
        gen.manual_seed(seed_loc)
        weights = torch.logspace(0.0, 1.0, n_modes, base=mixture_weight_factor)
        loc = 2 * n_modes * torch.rand((n_modes, dim), generator=gen) - n_modes
        super().__init__(dim=dim, loc=loc, scale=torch.sqrt(var * torch.ones_like(loc)), mixture_weights=weights, **kwargs)


class Gauss(GMM):
    def __init__(self, dim=1, loc: torch.Tensor | Number = 0.0, scale: torch.Tensor | Number = 1.0, **kwargs):
        super().__init__(dim=dim, loc=self._prepare_input(loc, dim), scale=self._prepare_input(scale, dim), **kwargs)

    @staticmethod
    def _prepare_input(param, dim=1):
        if not isinstance(param, torch.Tensor):
            param = torch.tensor(param, dtype=torch.float)
        param = torch.atleast_2d(param)
        return param.repeat(1, dim) if param.numel() == 1 else param

    @property
    def stddevs(self):
        return self.scale.squeeze(0)

    def score(self, x, *args, **kwargs):
        return score_gauss(x, self.loc, torch.square(self.scale))


class IsotropicGauss(Gauss):
    def __init__(self, dim=1, loc: float = 0.0, scale: float = 1.0, truncate_quartile=None, **kwargs):
        if truncate_quartile is not None:
            raise NotImplementedError("truncated priors are not part of the engine")
        super().__init__(dim=dim, loc=loc, scale=scale, **kwargs)
        self.truncate_quartile = None

    def unnorm_log_prob(self, x):
        var = self.scale[0, 0] ** 2
        const = -0.5 * self.dim * (2.0 * math.pi * var).log()
        return const - 0.5 * torch.sum((x - self.loc[0, 0]) ** 2, dim=-1, keepdim=True) / var

    def score(self, x, *args, **kwargs):
        return (self.loc[0, 0] - x) / self.scale[0, 0] ** 2

    def sample(self, shape=None):
        return self.loc[0, 0] + self.scale[0, 0] * torch.randn(*(shape or ()), self.dim, device=self.domain.device)


class GaussFull(Distribution):
    def __init__(self, dim=1, loc=None, cov=None, prec=None, n_reference_samples=int(1e7), domain_scale=5,
                 domain_tol=1e-5, **kwargs):
        super().__init__(dim=dim, log_norm_const=0.0, n_reference_samples=n_reference_samples, **kwargs)
        if loc.shape != (self.dim,):
            raise ValueError("Shape missmatch with loc.")
        if cov is None and prec is None:
            raise ValueError("Either cov or prec must be set.")
        if cov is None:
            cov = torch.linalg.inv(prec)
        if prec is None:
            prec = torch.linalg.inv(cov)
        self.register_buffer("loc", loc, persistent=False)
        self.register_buffer("cov", cov, persistent=False)
        self.register_buffer("prec", prec, persistent=False)
        if self.domain is None:
            std = torch.diagonal(cov).sqrt()
            self.set_domain(torch.stack([loc - domain_scale * std, loc + domain_scale * std], dim=1))

    @property
    def distr(self):
        return D.MultivariateNormal(loc=self.loc, covariance_matrix=self.cov)

    def unnorm_log_prob(self, x):
        return self.distr.log_prob(x).unsqueeze(-1)

    def sample(self, shape=None):
        return self.distr.sample(torch.Size(shape or ()))

    def score(self, x, *args, **kwargs):
        return -torch.matmul(self.prec.unsqueeze(0), (x - self.loc.unsqueeze(0)).unsqueeze(-1)).squeeze(-1)
