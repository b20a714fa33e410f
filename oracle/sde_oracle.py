"""CPU oracle for the SDE-sampler hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (torch, CPU, fp32) of the arithmetic the
reference ``vanilladucky/sde_sampler_lrds`` performs on its Euler-Maruyama /
exponential-integrator inner loops.  It is the *checker* for the HIP engine in
``sde_sampler_lrds_amd/``: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product never does.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against fixtures under ``tests/golden/`` that were
produced by running the real reference in the build container
(``tests/golden/gen_golden.py``; the reference itself cannot travel to the GPU box).

All citations are ``path:line`` inside ``/root/reference/sde_sampler``.
Operation order follows the reference so that, given the same injected noise,
results agree with it to fp32 round-off.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np
import torch

F32 = torch.float32


def _t(v) -> torch.Tensor:
    return v if isinstance(v, torch.Tensor) else torch.tensor(v, dtype=F32)


def rowdot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return (a * b).sum(dim=-1, keepdim=True)


def clip(v: torch.Tensor, m: Optional[float]) -> torch.Tensor:
    """utils/common.py:85-112 clip_and_log (logging is commented out upstream)."""
    return v if m is None else v.clip(min=-1.0 * m, max=m)


# --------------------------------------------------------------------------- #
# time grids -- utils/common.py:18-82
# --------------------------------------------------------------------------- #
def bisect_v(f, lo, hi, target, iters):
    """utils/common.py:18-27 binary_search_v."""
    for _ in range(iters):
        mid = (lo + hi) / 2.0
        val = f(mid)
        lo = torch.where(val > target, mid, lo)
        hi = torch.where(val <= target, mid, hi)
    return (lo + hi) / 2.0


def get_timesteps(start, end, dt=None, steps=None, rescale_t=None, n_attemps=1024, sde=None):
    """utils/common.py:30-82.  Note the cosine grid has steps+2 points (quirk kept)."""
    if (steps is None) is (dt is None):
        raise ValueError("Exactly one of `dt` and `steps` should be defined.")
    if steps is None:
        steps = int(math.ceil((end - start) / dt))
    if sde is not None:
        a = sde.log_snr(start)
        b = sde.log_snr(end)
        if torch.isnan(a) or torch.isnan(b):
            raise ValueError("NaN SNR")
        grid = torch.linspace(a, b, steps=steps + 1)
        mid = bisect_v(sde.log_snr, start, end, grid[1:-1], n_attemps)
        return torch.concat([torch.FloatTensor([start]), mid, torch.FloatTensor([end])]).sort().values
    if rescale_t is None:
        return torch.linspace(start, end, steps=steps + 1)
    if rescale_t == "quad":
        return torch.sqrt(torch.linspace(start, _t(end).square(), steps=steps + 1)).clip(max=end)
    if rescale_t == "cosine":
        pre = torch.linspace(start, end, steps + 1) / end
        phase = ((pre + 0.008) / (1 + 0.008)) * torch.pi * 0.5
        dts = torch.cos(phase) ** 4
        dts /= dts.sum()
        dts *= end
        return torch.concat((torch.tensor([start]), torch.cumsum(dts, -1)))
    raise ValueError("Unkown timestep rescaling method.")


# --------------------------------------------------------------------------- #
# noising processes -- eq/sdes.py
# --------------------------------------------------------------------------- #
class OUBase:
    """eq/sdes.py:117-351 (generic linear SDE pieces used on the path)."""
    T: torch.Tensor

    def drift(self, t, x):  # :143-145
        return self.drift_coeff(t) * x

    def diff(self, t):  # :147-149
        return self.diff_coeff(t)

    def transition_params(self, s, t):  # :167-178
        mean = torch.exp(torch.log(self.s(t)) - torch.log(self.s(s)))
        var = self.s(t) ** 2 * (self.sigma_sq(t) - self.sigma_sq(s))
        return mean, var

    def marginal_diag(self, t, loc0, var0=None):
        """:208-248 marginal_params restricted to diagonal (or no) initial variance."""
        loc = self.s(t) * loc0
        var = self.s(t) ** 2 * self.sigma_sq(t)
        if var0 is not None:
            var = var + self.s(t) ** 2 * var0
        return loc, var

    def marginal_full(self, t, loc0, cov0):
        """eq/sdes.py:208-248 with full covariance matrices [K,d,d]: (s m, s^2 sigma^2 I + s^2 C)."""
        loc = self.s(t) * loc0
        var = self.s(t) ** 2 * self.sigma_sq(t)
        return loc, var * torch.eye(cov0.shape[-1]).unsqueeze(0) + self.s(t) ** 2 * cov0

    def marginal_eigen(self, t, loc0, D, P):
        """eq/sdes.py:228-238: covariances given as (D, P), C = P diag(D) P^T -> (s m, precision, log det) of the noised marginal."""
        diag = D + self.sigma_sq(t)
        prec = torch.einsum("...ik,...k,...jk->...ij", P, 1.0 / diag, P)
        prec = prec / self.s(t) ** 2
        log_det = torch.sum(torch.log(diag), dim=-1)
        log_det = log_det + 2.0 * diag.shape[-1] * torch.log(self.s(t))
        return self.s(t) * loc0, prec, log_det

    def log_snr(self, t):  # :347-351  (t may be a python float: kept as is, like upstream)
        a = self.s(t)
        v = torch.square(a) * self.sigma_sq(t)
        return torch.log(torch.square(a) / v)


class VP(OUBase):
    """eq/sdes.py:427-555."""

    def __init__(self, beta_min=0.1, beta_max=20.0, sigma=1.0, T=1.0):
        self.bmin, self.bmax, self.sig, self.T = _t(beta_min), _t(beta_max), _t(sigma), _t(T)

    def beta(self, t):  # :456-459
        return torch.lerp(self.bmin, self.bmax, t / self.T)

    def drift_coeff(self, t):  # :461-463
        return -0.5 * self.beta(t)

    def diff_coeff(self, t):  # :465-467
        return self.sig * torch.sqrt(self.beta(t))

    def int_drift_coeff(self, s, t):  # :469-477
        return -0.25 * (self.beta(t) + self.beta(s)) * (t - s)

    def alpha(self, t):  # :490-493
        return self.bmin * t + (0.5 * t ** 2 / self.T) * (self.bmax - self.bmin)

    def transition_params(self, s, t):  # :495-507
        lam = 1.0 - torch.exp(self.alpha(s) - self.alpha(t))
        return torch.sqrt(1.0 - lam), self.sig ** 2 * lam

    def s(self, t):  # :509-511
        return torch.exp(-0.5 * self.alpha(t))

    def sigma_sq(self, t):  # :513-515
        return -self.sig ** 2 * (1.0 - (1.0 / self.s(t) ** 2))

    def omega(self, a, b):  # :517-520
        return 4.0 * self.sig ** 2 * torch.tanh((self.alpha(self.T - a) - self.alpha(self.T - b)) / 4.0)

    def lam(self, a, b):  # :522-524
        return torch.exp(self.alpha(self.T - a) - self.alpha(self.T - b)) - 1.0

    def omega_ddpm(self, a, b):  # :526-530
        la = 1.0 - torch.exp(-self.alpha(self.T - a))
        lb = 1.0 - torch.exp(-self.alpha(self.T - b))
        return self.sig ** 2 * (la / lb) * self.lam(a, b)

    def ei_step(self, x, a, b, sc, z):  # :532-539
        lam = self.lam(a, b)
        out = torch.sqrt(1.0 + lam) * x + 2.0 * self.sig ** 2 * (torch.sqrt(1.0 + lam) - 1.0) * sc
        out += self.sig * torch.sqrt(lam) * z
        return out

    def ddpm_step(self, x, a, b, sc, z):  # :541-555
        T = self.T
        lam = self.lam(a, b)
        lam2 = 1.0 - torch.exp(self.alpha(T - b) - self.alpha(T - a))
        la = 1.0 - torch.exp(-self.alpha(T - a))
        lb = 1.0 - torch.exp(-self.alpha(T - b))
        da = (self.alpha(T - a) - self.alpha(T - b)) / 2.0
        var = self.sig ** 2 * lam2 * (lb / la)
        mean = torch.sqrt(1.0 + lam) * x + 2.0 * self.sig ** 2 * torch.sinh(da) * sc
        return mean + torch.sqrt(var) * z


class CosineVP(VP):
    """eq/sdes.py:558-594."""

    def __init__(self, c=0.008, sigma=1.0, T=1.0):
        super().__init__(0.1, 20.0, sigma, T)
        self.c = _t(c)

    def beta(self, t):  # :579-582
        return torch.pi * torch.tan(0.5 * torch.pi * ((t / self.T) + self.c) / (1.0 + self.c)) / (self.T * (1.0 + self.c))

    def alpha(self, t):  # :592-594
        return -2.0 * torch.log(torch.cos(0.5 * torch.pi * ((t / self.T) + self.c) / (1.0 + self.c)))


class ConstOU(OUBase):
    """eq/sdes.py:354-403."""

    def __init__(self, drift_coeff=2.0, diff_coeff=2.0, T=1.0):
        self.a, self.g, self.T = _t(drift_coeff), _t(diff_coeff), _t(T)

    def drift_coeff(self, t):
        return -self.a

    def diff_coeff(self, t):
        return self.g

    def int_drift_coeff(self, s, t):
        return -self.a * (t - s)

    def s(self, t):
        return torch.exp(-self.a * t)

    def sigma_sq(self, t):
        return -0.5 * self.g ** 2 * (1.0 - torch.exp(2.0 * self.a * t))


class ScaledBM(ConstOU):
    """eq/sdes.py:406-424 (drift coefficient is -0.0)."""

    def __init__(self, diff_coeff=2.0, T=1.0):
        super().__init__(0.0, diff_coeff, T)

    def s(self, t):
        return torch.ones_like(_t(t))

    def sigma_sq(self, t):
        return self.g ** 2 * t


class PinnedBM(OUBase):
    """eq/sdes.py:597-678."""

    def __init__(self, diff_coeff=2.0, T=1.0):
        self.g, self.T = _t(diff_coeff), _t(T)

    def drift_coeff(self, t):
        return -1.0 / (self.T - t)

    def diff_coeff(self, t):
        return self.g

    def int_drift_coeff(self, s, t):
        return torch.log(self.T - t) - torch.log(self.T - s)

    def transition_params(self, s, t):
        m = (self.T - t) / (self.T - s)
        return m, m * (t - s) * self.g ** 2

    def s(self, t):
        return (self.T - t) / self.T

    def sigma_sq(self, t):
        return self.g ** 2 * self.T * t / (self.T - t)

    def omega(self, a, b):  # :649-651
        return self.g ** 2 * (a / b) * (b - a)

    def omega_ddpm(self, a, b):  # :653-656
        return self.g ** 2 * ((self.T - a) / (self.T - b)) * (b - a)

    def ei_step(self, x, a, b, sc, z):  # :658-666
        out = (b / a) * x + self.g ** 2 * (b - a) * sc
        var = self.g ** 2 * (b / a) * (b - a)
        out += torch.sqrt(var) * z
        return out

    def ddpm_step(self, x, a, b, sc, z):  # :668-678
        var = self.g ** 2 * ((self.T - b) / (self.T - a)) * (b - a)
        mean = (b / a) * x
        mean += self.g ** 2 * (b - a) * sc
        return mean + torch.sqrt(var) * z


# --------------------------------------------------------------------------- #
# targets / priors / references -- distr/*.py
# --------------------------------------------------------------------------- #
LOG_2PI = math.log(2.0 * math.pi)


def mog_component_logp(x, mean, var):
    """distr/gauss.py:67-73 log_prob_gaussian -> [B,K]."""
    lp = -0.5 * torch.sum(torch.square(x.unsqueeze(1) - mean.unsqueeze(0)) / var.unsqueeze(0), dim=-1)
    lp -= 0.5 * mean.shape[-1] * LOG_2PI
    lp -= 0.5 * torch.log(var).sum(dim=-1).unsqueeze(0)
    return lp


def mog_score(x, w, mean, var):
    """distr/gauss.py:97-107 score_mog (the in-place weight normalisation is reproduced on a copy)."""
    w = w / w.sum()
    p = torch.softmax(torch.log(w.unsqueeze(0)) + mog_component_logp(x, mean, var), dim=-1)
    return -torch.sum(p.unsqueeze(-1) * (x.unsqueeze(1) - mean.unsqueeze(0)) / var.unsqueeze(0), dim=1)


def mog_full_logp_and_ptd(x, mean, cov):
    """distr/gauss.py:75-94 log_prob_gaussian_full with covariance matrices: (log N(x; m_c, C_c) [B,K], C_c^-1 (x - m_c) [B,K,d])."""
    diff = x.unsqueeze(1) - mean.unsqueeze(0)
    ptd = torch.linalg.solve(cov.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)
    lp = -0.5 * torch.sum(diff * ptd, dim=-1)
    lp = lp - 0.5 * mean.shape[-1] * math.log(2.0 * math.pi)
    lp = lp - 0.5 * torch.logdet(cov).unsqueeze(0)
    return lp, ptd


def mog_score_full(x, w, mean, cov):
    """distr/gauss.py:110-121 score_mog_full (covariance form)."""
    w = w / w.sum()
    lp, ptd = mog_full_logp_and_ptd(x, mean, cov)
    p = torch.softmax(torch.log(w.unsqueeze(0)) + lp, dim=-1)
    return -torch.sum(p.unsqueeze(-1) * ptd, dim=1)


def mog_score_full_prec(x, w, mean, prec, log_det):
    """distr/gauss.py:110-121 score_mog_full in the precision form (precisions and log-determinants given, :81-90)."""
    w = w / w.sum()
    diff = x.unsqueeze(1) - mean.unsqueeze(0)
    ptd = torch.matmul(prec.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)
    lp = -0.5 * torch.sum(diff * ptd, dim=-1)
    lp = lp - 0.5 * mean.shape[-1] * math.log(2.0 * math.pi)
    lp = lp - 0.5 * log_det.unsqueeze(0)
    p = torch.softmax(torch.log(w.unsqueeze(0)) + lp, dim=-1)
    return -torch.sum(p.unsqueeze(-1) * ptd, dim=1)


class GMMFullPrec:
    """distr/gauss.py GMMFull built from precisions and covariance log-determinants (eq/sdes.py:316-318)."""

    def __init__(self, loc, prec, log_det, weights):
        self.loc, self.prec, self.log_det, self.w = loc, prec, log_det, weights / weights.sum()

    def logp(self, x):
        diff = x.unsqueeze(1) - self.loc.unsqueeze(0)
        ptd = torch.matmul(self.prec.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)
        lp = -0.5 * torch.sum(diff * ptd, dim=-1) - 0.5 * self.loc.shape[-1] * math.log(2.0 * math.pi) - 0.5 * self.log_det.unsqueeze(0)
        return torch.logsumexp(torch.log(self.w).unsqueeze(0) + lp, dim=-1, keepdim=True)


class GMMFullCov:
    """distr/gauss.py GMMFull (covariance form): log-density of a full-covariance mixture."""

    def __init__(self, loc, cov, weights):
        self.loc, self.cov, self.w = loc, cov, weights / weights.sum()

    def logp(self, x):
        lp, _ = mog_full_logp_and_ptd(x, self.loc, self.cov)
        return torch.logsumexp(torch.log(self.w).unsqueeze(0) + lp, dim=-1, keepdim=True)


class GMMFullTarget(GMMFullPrec):
    """distr/gauss.py:310-365 GMMFull(cov=...) as a TARGET: the constructor inverts the covariances once (fp32 ``torch.linalg.inv`` /
    ``torch.logdet``, :327-333) and both the log-density and ``score`` (:362-365, score_mog_full :110-121) use the precision form."""

    def __init__(self, loc, cov, weights):
        super().__init__(loc, torch.linalg.inv(cov), torch.logdet(cov), weights)

    def score(self, x):
        return mog_score_full_prec(x, self.w, self.loc, self.prec, self.log_det)


def gauss_score(x, mean, var):
    """distr/gauss.py:124-126."""
    return -(x - mean) / var


class GMMDiag:
    """distr/gauss.py:138-244 GMM with mixture weights (MixtureSameFamily log-prob written out)."""

    def __init__(self, loc, scale, weights):
        self.loc, self.scale, self.w = loc.to(F32), scale.to(F32), weights.to(F32)
        self.dim = loc.shape[-1]

    def logp(self, x):
        """:217-221 -> torch.distributions.MixtureSameFamily.log_prob -> [B,1]."""
        probs = self.w / self.w.sum(-1, keepdim=True)  # Categorical(probs=...)
        eps = torch.finfo(F32).eps
        logits = torch.log(probs.clamp(min=eps, max=1 - eps))
        log_mix = torch.log_softmax(logits, dim=-1)
        xe = x.unsqueeze(-2)
        var = self.scale ** 2
        comp = -((xe - self.loc) ** 2) / (2 * var) - self.scale.log() - math.log(math.sqrt(2 * math.pi))
        comp = comp.sum(-1)
        return torch.logsumexp(comp + log_mix, dim=-1).unsqueeze(-1)

    def score(self, x):
        """:240-242."""
        return mog_score(x, self.w, self.loc, torch.square(self.scale))

    def sample(self, n, generator=None):
        probs = self.w / self.w.sum()
        idx = torch.multinomial(probs, n, replacement=True, generator=generator)
        eps = torch.randn(n, self.dim, generator=generator)
        return self.loc[idx] + self.scale[idx] * eps


def many_modes(n_modes=3, dim=2, seed_loc=42, mixture_weight_factor=3.0, var=0.1) -> GMMDiag:
    """distr/gauss.py:569-594 ManyModes parameters."""
    g = torch.Generator()
    g.manual_seed(seed_loc)
    w = torch.logspace(0.0, 1.0, n_modes, base=mixture_weight_factor)
    loc = 2 * n_modes * torch.rand((n_modes, dim), generator=g) - n_modes
    scale = torch.sqrt(var * torch.ones_like(loc))
    return GMMDiag(loc, scale, w)


def two_modes(dim=2, a=1.0, ill_conditioned="not") -> GMMDiag:
    """distr/gauss.py:422-466 TwoModes parameters."""
    w = torch.FloatTensor([2.0, 1.0])
    loc = torch.stack([-a * torch.ones((dim,)), a * torch.ones((dim,))])
    if ill_conditioned == "medium":
        scale = torch.sqrt(0.05 * torch.logspace(-1, 0.0, dim)).unsqueeze(0).expand(2, -1)
    elif ill_conditioned == "hard":
        scale = torch.sqrt(0.05 * torch.logspace(-2.0, 0.0, dim)).unsqueeze(0).expand(2, -1)
    else:
        scale = torch.sqrt(0.05 * torch.ones_like(loc))
    return GMMDiag(loc, scale.contiguous(), w)


class GaussDiag:
    """distr/gauss.py:597-629 Gauss (a one-component GMM without mixture weights)."""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc.to(F32).reshape(-1), scale.to(F32).reshape(-1)
        self.dim = self.loc.numel()

    def logp(self, x):
        var = self.scale ** 2
        lp = -((x - self.loc) ** 2) / (2 * var) - self.scale.log() - math.log(math.sqrt(2 * math.pi))
        return lp.sum(-1, keepdim=True)

    def score(self, x):
        return gauss_score(x, self.loc.unsqueeze(0), torch.square(self.scale).unsqueeze(0))


class IsoGauss:
    """distr/gauss.py:720-787 IsotropicGauss."""

    def __init__(self, dim, loc=0.0, scale=1.0):
        self.dim, self.loc, self.scale = dim, _t(loc), _t(scale)

    def logp(self, x):  # :757-762
        var = self.scale ** 2
        nc = -0.5 * self.dim * (2.0 * math.pi * var).log()
        return nc - 0.5 * torch.sum((x - self.loc) ** 2, dim=-1, keepdim=True) / var

    def score(self, x):  # :764-766
        return (self.loc - x) / self.scale ** 2

    def sample(self, n, generator=None):  # :777-779
        return self.loc + self.scale * torch.randn(n, self.dim, generator=generator)


class GaussFull:
    """distr/gauss.py:632-717 GaussFull (MultivariateNormal log-prob written out)."""

    def __init__(self, loc, cov):
        self.loc, self.cov = loc.to(F32), cov.to(F32)
        self.dim = loc.numel()
        self.prec = torch.linalg.inv(self.cov)
        self.tril = torch.linalg.cholesky(self.cov)

    def logp(self, x):
        diff = x - self.loc
        y = torch.linalg.solve_triangular(self.tril, diff.T, upper=False).T
        m = y.pow(2).sum(-1)
        half_log_det = self.tril.diagonal().log().sum(-1)
        return (-0.5 * (self.dim * LOG_2PI + m) - half_log_det).unsqueeze(-1)

    def score(self, x):  # :715-717 / :129-135
        diff = x - self.loc.unsqueeze(0)
        return -torch.matmul(self.prec.unsqueeze(0), diff.unsqueeze(-1)).squeeze(-1)

    def sample(self, n, generator=None):
        return self.loc + torch.randn(n, self.dim, generator=generator) @ self.tril.T


class PhiFour:
    """distr/phi_four.py:8-96 (1-D lattice, Dirichlet-0 boundary, no tilt)."""

    def __init__(self, a, b, dim, beta=1.0):
        self.a, self.b, self.dim, self.beta = a, b, dim, beta
        self.coef = a * dim

    def U(self, x):  # :54-79, :44-52
        xp = torch.nn.functional.pad(x, (1, 1), value=0.0)
        grad = ((xp[:, 1:] - xp[:, :-1]) ** 2 / 2).sum(1)
        V = ((1 - x ** 2) ** 2 / 4 + self.b * x).sum(1) / self.coef
        return grad * self.coef + V

    def grad_U(self, x):  # :81-90
        g = (self.b - x * (1.0 - torch.square(x))) / self.coef
        g[:, 1:-1] += self.coef * (2.0 * x[:, 1:-1] - x[:, 2:] - x[:, :-2])
        g[:, 0] += self.coef * (2.0 * x[:, 0] - x[:, 1])
        g[:, -1] += self.coef * (2.0 * x[:, -1] - x[:, -2])
        return g

    def logp(self, x):  # :92-93
        return -self.beta * self.U(x).unsqueeze(-1)

    def score(self, x):  # :95-96
        return -self.beta * self.grad_U(x)


class LogReg:
    """distr/logistic_regression.py:11-92 with the score the reference actually uses:
    autograd through sigmoid -> clip -> probs_to_logits(clamp eps) -> BCE-with-logits
    (distr/base.py:146-154), restated here as a hand-derived closed form."""

    def __init__(self, X, y, weight_scale=1.0, intercept_mean=0.0, intercept_scale=2.5, threshold=1e-8):
        self.X, self.y = X.to(F32), y.to(F32).flatten()
        self.dim = X.shape[1] + 1
        self.ws, self.im, self.isc = _t(weight_scale), _t(intercept_mean), _t(intercept_scale)
        self.thr = threshold

    def _split(self, p):
        return p[..., :-1], p[..., -1]

    def _lik_terms(self, params):
        w, c = self._split(params)
        probs = torch.special.expit(torch.matmul(self.X, w.T).T + c.unsqueeze(-1))
        pc = torch.clip(probs, self.thr, 1.0 - self.thr)
        eps = torch.finfo(F32).eps
        pcc = pc.clamp(min=eps, max=1 - eps)
        logits = torch.log(pcc) - torch.log1p(-pcc)
        return probs, pc, pcc, logits

    def logp(self, params):  # :41-61
        w, c = self._split(params)
        dw = w.shape[-1]
        prior = (-(w ** 2) / (2 * self.ws ** 2) - self.ws.log() - math.log(math.sqrt(2 * math.pi))).sum(-1)
        prior = prior + (-((c - self.im) ** 2) / (2 * self.isc ** 2) - self.isc.log() - math.log(math.sqrt(2 * math.pi)))
        _, _, _, logits = self._lik_terms(params)
        yb = self.y.unsqueeze(0).expand((logits.shape[0], -1))
        ll = -torch.nn.functional.binary_cross_entropy_with_logits(logits, yb, reduction="none").sum(dim=-1)
        return (ll + prior).unsqueeze(-1)

    def score(self, params):
        """Closed form of the autograd gradient ('Logreg autograd parity')."""
        w, c = self._split(params)
        probs, pc, pcc, logits = self._lik_terms(params)
        eps = torch.finfo(F32).eps
        m = ((probs >= self.thr) & (probs <= 1.0 - self.thr) & (pc >= eps) & (pc <= 1 - eps)).to(F32)
        r = (self.y.unsqueeze(0) - torch.sigmoid(logits)) * (1.0 / pcc + 1.0 / (1.0 - pcc)) * (probs * (1.0 - probs)) * m
        gw = -w / self.ws ** 2 + r @ self.X
        gc = -(c - self.im) / self.isc ** 2 + r.sum(-1)
        return torch.cat([gw, gc.unsqueeze(-1)], dim=-1)


class Rings:
    """distr/rings.py:38-109 (2-D)."""

    def __init__(self, lower_rad=1.0, upper_rad=5.0, num_rad=3, scale=0.1, equilibrated=False):
        self.rad = torch.linspace(lower_rad, upper_rad, num_rad)
        w = torch.ones((num_rad,)) if equilibrated else self.rad / self.rad.sum()
        self.w = w / w.sum()
        self.scale = _t(scale)
        self.dim = 2

    def logp(self, v):  # :93-98
        r = torch.linalg.norm(v, dim=-1)
        th = torch.atan2(v[..., 1], v[..., 0])
        th = th + (th < 0).type_as(v) * (2 * torch.pi)
        eps = torch.finfo(F32).eps
        log_mix = torch.log_softmax(torch.log(self.w.clamp(min=eps, max=1 - eps)), -1)
        comp = -((r.unsqueeze(-1) - self.rad) ** 2) / (2 * self.scale ** 2) - self.scale.log() - math.log(math.sqrt(2 * math.pi))
        lr = torch.logsumexp(comp + log_mix, dim=-1)
        la = -torch.log(torch.tensor(2 * torch.pi))  # Uniform(0, 2pi).log_prob
        return (lr + la - torch.log(r)).view((-1, 1))

    def score(self, x, eps=1e-7):  # :100-109
        n = torch.linalg.norm(x, dim=-1, keepdim=True) + eps
        var = (self.scale ** 2) * torch.ones_like(self.rad)
        sr = mog_score(n, self.w.clone(), self.rad.unsqueeze(-1), var.unsqueeze(-1))
        return x * ((sr / n) - (1.0 / torch.square(n)))


# --------------------------------------------------------------------------- #
# drift nets -- models/mlp.py, models/reparam.py
# --------------------------------------------------------------------------- #
def gelu(v):
    return torch.nn.functional.gelu(v)


def time_embed(p: dict, pre: str, t: torch.Tensor) -> torch.Tensor:
    """models/mlp.py:85-96 TimeEmbed.forward; ``p`` is a state_dict, ``pre`` the key prefix.
    ``timestep_coeff`` is a non-persistent buffer: linspace(0.1, 100, channels) (:70-74)."""
    t = t.view(-1, 1).float()
    ch = p[pre + "timestep_phase"].shape[1]
    coeff = torch.linspace(start=0.1, end=100, steps=ch).unsqueeze(0)
    ph = p[pre + "timestep_phase"]
    e = torch.cat([torch.sin((coeff * t) + ph), torch.cos((coeff * t) + ph)], dim=1)
    i = 0
    while pre + f"hidden_layer.{i}.weight" in p:
        e = gelu(torch.nn.functional.linear(e, p[pre + f"hidden_layer.{i}.weight"], p[pre + f"hidden_layer.{i}.bias"]))
        i += 1
    return torch.nn.functional.linear(e, p[pre + "out_layer.weight"], p[pre + "out_layer.bias"])


def fourier_mlp(p: dict, pre: str, t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """models/mlp.py:135-143 FourierMLP.forward."""
    lin = torch.nn.functional.linear
    te = time_embed(p, pre + "timestep_embed.", t.view(-1, 1).expand(x.shape[0], 1).float())
    h = lin(x, p[pre + "input_embed.weight"], p[pre + "input_embed.bias"]) + te
    i = 0
    while pre + f"hidden_layer.{i}.weight" in p:
        h = lin(gelu(h), p[pre + f"hidden_layer.{i}.weight"], p[pre + f"hidden_layer.{i}.bias"])
        i += 1
    return lin(gelu(h), p[pre + "out_layer.weight"], p[pre + "out_layer.bias"])


class Ctrl:
    """models/reparam.py:18-43 (ClippedCtrl), :63-117 (ScoreCtrl), :148-199 (LerpCtrl),
    :120-145 (CancelDriftCtrl).  ``params`` is the wrapper module's state_dict."""

    def __init__(self, params: dict, kind="clipped", clip_model=1e4, target_score: Callable | None = None,
                 clip_score=1e4, scale_score=1.0, sde=None, prior_score: Callable | None = None,
                 use_rescaling=True):
        self.p = {k: v.detach().to(F32) for k, v in params.items()}
        self.kind, self.clip_model, self.clip_score, self.scale_score = kind, clip_model, clip_score, scale_score
        self.target_score, self.prior_score, self.sde, self.use_rescaling = target_score, prior_score, sde, use_rescaling
        self.has_score_model = any(k.startswith("score_model.") for k in self.p)

    def base(self, t, x):
        return clip(fourier_mlp(self.p, "base_model.", t, x), self.clip_model)

    def score_model(self, t):
        return clip(time_embed(self.p, "score_model.", t), self.clip_model)

    def __call__(self, t, x):
        u = self.base(t, x)
        if self.kind == "clipped":
            return u
        if self.kind == "score":  # reparam.py:112-117
            sc = self.scale_score * clip(self.target_score(x), self.clip_score)
            if self.has_score_model:
                sc *= self.score_model(t)
            return u + sc
        if self.kind == "lerp":  # reparam.py:166-199 (hard_constrain False)
            sc = torch.lerp(self.prior_score(x), self.target_score(x), t / self.sde.T)
            sc = self.scale_score * clip(sc, self.clip_score)
            if self.has_score_model:
                sc *= self.score_model(t)
            return u + self.sde.diff(t) * sc
        if self.kind == "cancel_drift":  # reparam.py:131-145
            g = self.sde.diff(t)
            f = self.sde.drift(t, x)
            sc = self.scale_score * clip(self.target_score(x), self.clip_score)
            if self.has_score_model:
                sc *= self.score_model(t)
            if self.use_rescaling:
                return u + (f / g) + 0.5 * g * sc
            return u + (f / torch.square(g)) + 0.5 * sc
        raise ValueError(self.kind)


class RemoveReference:
    """models/reparam.py:46-64 RemoveReferenceCtrl in the form its forward can evaluate (``use_rescaling=False``):
    ``ret = score(t, x); ret -= ref_score(t, x)``."""

    def __init__(self, score, ref_score):
        self.score, self.ref_score = score, ref_score

    def __call__(self, t, x):
        ret = self.score(t, x)
        ret -= self.ref_score(t, x)
        return ret


# --------------------------------------------------------------------------- #
# noise sources
# --------------------------------------------------------------------------- #
class InjectedNoise:
    """z[k] = noise[k] -- the reference draws one randn_like(x) per step (losses/oc.py:277,
    eq/sdes.py:537, losses/oc.py:722, :1372, :1222), so a pre-drawn [N,B,d] tensor replays it."""

    def __init__(self, z):
        self.z = z

    def __call__(self, k, x):
        return self.z[k]


class TorchNoise:
    def __call__(self, k, x):
        return torch.randn_like(x)


PHILOX_M0, PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
PHILOX_W0, PHILOX_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011), vectorised over numpy uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * PHILOX_M0
            p1 = c2.astype(np.uint64) * PHILOX_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + PHILOX_W0)
            k1 = np.uint32(k1 + PHILOX_W1)
    return c0, c1, c2, c3


def philox_normal(seed: int, step: int, particle0: int, B: int, d: int, stream: int = 0) -> torch.Tensor:
    """Engine noise definition ("identical seeds" mode): the four normals of features
    4j..4j+3 of global particle p at step k come from Philox4x32-10 with counter
    (p, j, k, stream) and key (seed_lo, seed_hi), through two Box-Muller pairs on
    u = ((bits >> 9) + 0.5) * 2^-23.  Independent of batch sharding by construction."""
    nj = (d + 3) // 4
    p = (np.arange(B, dtype=np.uint64) + np.uint64(particle0))[:, None].astype(np.uint32)
    j = np.arange(nj, dtype=np.uint32)[None, :]
    pp, jj = np.broadcast_arrays(p, j)
    r = philox4x32_10(pp, jj, np.full_like(pp, step), np.full_like(pp, stream), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [((ri >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23) for ri in r]
    out = np.empty((B, nj, 4), dtype=np.float32)
    two_pi = np.float32(2.0 * math.pi)
    for a in (0, 1):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[2 * a])).astype(np.float32)
        ang = (two_pi * u[2 * a + 1]).astype(np.float32)
        out[:, :, 2 * a] = rad * np.cos(ang).astype(np.float32)
        out[:, :, 2 * a + 1] = rad * np.sin(ang).astype(np.float32)
    return torch.from_numpy(out.reshape(B, nj * 4)[:, :d].copy())


class PhiloxNoise:
    def __init__(self, seed, particle0=0):
        self.seed, self.p0 = seed, particle0

    def __call__(self, k, x):
        return philox_normal(self.seed, k, self.p0, x.shape[0], x.shape[1])


# --------------------------------------------------------------------------- #
# the step loops -- losses/oc.py  (eval path: change_sde_ctrl=False, sde_ctrl is ctrl)
# --------------------------------------------------------------------------- #
def simulate_em_ref(ts, x, ctrl, sde, terminal_logp, ref_logp, ref_score=None, noise=None,
                    use_rescaling=True, return_traj=False):
    """losses/oc.py:218-296 EMReferenceSDELoss.simulate (PIS when ref_score is None)."""
    noise = noise or TorchNoise()
    rnd = 0.0
    T = ts[-1]
    xs = [x] if return_traj else None
    for k, (s, t) in enumerate(zip(ts[:-1], ts[1:])):
        u = ctrl(T - s, x)
        g = sde.diff(T - s)
        dt = t - s
        if not use_rescaling:  # :265-267 -- same tensor scaled twice (quirk kept)
            u = u * g
            u = u * g
        rnd = rnd + 0.5 * (u ** 2).sum(dim=-1, keepdim=True) * dt
        db = noise(k, x) * dt.sqrt()
        f = -sde.drift(T - s, x)
        if ref_score is not None:
            f = f + torch.square(g) * ref_score(T - s, x)
        x = x + (f + g * u) * dt + g * db
        rnd = rnd + rowdot(u, db)
        if return_traj:
            xs.append(x)
    rnd = rnd + (ref_logp(x).view((-1, 1)) - terminal_logp(x))
    return x, rnd, (torch.stack(xs) if return_traj else None)


def simulate_ei_ref(ts, x, ctrl, sde, terminal_logp, ref_logp, ref_score, noise=None, ddpm=False,
                    return_traj=False):
    """losses/oc.py:444-510 EIReferenceSDELoss.simulate; ddpm=True -> :584-651 DDPMLikeReferenceSDELoss."""
    noise = noise or TorchNoise()
    omega = sde.omega_ddpm if ddpm else sde.omega
    step = sde.ddpm_step if ddpm else sde.ei_step
    rnd = 0.0
    T = ts[-1]
    xs = [x] if return_traj else None
    for k, (s, t) in enumerate(zip(ts[:-1], ts[1:])):
        u = ctrl(T - s, x)
        rnd = rnd + 0.5 * omega(s, t) * (u ** 2).sum(dim=-1, keepdim=True)
        z = noise(k, x)
        x = step(x, s, t, ref_score(T - s, x) + u, z)
        rnd = rnd + torch.sqrt(omega(s, t)) * rowdot(u, z)
        if return_traj:
            xs.append(x)
    rnd = rnd + (ref_logp(x).view((-1, 1)) - terminal_logp(x))
    return x, rnd, (torch.stack(xs) if return_traj else None)


def eubo_ei_ref(ts, x, ctrl, sde, terminal_logp, ref_logp, ref_score, noise=None):
    """losses/oc.py:512-568 EIReferenceSDELoss.compute_eubo: noising trajectories from target samples x."""
    noise = noise or TorchNoise()
    rnd = ref_logp(x).view((-1, 1)) - terminal_logp(x)
    T = ts[-1]
    times_s, times_t = ts[:-1].flip((0,)), ts[1:].flip((0,))
    mean_f, var_f = sde.transition_params(T - times_t, T - times_s)
    std_f = var_f.sqrt()
    for i, (s, t) in enumerate(zip(times_s, times_t)):
        z = noise(i, x)
        x = x * mean_f[i]
        x = x + std_f[i] * z
        u = ctrl(T - s, x)
        r = ref_score(T - s, x)
        cost = u * (r + 0.5 * u)
        rnd = rnd - cost.sum(dim=-1, keepdim=True) * sde.omega(s, t)
        rnd = rnd - (u * z).sum(dim=-1, keepdim=True) * torch.sqrt(sde.omega(s, t))
    return x, rnd


def eubo_em_ref(ts, x, ctrl, sde, terminal_logp, ref_logp, ref_score, noise=None, use_rescaling=True):
    """losses/oc.py:298-362 EMReferenceSDELoss.compute_eubo (also DDPMLikeReferenceSDELoss, which inherits it)."""
    noise = noise or TorchNoise()
    rnd = ref_logp(x).view((-1, 1)) - terminal_logp(x)
    T = ts[-1]
    times_s, times_t = ts[:-1].flip((0,)), ts[1:].flip((0,))
    mean_f, var_f = sde.transition_params(T - times_t, T - times_s)
    std_f = var_f.sqrt()
    for i, (s, t) in enumerate(zip(times_s, times_t)):
        z = noise(i, x)
        x = x * mean_f[i]
        x = x + std_f[i] * z
        u = ctrl(T - s, x)
        r = ref_score(T - s, x)
        g = sde.diff(T - s)
        dt = t - s
        if use_rescaling:
            u = u / g
        cost = u * (r + 0.5 * u)
        rnd = rnd - cost.sum(dim=-1, keepdim=True) * dt * g ** 2
        rnd = rnd + (u * x).sum(dim=-1, keepdim=True) * (1.0 / mean_f[i] - 1.0 + sde.drift_coeff(T - s) * dt)
        rnd = rnd - (u * z).sum(dim=-1, keepdim=True) * (std_f[i] / mean_f[i])
    return x, rnd


def eubo_dis_ei(ts, x, ctrl, sde, terminal_logp, initial_logp, noise=None):
    """losses/oc.py:980-1036 DiscreteTimeReversalLossEI.compute_eubo."""
    noise = noise or TorchNoise()
    rnd = -terminal_logp(x)
    T = ts[-1]
    times_s, times_t = ts[:-1].flip((0,)), ts[1:].flip((0,))
    mean_f, var_f = sde.transition_params(T - times_t, T - times_s)
    std_f = var_f.sqrt()
    for i, (s, t) in enumerate(zip(times_s, times_t)):
        z = noise(i, x)
        x = x * mean_f[i]
        x = x + std_f[i] * z
        u = ctrl(T - s, x)
        cost = 0.5 * u ** 2
        rnd = rnd - cost.sum(dim=-1, keepdim=True) * sde.omega(s, t)
        rnd = rnd - (u * z).sum(dim=-1, keepdim=True) * torch.sqrt(sde.omega(s, t))
    rnd = rnd + initial_logp(x)
    return x, rnd


def simulate_dis_ei(ts, x, ctrl, sde, terminal_logp, initial_logp, noise=None, return_traj=False):
    """losses/oc.py:906-978 DiscreteTimeReversalLossEI.simulate (eval: rnd0 = prior log-prob)."""
    noise = noise or TorchNoise()
    rnd = initial_logp(x)
    T = ts[-1]
    xs = [x] if return_traj else None
    for k, (s, t) in enumerate(zip(ts[:-1], ts[1:])):
        u = ctrl(T - s, x)
        rnd = rnd + 0.5 * sde.omega(s, t) * (u ** 2).sum(dim=-1, keepdim=True)
        z = noise(k, x)
        x = sde.ei_step(x, s, t, u, z)
        rnd = rnd + torch.sqrt(sde.omega(s, t)) * rowdot(u, z)
        if return_traj:
            xs.append(x)
    rnd = rnd - terminal_logp(x)
    return x, rnd, (torch.stack(xs) if return_traj else None)


def simulate_time_reversal(ts, x, ctrl, sde, terminal_logp, initial_logp, noise=None,
                           compute_ito_int=True, train=False, return_traj=False):
    """losses/oc.py:1133-1238 TimeReversalLoss.simulate without inference control."""
    noise = noise or TorchNoise()
    rnd = initial_logp(x)
    xs = [x] if return_traj else None
    for k, (s, t) in enumerate(zip(ts[:-1], ts[1:])):
        u = ctrl(s, x)
        g = sde.diff(s)
        dt = t - s
        rnd = rnd + 0.5 * (u ** 2).sum(dim=-1, keepdim=True) * dt
        if not train:
            rnd = rnd - sde.int_drift_coeff(s, t) * x.shape[-1]
        db = noise(k, x) * dt.sqrt()
        x = x + (sde.drift(s, x) + g * u) * dt + g * db
        if compute_ito_int:
            rnd = rnd + rowdot(u, db)
        if return_traj:
            xs.append(x)
    rnd = rnd - terminal_logp(x)
    return x, rnd, (torch.stack(xs) if return_traj else None)


def simulate_dds(ts, x, ctrl, alpha, sigma, terminal_logp, ref_logp, noise=None, compute_ito_int=True,
                 return_traj=False):
    """losses/oc.py:1319-1397 ExponentialIntegratorSDELoss.simulate (net time is s)."""
    noise = noise or TorchNoise()
    rnd = 0.0
    xs = [x] if return_traj else None
    for k, (s, t) in enumerate(zip(ts[:-1], ts[1:])):
        u = ctrl(s, x)
        cost = 0.5 * (u ** 2).sum(dim=-1, keepdim=True)
        dt = t - s
        beta_k = torch.clip(alpha * dt.sqrt(), 0, 1)
        alpha_k = torch.sqrt(1.0 - beta_k ** 2)
        rnd = rnd + beta_k ** 2 * sigma ** 2 * cost
        eps = noise(k, x)
        x = x * alpha_k + (beta_k ** 2) * (sigma ** 2) * u + sigma * beta_k * eps
        if compute_ito_int:
            rnd = rnd + (sigma * u * eps * beta_k).sum(dim=-1, keepdim=True)
        if return_traj:
            xs.append(x)
    rnd = rnd + (ref_logp(x).view((-1, 1)) - terminal_logp(x))
    return x, rnd, (torch.stack(xs) if return_traj else None)


def langevin_drift(t, x, target_score, prior_score, g, T, clip_score):
    """eq/sdes.py:101-110 ControlledLangevinSDE.drift."""
    d = target_score(x) * (t / T) + prior_score(x) * (1.0 - t / T)
    d = d * (0.5 * g ** 2)
    return clip(d, clip_score)


def eubo_cmcd(ts, x, ctrl, target_score, prior_score, g, T, clip_score, terminal_logp, initial_logp, noise=None):
    """losses/oc.py:757-828 ControlledLangevinSDELoss.compute_eubo (use_rescaling=True).  Note :806: the drift at y is
    evaluated at time t, not s."""
    noise = noise or TorchNoise()
    g, T = _t(g), _t(T)
    rnd = -terminal_logp(x)
    times_s, times_t = ts[:-1].flip((0,)), ts[1:].flip((0,))
    for i, (s, t) in enumerate(zip(times_s, times_t)):
        u_t = ctrl(t, x)
        dt = t - s
        db = dt.sqrt() * noise(i, x)
        drift_t = langevin_drift(t, x, target_score, prior_score, g, T, clip_score)
        y = x + (drift_t - u_t * g) * dt + g * db
        drift_s = langevin_drift(t, y, target_score, prior_score, g, T, clip_score)
        u_s = ctrl(s, y)
        cost = (drift_s + drift_t) / g + u_s - u_t
        rnd = rnd - 0.5 * (cost ** 2).sum(dim=-1, keepdim=True) * dt
        rnd = rnd - (cost * db).sum(dim=-1, keepdim=True)
        x = y
    rnd = rnd + initial_logp(x)
    return x, rnd


def simulate_cmcd(ts, x, ctrl, target_score, prior_score, g, T, clip_score, terminal_logp, initial_logp,
                  noise=None, return_traj=False):
    """losses/oc.py:666-755 ControlledLangevinSDELoss.simulate (eval path, use_rescaling=True)."""
    noise = noise or TorchNoise()
    g, T = _t(g), _t(T)
    rnd = initial_logp(x)
    xs = [x] if return_traj else None
    for k, (s, t) in enumerate(zip(ts[:-1], ts[1:])):
        u_s = ctrl(s, x)
        dt = t - s
        db = dt.sqrt() * noise(k, x)
        b_s = langevin_drift(s, x, target_score, prior_score, g, T, clip_score)
        y = x + (b_s + u_s * g) * dt + g * db
        b_t = langevin_drift(t, y, target_score, prior_score, g, T, clip_score)
        u_t = ctrl(t, y)
        c = (b_s + b_t) / g + u_s - u_t
        rnd = rnd + 0.5 * (c ** 2).sum(dim=-1, keepdim=True) * dt
        rnd = rnd + rowdot(c, u_s - u_s) * dt
        rnd = rnd + rowdot(c, db)
        x = y
        if return_traj:
            xs.append(x)
    rnd = rnd - terminal_logp(x)
    return x, rnd, (torch.stack(xs) if return_traj else None)


# --------------------------------------------------------------------------- #
# estimators -- losses/oc.py:134-173, eval/metrics.py:135-140
# --------------------------------------------------------------------------- #
# --------------------------------------------------------------------------- #
# eq/integrator.py -- EulerIntegrator
# --------------------------------------------------------------------------- #
def langevin_sde_drift(x, target_score, g, clip_score=None):
    """eq/sdes.py:63-71 LangevinSDE.drift: clip(score * g^2 / 2)."""
    return clip(target_score(x) * g ** 2 / 2.0, clip_score)


def euler_integrate(drift, diff, ts, x_init, timesteps, increment, eps=1e-8):
    """eq/integrator.py:93-129 EulerIntegrator.integrate with explicit ``timesteps``.

    ``drift(s, x)`` / ``diff(s)`` are the SDE's coefficient functions, ``increment(k, s, t, x)`` the Brownian increment of
    step k (the reference: ``randn * sqrt(t - s)`` :115, or ``bm(s, t)`` :117).  Returns the states interpolated onto
    ``ts`` (:119-122 with ``interpolate`` :66-77)."""
    ts_count, out, xs = 0, [], x_init
    for k, (s, t) in enumerate(zip(timesteps[:-1], timesteps[1:])):
        noise = increment(k, s, t, xs)
        xt = xs + drift(s, xs) * (t - s) + diff(s) * noise  # :118
        if ts[ts_count] <= t + eps:  # :120
            rest = ts[ts_count:]
            ind = torch.searchsorted(rest, t + eps, side="right")  # :73
            t_eval = rest[:ind]
            assert (s <= t_eval).all() and (t_eval <= t + eps).all()  # :75
            out.append(torch.lerp(xs, xt, (t_eval.view(-1, 1, 1) - s) / (t - s)))  # :76
            ts_count += out[-1].shape[0]
        xs = xt
    out = torch.cat(out)
    assert ts_count == out.shape[0]  # :128
    return out


def controlled_sde_drift(sde, ctrl):
    """eq/sdes.py:709-720 ControlledSDE.f_and_g: sde.drift(t, x) + sde.diff(t) * ctrl(T - t, x)."""
    def drift(t, x):
        out = sde.drift(t, x)
        if ctrl is not None:
            out = out + sde.diff(t) * ctrl(sde.T - t, x)
        return out
    return drift


def compute_results(rnd: torch.Tensor) -> dict:
    neg = -rnd
    w = torch.softmax(neg, dim=0)
    return {
        "elbo": neg.mean().item(),
        "log_norm_const_is": (neg.logsumexp(dim=0) - math.log(len(w))).item(),
        "lv_loss": rnd.var().item(),
        "ess": (w.sum() ** 2 / (w ** 2).sum()).item() / len(w),
        "weights": w,
    }
