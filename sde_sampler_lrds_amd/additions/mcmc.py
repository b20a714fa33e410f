"""MCMC helpers of the learned-reference pipeline (mirror of ``sde_sampler/additions/mcmc.py:8-135, 256-293``): MALA and
random-walk Metropolis steps over a batch of chains.  The per-step work that matters -- the target's log-density and score at
the proposals -- comes from the HIP distribution kernels (``hip_log_prob_and_grad``: one ``sdeng_dist_eval`` launch, no
autograd); the accept/reject bookkeeping is a handful of elementwise torch ops on the device."""
from __future__ import annotations

import math

import torch

from .. import engine as E


def hip_log_prob_and_grad(target):
    """target -> callable y -> (log pi~(y) [B], score(y) [B,d]) evaluated by the HIP kernels (any distribution engine.dist_desc knows)."""
    def fn(y):
        logp, score = E.dist_eval(target, y.detach())
        return logp.flatten(), score
    fn.hip_target = target  # lets mcmc_sample run all its MALA steps in one launch (sdeng_langevin_moves)
    return fn


def sample_multivariate_normal_diag(batch_size, mean, variance):
    z = torch.randn((batch_size, *mean.shape[1:]), device=mean.device)
    return (torch.sqrt(variance) if isinstance(variance, torch.Tensor) else math.sqrt(variance)) * z + mean


def log_prob_multivariate_normal_diag(samples, mean, variance, sum_indexes):
    """Unnormalised diagonal-Gaussian log-density, one variance per chain (additions/mcmc.py:17-31)."""
    ret = -0.5 * torch.sum(torch.square(samples - mean), dim=sum_indexes)
    return ret / (variance.flatten() if isinstance(variance, torch.Tensor) and variance.dim() > 0 else variance)


def heuristics_step_size(stepsize, mean_log_acceptance, target_acceptance=0.75, factor=1.01, tol=0.05):
    """additions/mcmc.py:55-74."""
    shape = (-1, *(1,) * (stepsize.dim() - 1))
    stepsize = torch.where((mean_log_acceptance - math.log(target_acceptance) > math.log1p(tol)).view(shape), stepsize * factor, stepsize)
    return torch.where((math.log(target_acceptance) - mean_log_acceptance > -math.log1p(-tol)).view(shape), stepsize / factor, stepsize)


@torch.no_grad()
def mala_step(y, target_log_prob_y, target_grad_y, target_log_prob_and_grad, step_size):
    """additions/mcmc.py:77-135: y, log pi~(y), score(y) are updated in place where the proposal is accepted."""
    mean_fwd = y + step_size * target_grad_y
    y_prop = sample_multivariate_normal_diag(y.shape[0], mean_fwd, 2.0 * step_size)
    lp_prop, grad_prop = target_log_prob_and_grad(y_prop)
    joint_prop = lp_prop - log_prob_multivariate_normal_diag(y_prop, mean_fwd, 2.0 * step_size, -1)
    joint_orig = target_log_prob_y - log_prob_multivariate_normal_diag(y, y_prop + step_size * grad_prop, 2.0 * step_size, -1)
    log_acc = joint_prop - joint_orig
    mask = torch.log(torch.rand_like(lp_prop)) < log_acc
    y[mask] = y_prop[mask]
    target_log_prob_y[mask] = lp_prop[mask]
    target_grad_y[mask] = grad_prop[mask]
    return y, target_log_prob_y, target_grad_y, log_acc


@torch.no_grad()
def ula_step(y, target_log_prob_y, target_grad_y, target_log_prob_and_grad, step_size):
    """additions/mcmc.py:189-221: the unadjusted Langevin move (always accepted)."""
    y_prop = sample_multivariate_normal_diag(y.shape[0], y + step_size * target_grad_y, 2.0 * step_size)
    return (y_prop, *target_log_prob_and_grad(y_prop))


def _precond_proposal(y, precond_grad_y, step_size, precond_matrix_chol):
    noise = torch.matmul(precond_matrix_chol, torch.randn((*y.shape, 1), device=y.device)).squeeze(-1)
    return y + step_size * precond_grad_y + torch.sqrt(2.0 * step_size) * noise


@torch.no_grad()
def precond_mala_step(y, target_log_prob_y, target_grad_y, precond_grad_y, target_log_prob_and_grad, step_size, precond_matrix,
                      precond_matrix_chol):
    """additions/mcmc.py:137-187: MALA with proposal covariance 2 h P (P = ``precond_matrix``, its Cholesky factor given); the
    acceptance ratio in the gradient form of arXiv:2305.14442, Prop. 1."""
    y_prop = _precond_proposal(y, precond_grad_y, step_size, precond_matrix_chol)
    lp_prop, grad_prop = target_log_prob_and_grad(y_prop)
    pgrad_prop = torch.matmul(precond_matrix, grad_prop.unsqueeze(-1)).squeeze(-1)
    log_acc = lp_prop - target_log_prob_y
    log_acc = log_acc + 0.5 * torch.sum((y - y_prop - 0.5 * step_size * pgrad_prop) * grad_prop, dim=-1)
    log_acc = log_acc - 0.5 * torch.sum((y_prop - y - 0.5 * step_size * precond_grad_y) * target_grad_y, dim=-1)
    mask = torch.log(torch.rand_like(lp_prop)) < log_acc
    y[mask] = y_prop[mask]
    target_log_prob_y[mask] = lp_prop[mask]
    target_grad_y[mask] = grad_prop[mask]
    precond_grad_y[mask] = pgrad_prop[mask]
    return y, target_log_prob_y, target_grad_y, precond_grad_y, log_acc


@torch.no_grad()
def precond_ula_step(y, target_log_prob_y, target_grad_y, precond_grad_y, target_log_prob_and_grad, step_size, precond_matrix,
                     precond_matrix_chol):
    """additions/mcmc.py:224-253."""
    y_prop = _precond_proposal(y, precond_grad_y, step_size, precond_matrix_chol)
    lp_prop, grad_prop = target_log_prob_and_grad(y_prop)
    return y_prop, lp_prop, grad_prop, torch.matmul(precond_matrix, grad_prop.unsqueeze(-1)).squeeze(-1)


@torch.no_grad()
def rwmh_step(y, target_log_prob_y, target_log_prob, step_size):
    """additions/mcmc.py:256-293."""
    y_prop = y + step_size * torch.randn_like(y)
    lp_prop = target_log_prob(y_prop).flatten()
    log_acc = lp_prop - target_log_prob_y
    mask = torch.log(torch.rand((y.shape[0],), device=y.device)) < log_acc
    y[mask] = y_prop[mask]
    target_log_prob_y[mask] = lp_prop[mask]
    return y, target_log_prob_y, log_acc
