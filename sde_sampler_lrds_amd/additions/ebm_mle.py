"""Annealed samplers of ``sde_sampler/additions/ebm_mle.py``: ``smc_sampler`` (:11-195, annealed Langevin / sequential
Monte Carlo), ``make_re_pairings`` (:198-216), ``re_step`` (:219-266) and ``re_sampler`` (:269-400, replica exchange), on top
of the local moves of ``additions/mcmc.py``.  They are host compositions: what costs time is ``log_prob_and_grads(t, x)`` at
every proposal, which ``hip_tempered_log_prob_and_grads`` serves from the HIP distribution kernels (no autograd).

Same arguments, return values, random-number consumption order and diagnostics as the reference for the plain (diagonal)
moves.  Not provided: the preconditioned moves (``precond_matrix_per_noise``), the PDDS transition (``use_pdds_weights``) and
the ``MaximumLikelihoodEBM`` trainer around the samplers (an energy-net training loop, outside the simulate path)."""
from __future__ import annotations

import torch

from .. import engine as E
from .mcmc import heuristics_step_size, mala_step, ula_step


def hip_tempered_log_prob_and_grads(target, prior):
    """(t, x) -> (log pi_t(x) [B], grad [B,d]) for the geometric path pi_t = prior^(1-t) target^t, t in [0,1] one value per
    row ([B,1]); both log-densities and scores come from ``sdeng_dist_eval`` launches (any distribution the engine knows)."""
    def fn(t, x):
        lp1, s1 = E.dist_eval(target, x.detach())
        lp0, s0 = E.dist_eval(prior, x.detach())
        w = t.to(x.dtype).expand(x.shape[0], 1) if t.dim() else t
        return ((1.0 - w) * lp0 + w * lp1).flatten(), (1.0 - w) * s0 + w * s1
    return fn


def _refuse(precond_matrix_per_noise, precond_matrix_chol_per_noise, use_pdds_weights=False):
    if precond_matrix_per_noise is not None or precond_matrix_chol_per_noise is not None:
        raise NotImplementedError("preconditioned MALA / ULA moves (additions/mcmc.py:137-187, 224-254) are not provided")
    if use_pdds_weights:
        raise NotImplementedError("PDDS transitions and weights (additions/ebm_mle.py:88-99) are not provided")


class _LocalMove:
    """One MALA (or ULA) move at a fixed level with the step-size heuristic of additions/mcmc.py:54-72."""

    def __init__(self, log_prob_and_grad, use_ula, target_acceptance):
        self.f, self.use_ula, self.target = log_prob_and_grad, use_ula, target_acceptance

    def __call__(self, x, lp, grad, step):
        if self.use_ula:
            x, lp, grad = ula_step(x, lp, grad, self.f, step)
            return x, lp, grad, step, None
        x, lp, grad, log_acc = mala_step(x, lp, grad, self.f, step)
        if self.target > 0.0:
            step = heuristics_step_size(step, log_acc, target_acceptance=self.target)
        return x, lp, grad, step, log_acc


def smc_sampler(x_init, times, log_prob_and_grads, n_warmup_mcmc_steps, n_mcmc_steps, step_sizes_per_noise, per_noise_init=False,
                reweight_threshold=1.0, use_pdds_weights=False, sde=None, target_acceptance=0.75, precond_matrix_per_noise=None,
                precond_matrix_chol_per_noise=None, use_ula=False, verbose=False):
    """additions/ebm_mle.py:11-195.  Levels are visited from the last (``times[-1]``) to the first; with
    ``reweight_threshold > 0`` the particles carry importance weights between levels and are resampled (multinomial) when
    the normalised ESS drops below the threshold.  Returns (samples [n_levels, n_mcmc_steps, B, *data], updated step sizes, diags)."""
    _refuse(precond_matrix_per_noise, precond_matrix_chol_per_noise, use_pdds_weights)
    if per_noise_init and reweight_threshold > 0.0:
        raise ValueError("Can't use per_noise_init in SMC mode.")
    n_levels = times.shape[0]
    B = x_init.shape[1] if per_noise_init else x_init.shape[0]
    data_shape = x_init.shape[2:] if per_noise_init else x_init.shape[1:]
    smc = reweight_threshold > 0.0
    samples = torch.empty((n_levels, n_mcmc_steps, B, *data_shape), device=x_init.device)
    ess_logs = torch.ones((n_levels,))
    mean_accs = torch.empty((n_levels,))
    log_w = torch.zeros((B,), device=x_init.device)
    x, lp_prev = x_init.clone(), None
    for lvl in range(n_levels - 1, -1, -1):
        x = x_init[lvl].clone() if per_noise_init else x.clone()
        move = _LocalMove(lambda y, lvl=lvl: log_prob_and_grads(times[lvl], y), use_ula, target_acceptance)
        step = step_sizes_per_noise[lvl]
        lp, grad = move.f(x)
        if smc and lvl != n_levels - 1:  # incremental weight: this level's density over the previous level's, at the same points
            log_w += lp - lp_prev
            w = torch.nn.functional.softmax(log_w, dim=0)
            ess = (1.0 / torch.sum(torch.square(w))) / B
            ess_logs[lvl] = ess.cpu().clone()
            if ess < reweight_threshold:
                idx = torch.multinomial(w, B, replacement=True)
                x, lp, grad = x[idx], lp[idx], grad[idx]
                log_w.zero_()
        for _ in range(n_warmup_mcmc_steps):
            x, lp, grad, step, _ = move(x, lp, grad, step)
        acc_sum = 0.0
        for i in range(n_mcmc_steps):
            x, lp, grad, step, log_acc = move(x, lp, grad, step)
            if log_acc is not None:
                acc_sum = acc_sum + torch.exp(torch.minimum(torch.zeros_like(log_acc), log_acc))
            samples[lvl, i] = x.clone()
        if not use_ula:
            mean_accs[lvl] = (acc_sum / n_mcmc_steps).mean()
        step_sizes_per_noise[lvl] = step.clone()
        lp_prev = lp.clone()
        if verbose:
            print(f"smc level {lvl}: ess {float(ess_logs[lvl]):.3f}" + ("" if use_ula else f", local acc {float(mean_accs[lvl]):.3f}"))
    diags = {}
    if not use_ula:
        diags["local_acc"] = mean_accs
    if smc:
        diags["ess"] = ess_logs
    return samples, step_sizes_per_noise, diags


def make_re_pairings(num_noise_levels, device=None):
    """additions/ebm_mle.py:198-216: neighbour pairs (i, i+1) with i even, and with i odd."""
    lvl = torch.arange(num_noise_levels, device=device)
    has_next = lvl + 1 < num_noise_levels
    return [torch.stack([lvl[sel], lvl[sel] + 1], dim=-1) for sel in ((lvl % 2 == 0) & has_next, (lvl % 2 == 1) & has_next)]


def re_step(x, log_prob_x, grad_x, log_prob_and_grads, times, idx_i, idx_j, batch_size, data_shape, data_shape_ones):
    """additions/ebm_mle.py:219-266: propose swapping the states of levels ``idx_i`` and ``idx_j`` chain by chain; accept with
    probability min(1, pi_i(x_j) pi_j(x_i) / (pi_i(x_i) pi_j(x_j)))."""
    n_pairs = idx_i.shape[0]
    lp_ii, lp_jj = log_prob_x[idx_i], log_prob_x[idx_j]
    g_ii, g_jj = grad_x[idx_i], grad_x[idx_j]
    with torch.no_grad():
        lp_ij, g_ij = log_prob_and_grads(times[idx_i], x[idx_j])
        lp_ji, g_ji = log_prob_and_grads(times[idx_j], x[idx_i])
    log_acc = (lp_ij + lp_ji) - (lp_ii + lp_jj)
    swap = torch.rand_like(log_acc).log_().lt_(log_acc).bool()
    re_acc = swap.float().mean()
    old = x.clone()
    log_prob_x[idx_i] = torch.where(swap, lp_ij, lp_ii)
    log_prob_x[idx_j] = torch.where(swap, lp_ji, lp_jj)
    swap = swap.view((n_pairs, batch_size, *data_shape_ones))
    x[idx_i] = torch.where(swap, old[idx_j], old[idx_i])
    x[idx_j] = torch.where(swap, old[idx_i], old[idx_j])
    grad_x[idx_i] = torch.where(swap, g_ij, g_ii)
    grad_x[idx_j] = torch.where(swap, g_ji, g_jj)
    return x, log_prob_x, grad_x, re_acc


def re_sampler(x_init, times, log_prob_and_grads, swap_frequency, n_warmup_mcmc_steps, n_mcmc_steps, step_sizes_per_noise,
               per_noise_init=False, target_acceptance=0.75, precond_matrix_per_noise=None, precond_matrix_chol_per_noise=None,
               use_ula=False, verbose=False):
    """additions/ebm_mle.py:269-400: all levels advance together as one flattened batch of n_levels*B chains; every
    ``swap_frequency``-th step is a swap step (even pairs, then odd pairs, alternating) instead of a local move."""
    _refuse(precond_matrix_per_noise, precond_matrix_chol_per_noise)
    n_levels = times.shape[0]
    B = x_init.shape[1] if per_noise_init else x_init.shape[0]
    data_shape = x_init.shape[2:] if per_noise_init else x_init.shape[1:]
    ones = (1,) * len(data_shape)
    samples = torch.empty((n_levels, n_mcmc_steps, B, *data_shape), device=x_init.device)
    mean_local_accs = torch.zeros((n_levels,))
    mean_swap_acc = 0.0
    t_flat = times.reshape((-1, *ones))

    def batched(t, y):  # (K, B, ...) inputs of a swap step
        lp, g = log_prob_and_grads(t.view((-1, *ones)), y.view((-1, *data_shape)))
        return lp.view(y.shape[:2]), g.view(y.shape)

    move = _LocalMove(lambda y: log_prob_and_grads(t_flat, y), use_ula, target_acceptance)
    x = x_init.clone() if per_noise_init else x_init.unsqueeze(0).repeat((n_levels, 1, *ones))
    x = x.view((-1, *data_shape))
    step = step_sizes_per_noise.view((-1, *ones))
    lp, grad = move.f(x)
    pairs = make_re_pairings(n_levels, x_init.device)
    for it in range(n_warmup_mcmc_steps + n_mcmc_steps):
        if it % swap_frequency == 0:
            pr = pairs[(it // swap_frequency) % 2]
            x, lp, grad, mean_swap_acc = re_step(x.view((-1, B, *data_shape)), lp.view((-1, B)), grad.view((-1, B, *data_shape)), batched,
                                                 times, pr[:, 0], pr[:, 1], B, data_shape, ones)
            x, grad, lp = x.view((-1, *data_shape)), grad.view((-1, *data_shape)), lp.flatten()
        else:
            x, lp, grad, step, log_acc = move(x, lp, grad, step)
            if log_acc is not None:
                mean_local_accs = torch.exp(torch.minimum(torch.zeros_like(log_acc), log_acc)).view((-1, B)).mean(dim=-1)
        if it >= n_warmup_mcmc_steps:
            samples[:, it - n_warmup_mcmc_steps] = x.reshape((-1, B, *data_shape)).clone()
        if verbose and it % 50 == 0:
            print(f"re step {it}: swap acc {float(mean_swap_acc):.3f}")
    diags = {"swap_acc": mean_swap_acc}
    if not use_ula:
        diags["local_acc"] = mean_local_accs
    return samples, step, diags
