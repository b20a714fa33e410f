"""Annealed samplers of ``sde_sampler/additions/ebm_mle.py``: ``smc_sampler`` (:11-195, annealed Langevin / sequential
Monte Carlo), ``make_re_pairings`` (:198-216), ``re_step`` (:219-266) and ``re_sampler`` (:269-400, replica exchange), on top
of the local moves of ``additions/mcmc.py``.  They are host compositions: what costs time is ``log_prob_and_grads(t, x)`` at
every proposal, which ``hip_tempered_log_prob_and_grads`` serves from the HIP distribution kernels (no autograd).

Same arguments, return values, random-number consumption order and diagnostics as the reference, including the
preconditioned moves (``precond_matrix_per_noise``) and the PDDS transition and weights (``use_pdds_weights``).  Not provided:
the ``MaximumLikelihoodEBM`` trainer around the samplers (an energy-net training loop, outside the simulate path)."""
from __future__ import annotations

import torch

from .. import engine as E
from .mcmc import heuristics_step_size, mala_step, precond_mala_step, precond_ula_step, ula_step


def hip_tempered_log_prob_and_grads(target, prior):
    """(t, x) -> (log pi_t(x) [B], grad [B,d]) for the geometric path pi_t = prior^(1-t) target^t, t in [0,1] one value per
    row ([B,1]); both log-densities and scores come from ``sdeng_dist_eval`` launches (any distribution the engine knows)."""
    def fn(t, x):
        lp1, s1 = E.dist_eval(target, x.detach())
        lp0, s0 = E.dist_eval(prior, x.detach())
        w = t.to(x.dtype).expand(x.shape[0], 1) if t.dim() else t
        return ((1.0 - w) * lp0 + w * lp1).flatten(), (1.0 - w) * s0 + w * s1
    fn.hip_pair = (target, prior)  # lets the samplers below run whole levels of local moves in one launch (sdeng_langevin_moves)
    return fn


# Local moves of a level as ONE HIP launch (csrc/prep_kernels.hip k_langevin_moves) when the annealing path is
# hip_tempered_log_prob_and_grads and the moves are not preconditioned.  NATIVE_NOISE = False: the random numbers still come from torch's
# generator, drawn in the reference's order (the chains are then the reference's chains, tests/golden/smc_*.npz, re_*.npz);
# True: counter-based draws inside the kernel (seeded by NATIVE_SEED and a per-call counter).
NATIVE_MOVES, NATIVE_NOISE, NATIVE_SEED = True, False, 1
_native_calls = [0]


def _native_pair(log_prob_and_grads, x, precond):
    pair = getattr(log_prob_and_grads, "hip_pair", None)
    if not NATIVE_MOVES or pair is None or precond is not None or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32:
        return None
    return pair


def _native_run(pair, t, x, lp, grad, step, n_moves, keep_from, use_ula, target_acceptance, want_samples=True):
    """n_moves local moves of all chains; x / lp / grad / step ([B,d], [B], [B,d], [B] contiguous) are updated in place."""
    _native_calls[0] += 1
    return E.langevin_moves(pair[0], pair[1], x, lp, grad, step, n_moves, t=t, keep_from=keep_from, unadjusted=use_ula,
                            target_acceptance=target_acceptance if not use_ula else 0.0, noise="philox" if NATIVE_NOISE else "torch",
                            seed=(NATIVE_SEED + 0x9E3779B97F4A7C15 * _native_calls[0]) & 0xFFFFFFFFFFFFFFFF, want_samples=want_samples)


def _pmul(mat, v):
    return torch.matmul(mat, v.unsqueeze(-1)).squeeze(-1)


class _LocalMove:
    """One MALA (or ULA) move at a fixed level, plain or preconditioned, with the step-size heuristic of
    additions/mcmc.py:54-72.  State: (x, log-density, gradient, preconditioned gradient or None)."""

    def __init__(self, log_prob_and_grad, use_ula, target_acceptance, precond=None):
        self.f, self.use_ula, self.target, self.precond = log_prob_and_grad, use_ula, target_acceptance, precond

    def pgrad(self, grad):
        return None if self.precond is None else _pmul(self.precond[0], grad)

    def __call__(self, x, lp, grad, pgrad, step):
        log_acc = None
        if self.precond is not None:
            if self.use_ula:
                x, lp, grad, pgrad = precond_ula_step(x, lp, grad, pgrad, self.f, step, *self.precond)
            else:
                x, lp, grad, pgrad, log_acc = precond_mala_step(x, lp, grad, pgrad, self.f, step, *self.precond)
        elif self.use_ula:
            x, lp, grad = ula_step(x, lp, grad, self.f, step)
        else:
            x, lp, grad, log_acc = mala_step(x, lp, grad, self.f, step)
        if log_acc is not None and self.target > 0.0:
            step = heuristics_step_size(step, log_acc, target_acceptance=self.target)
        return x, lp, grad, pgrad, step, log_acc


def smc_sampler(x_init, times, log_prob_and_grads, n_warmup_mcmc_steps, n_mcmc_steps, step_sizes_per_noise, per_noise_init=False,
                reweight_threshold=1.0, use_pdds_weights=False, sde=None, target_acceptance=0.75, precond_matrix_per_noise=None,
                precond_matrix_chol_per_noise=None, use_ula=False, verbose=False):
    """additions/ebm_mle.py:11-195.  Levels are visited from the last (``times[-1]``) to the first; with
    ``reweight_threshold > 0`` the particles carry importance weights between levels and are resampled (multinomial) when
    the normalised ESS drops below the threshold; with ``use_pdds_weights`` they are first moved by the SDE's EI denoising
    kernel and weighted with the forward / backward transition densities (:88-107).
    Returns (samples [n_levels, n_mcmc_steps, B, *data], updated step sizes, diags)."""
    if per_noise_init and reweight_threshold > 0.0:
        raise ValueError("Can't use per_noise_init in SMC mode.")
    if sde is None and use_pdds_weights:
        raise ValueError("Can't use PDDS weights without the SDE object.")
    n_levels = times.shape[0]
    B = x_init.shape[1] if per_noise_init else x_init.shape[0]
    data_shape = x_init.shape[2:] if per_noise_init else x_init.shape[1:]
    smc = reweight_threshold > 0.0
    use_precond = precond_matrix_per_noise is not None and precond_matrix_chol_per_noise is not None
    samples = torch.empty((n_levels, n_mcmc_steps, B, *data_shape), device=x_init.device)
    ess_logs = torch.ones((n_levels,))
    mean_accs = torch.empty((n_levels,))
    log_w = torch.zeros((B,), device=x_init.device)
    x, x_prev, lp_prev, grad_prev = x_init.clone(), None, None, None
    for lvl in range(n_levels - 1, -1, -1):
        first = lvl == n_levels - 1
        x = x_init[lvl].clone() if per_noise_init else x.clone()
        precond = (precond_matrix_per_noise[lvl], precond_matrix_chol_per_noise[lvl]) if use_precond else None
        move = _LocalMove(lambda y, lvl=lvl: log_prob_and_grads(times[lvl], y), use_ula, target_acceptance, precond)
        step = step_sizes_per_noise[lvl]
        lp, grad = move.f(x)
        pgrad = move.pgrad(grad)
        if use_pdds_weights and not first:  # reverse-SDE move from the previous level, then the densities at the moved points
            x, z = sde.ei_integration_step(x_prev, sde.terminal_t - times[lvl + 1], sde.terminal_t - times[lvl], grad_prev)
            lp_backward = -0.5 * torch.sum(torch.square(z), dim=-1)
            mean_f, var_f = sde.transition_params(times[lvl], times[lvl + 1])
            lp_forward = -0.5 * torch.sum(torch.square(mean_f * x - x_prev) / var_f, dim=-1)
            lp, grad = move.f(x)
            pgrad = move.pgrad(grad)
        if smc and not first:
            if use_pdds_weights:
                log_w = lp - lp_prev
                log_w += lp_forward - lp_backward
            else:  # this level's density over the previous level's, at the same points
                log_w += lp - lp_prev
            w = torch.nn.functional.softmax(log_w, dim=0)
            ess = (1.0 / torch.sum(torch.square(w))) / B
            ess_logs[lvl] = ess.cpu().clone()
            if ess < reweight_threshold:
                idx = torch.multinomial(w, B, replacement=True)
                x, lp, grad = x[idx], lp[idx], grad[idx]
                if use_precond:
                    pgrad = pgrad[idx]
                log_w.zero_()
        pair = _native_pair(log_prob_and_grads, x, precond)
        if pair is not None:  # all warm-up and sampling moves of the level in one launch
            x, lp, grad = x.contiguous(), lp.contiguous().clone(), grad.contiguous().clone()
            step_flat = step.to(torch.float32).reshape(-1).expand(B).contiguous().clone()
            got, acc_sum, _ = _native_run(pair, times[lvl].reshape(-1), x, lp, grad, step_flat, n_warmup_mcmc_steps + n_mcmc_steps,
                                          n_warmup_mcmc_steps, use_ula, target_acceptance)
            samples[lvl] = got
            step = step_flat.view(step.shape)
        else:
            for _ in range(n_warmup_mcmc_steps):
                x, lp, grad, pgrad, step, _ = move(x, lp, grad, pgrad, step)
            acc_sum = 0.0
            for i in range(n_mcmc_steps):
                x, lp, grad, pgrad, step, log_acc = move(x, lp, grad, pgrad, step)
                if log_acc is not None:
                    acc_sum = acc_sum + torch.exp(torch.minimum(torch.zeros_like(log_acc), log_acc))
                samples[lvl, i] = x.clone()
        if not use_ula:
            mean_accs[lvl] = (acc_sum / n_mcmc_steps).mean()
        step_sizes_per_noise[lvl] = step.clone()
        x_prev, grad_prev, lp_prev = x.clone(), grad.clone(), lp.clone()
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

    precond = None
    if precond_matrix_per_noise is not None and precond_matrix_chol_per_noise is not None:  # [n_levels, B, d, d] -> one per chain
        precond = (precond_matrix_per_noise.view((-1, *precond_matrix_per_noise.shape[2:])),
                   precond_matrix_chol_per_noise.view((-1, *precond_matrix_chol_per_noise.shape[2:])))
    move = _LocalMove(lambda y: log_prob_and_grads(t_flat, y), use_ula, target_acceptance, precond)
    x = x_init.clone() if per_noise_init else x_init.unsqueeze(0).repeat((n_levels, 1, *ones))
    x = x.view((-1, *data_shape))
    step = step_sizes_per_noise.view((-1, *ones))
    lp, grad = move.f(x)
    pgrad = move.pgrad(grad)
    pairs = make_re_pairings(n_levels, x_init.device)
    native = _native_pair(log_prob_and_grads, x, precond)
    total = n_warmup_mcmc_steps + n_mcmc_steps
    if native is not None:
        x, lp, grad = x.contiguous().clone(), lp.contiguous().clone(), grad.contiguous().clone()
        step_shape, step = step.shape, step.to(torch.float32).reshape(-1).expand(x.shape[0]).contiguous().clone()
    it = 0
    while native is not None and it < total:
        if it % swap_frequency == 0:
            pr = pairs[(it // swap_frequency) % 2]
            xs_, lps_, gs_, mean_swap_acc = re_step(x.view((-1, B, *data_shape)), lp.view((-1, B)), grad.view((-1, B, *data_shape)), batched,
                                                    times, pr[:, 0], pr[:, 1], B, data_shape, ones)
            x, grad, lp = xs_.reshape((-1, *data_shape)).contiguous(), gs_.reshape((-1, *data_shape)).contiguous(), lps_.flatten().contiguous()
            if it >= n_warmup_mcmc_steps:
                samples[:, it - n_warmup_mcmc_steps] = x.reshape((-1, B, *data_shape))
            it += 1
            continue
        run = min(swap_frequency - it % swap_frequency, total - it) if swap_frequency > 0 else total - it  # local moves up to the next swap step
        keep = max(0, n_warmup_mcmc_steps - it)
        got, _, last = _native_run(native, t_flat.reshape(-1), x, lp, grad, step, run, min(keep, run), use_ula, target_acceptance, want_samples=keep < run)
        if keep < run:
            first = it + keep - n_warmup_mcmc_steps
            samples[:, first:first + run - keep] = got.view(run - keep, n_levels, B, *data_shape).transpose(0, 1)
        if last is not None:
            mean_local_accs = last.view((-1, B)).mean(dim=-1)
        it += run
    if native is not None:
        step = step.view(step_shape)
    for it in range(0 if native is None else total, total):
        if it % swap_frequency == 0:
            pr = pairs[(it // swap_frequency) % 2]
            x, lp, grad, mean_swap_acc = re_step(x.view((-1, B, *data_shape)), lp.view((-1, B)), grad.view((-1, B, *data_shape)), batched,
                                                 times, pr[:, 0], pr[:, 1], B, data_shape, ones)
            x, grad, lp = x.view((-1, *data_shape)), grad.view((-1, *data_shape)), lp.flatten()
            pgrad = move.pgrad(grad)
        else:
            x, lp, grad, pgrad, step, log_acc = move(x, lp, grad, pgrad, step)
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
