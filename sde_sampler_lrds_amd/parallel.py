"""Multi-GPU layer: particles are i.i.d. and never interact before the final estimator, so the batch is
sharded across ranks (one process per GPU) with NO collective inside the step loop.  The only exchange is
for the log-Z / ESS estimators of ``BaseOCLoss.compute_results`` (losses/oc.py:150-161): every rank reduces
its own shard on the device (``sdeng_logz``), the 8-float partial-statistics vectors are all-gathered
(RCCL over xGMI; 32 bytes per rank, latency-bound) and combined with a max-shifted log-sum-exp and Chan's
parallel variance.  Noise is keyed by the global particle index, so results do not depend on the sharding.
"""
from __future__ import annotations

import math

import torch


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous particle range [lo, hi) of ``rank`` (first ``total % world`` ranks get one extra)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def combine_stats(stats: torch.Tensor, counts: torch.Tensor) -> dict:
    """stats [W,8] as written by sdeng_logz (include/sdeng.h), counts [W] -> global estimators.
    stats[:,2] = unbiased var(rnd), [:,4] = max(-rnd), [:,5] = sum exp(-rnd-max), [:,6] = sum exp(2(-rnd-max)),
    [:,7] = sum(-rnd)."""
    stats, counts = stats.double().cpu(), counts.double().cpu()
    n = counts.sum()
    gmax = stats[:, 4].max()
    shift = torch.exp(stats[:, 4] - gmax)
    se = (stats[:, 5] * shift).sum()
    se2 = (stats[:, 6] * shift ** 2).sum()
    total = stats[:, 7].sum()
    mean = total / n
    local_mean = stats[:, 7] / counts
    m2 = (stats[:, 2] * (counts - 1).clamp(min=0)).sum() + (counts * (local_mean - mean) ** 2).sum()
    return {
        "elbo": float(mean),
        "log_norm_const_is": float(gmax + torch.log(se) - math.log(float(n))),
        "lv_loss": float(m2 / (n - 1)) if n > 1 else 0.0,
        "ess": float(se * se / se2 / n),
        "n": int(n), "max_neg_rnd": float(gmax), "sum_exp": float(se),
    }


class PendingResults:
    """Estimators of one pass, still on the device: every kernel (and the all-gather) is enqueued, nothing has been
    read back.  ``result()`` synchronises and combines on the host.  Lets a caller keep several passes in flight."""

    def __init__(self, gathered: torch.Tensor):
        self.gathered = gathered  # [W, 9]: sdeng_logz statistics + particle count per rank

    def result(self) -> dict:
        g = self.gathered
        return combine_stats(g[:, :8], g[:, 8])


def global_results_async(rnd: torch.Tensor, dist=None) -> PendingResults:
    """Enqueue the estimators over ALL ranks' particles from this rank's ``rnd`` shard [B,1] (device tensor)."""
    from . import engine
    stats, _ = engine.logz_stats(rnd, want_weights=False)
    # torch.full, not torch.tensor([...], device=...): a host->device copy of pageable memory synchronises the stream and
    # would stop the caller from keeping several passes in flight
    count = torch.full((1,), float(rnd.shape[0]), dtype=stats.dtype, device=rnd.device)
    payload = torch.cat([stats, count])
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return PendingResults(payload.view(1, 9))
    world = dist.get_world_size()
    gathered = torch.empty(world * 9, dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(gathered, payload)
    return PendingResults(gathered.view(world, 9))


def global_results(rnd: torch.Tensor, dist=None) -> dict:
    """Estimators over ALL ranks' particles from this rank's ``rnd`` shard [B,1] (device tensor)."""
    return global_results_async(rnd, dist).result()
