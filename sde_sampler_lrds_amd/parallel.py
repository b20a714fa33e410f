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


# Layout of the 8-float statistics vector sdeng_logz writes (include/sdeng.h) -- the ONE definition the kernel's consumers share:
# engine.logz_stats produces it, combine_stats / global_weights consume it, tests/test_gpu_units.py checks the kernel against
# stats_reference() index by index, and the 2-rank gloo test runs this module's own gather / combine code on it.
ELBO, LOGZ, VAR, ESS, MAX, SUM_EXP, SUM_EXP2, SUM = range(8)
N_STATS = 8


def stats_reference(rnd: torch.Tensor) -> torch.Tensor:
    """What sdeng_logz writes for one shard of log-weights ``rnd`` [B,1], restated with torch in fp64 (CPU or GPU).  Not used by the
    product path (``engine.logz_stats`` is): it pins the layout in tests and lets the gloo test run without a GPU."""
    v = -rnd.double().reshape(-1)
    n = v.numel()
    mx = v.max()
    e = torch.exp(v - mx)
    out = torch.empty(N_STATS, dtype=torch.float64)
    out[ELBO] = v.mean()
    out[LOGZ] = mx + e.sum().log() - math.log(n)
    out[VAR] = rnd.double().var() if n > 1 else 0.0
    out[ESS] = e.sum() ** 2 / (e ** 2).sum() / n
    out[MAX], out[SUM_EXP], out[SUM_EXP2], out[SUM] = mx, e.sum(), (e ** 2).sum(), v.sum()
    return out.float()


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
    gmax = stats[:, MAX].max()
    shift = torch.exp(stats[:, MAX] - gmax)
    se = (stats[:, SUM_EXP] * shift).sum()
    se2 = (stats[:, SUM_EXP2] * shift ** 2).sum()
    total = stats[:, SUM].sum()
    mean = total / n
    local_mean = stats[:, SUM] / counts
    m2 = (stats[:, VAR] * (counts - 1).clamp(min=0)).sum() + (counts * (local_mean - mean) ** 2).sum()
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


def _local_stats(rnd, stats_fn):
    if stats_fn is None:
        from . import engine
        return engine.logz_stats(rnd, want_weights=False)[0]
    return stats_fn(rnd).to(rnd.device)


def global_results_async(rnd: torch.Tensor, dist=None, stats_fn=None) -> PendingResults:
    """Enqueue the estimators over ALL ranks' particles from this rank's ``rnd`` shard [B,1] (device tensor).
    ``stats_fn`` replaces the HIP reduction (``engine.logz_stats``) -- the CPU tests pass ``stats_reference``."""
    stats = _local_stats(rnd, stats_fn)
    # torch.full, not torch.tensor([...], device=...): a host->device copy of pageable memory synchronises the stream and
    # would stop the caller from keeping several passes in flight
    count = torch.full((1,), float(rnd.shape[0]), dtype=stats.dtype, device=rnd.device)
    payload = torch.cat([stats, count])
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return PendingResults(payload.view(1, N_STATS + 1))
    world = dist.get_world_size()
    gathered = torch.empty(world * (N_STATS + 1), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(gathered, payload)
    return PendingResults(gathered.view(world, N_STATS + 1))


def global_results(rnd: torch.Tensor, dist=None, stats_fn=None) -> dict:
    """Estimators over ALL ranks' particles from this rank's ``rnd`` shard [B,1] (device tensor)."""
    return global_results_async(rnd, dist, stats_fn).result()


def global_weights(rnd: torch.Tensor, dist=None, stats_fn=None):
    """This rank's shard of ``Results.weights = softmax(-rnd, 0)`` (losses/oc.py:150-161) normalised over ALL ranks' particles:
    w_i = exp(-rnd_i - M) / S with M, S the global maximum and exponential sum from the same 36-byte all-gather.  Returns
    (weights [B_local, 1], global estimators dict); summed over the ranks the weights give 1."""
    res = global_results(rnd, dist, stats_fn)
    w = torch.exp((-rnd.double() - res["max_neg_rnd"])) / res["sum_exp"]
    return w.to(rnd.dtype).view(-1, 1), res
