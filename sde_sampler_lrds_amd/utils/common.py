"""Host-side helpers mirroring ``sde_sampler/utils/common.py`` of the reference
(Results :9-13, binary_search_v :18-27, get_timesteps :30-82, clip_and_log :85-112)."""
from __future__ import annotations

import math
from collections import namedtuple

import torch

# same field order and defaults as the reference (the shared mutable default dicts are NOT reproduced:
# each Results gets fresh dicts through `make_results`)
Results = namedtuple(
    "Results",
    "samples weights log_norm_const_preds expectation_preds ts xs metrics plots",
    defaults=[None, None, None, None, None, None, None, None],
)

CKPT_DIR = "ckpt"


def make_results(**kw) -> Results:
    for key in ("log_norm_const_preds", "expectation_preds", "metrics", "plots"):
        kw.setdefault(key, {})
    return Results(**kw)


def binary_search_v(f, low, high, target_value, n_attemps):
    """Vectorised bisection on a decreasing-or-increasing scalar map (reference :18-27)."""
    lo, hi = low, high
    for _ in range(n_attemps):
        mid = (lo + hi) / 2.0
        val = f(mid)
        lo = torch.where(val > target_value, mid, lo)
        hi = torch.where(val <= target_value, mid, hi)
    return (lo + hi) / 2.0


def get_timesteps(start, end, dt=None, steps=None, rescale_t=None, n_attemps=1024, sde=None, device=None):
    """Time grid [steps+1] (uniform / quad / cosine / SNR-adapted).  The cosine grid has steps+2 points with
    a last increment of 0, exactly like the reference (:62-80)."""
    if (steps is None) is (dt is None):
        raise ValueError("Exactly one of `dt` and `steps` should be defined.")
    if steps is None:
        steps = int(math.ceil((end - start) / dt))
    if sde is not None:
        dev = sde.terminal_t.device
        lo_snr, hi_snr = sde.log_snr(start), sde.log_snr(end)
        if torch.isnan(lo_snr):
            raise ValueError("NaN SNR at t_0")
        if torch.isnan(hi_snr):
            raise ValueError("NaN SNR at t_K")
        levels = torch.linspace(lo_snr, hi_snr, steps=steps + 1, device=dev)
        inner = binary_search_v(sde.log_snr, start, end, levels[1:-1], n_attemps=n_attemps)
        grid = torch.concat([torch.FloatTensor([start]).to(dev), inner, torch.FloatTensor([end]).to(dev)], dim=0)
        return grid.sort().values
    if rescale_t is None:
        return torch.linspace(start, end, steps=steps + 1, device=device)
    if rescale_t == "quad":
        return torch.sqrt(torch.linspace(start, end.square(), steps=steps + 1, device=device)).clip(max=end)
    if rescale_t == "cosine":
        s = 0.008
        phase = ((torch.linspace(start, end, steps + 1, device=device) / end + s) / (1 + s)) * torch.pi * 0.5
        dts = torch.cos(phase) ** 4
        dts /= dts.sum()
        dts *= end
        return torch.concat((torch.tensor([start], device=device), torch.cumsum(dts, -1)))
    raise ValueError("Unkown timestep rescaling method.")


def clip_and_log(tensor, max_norm=None, name=None, t=None, log_dt=0.2):
    if max_norm is not None:
        tensor = tensor.clip(min=-1.0 * max_norm, max=max_norm)
    return tensor
