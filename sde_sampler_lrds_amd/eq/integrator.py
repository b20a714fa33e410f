"""``sde_sampler/eq/integrator.py`` on the engine: ``EulerIntegrator`` (:80-129) and ``interpolate`` (:66-77).

The whole Euler-Maruyama loop is ONE launch that writes all N+1 states (``engine.euler_states``); the states are then
interpolated onto the caller's grid ``ts`` with the reference's rule.  ``TorchSDEIntegrator`` (:24-63) wraps the
third-party ``torchsde`` solvers and is not provided.  There is no CPU implementation: tensors must live on an MI355X.
"""
from __future__ import annotations

import torch

from .. import engine as E
from ..utils.common import get_timesteps


class Integrator:
    def integrate(self, sde, ts, x_init, timesteps=None, bm=None):
        raise NotImplementedError


def interpolate(ts, s, t, xs, xt, eps: float = 1e-8):
    """eq/integrator.py:66-77: the entries of ``ts`` up to ``t + eps`` linearly interpolated between ``xs`` and ``xt``."""
    ind = torch.searchsorted(ts, t + eps, side="right")
    t_eval = ts[:ind]
    assert (s <= t_eval).all() and (t_eval <= t + eps).all()
    return torch.lerp(xs, xt, (t_eval.view(-1, 1, 1) - s) / (t - s))


def interpolate_states(ts, timesteps, states, eps: float = 1e-8):
    """The bookkeeping of eq/integrator.py:110-128 for all steps at once: every ``ts[j]`` is emitted by the first step
    ``(s, t)`` with ``ts[j] <= t + eps``, as ``lerp(x_s, x_t, (ts[j] - s) / (t - s))``."""
    ts = ts.to(states.device, torch.float32)
    grid = timesteps.to(states.device, torch.float32)
    k = torch.searchsorted((grid[1:] + eps).contiguous(), ts.contiguous(), side="left")
    if int(k.max()) >= grid.numel() - 1:
        raise AssertionError("ts reaches past the integration grid (eq/integrator.py:128 asserts the same)")
    s, t = grid[k], grid[k + 1]
    assert bool((s <= ts).all())
    return torch.lerp(states[k], states[k + 1], ((ts - s) / (t - s)).view(-1, 1, 1))


class EulerIntegrator(Integrator):
    """eq/integrator.py:80-129.  ``seed``: key of the engine's counter-based noise; every ``integrate`` call without an
    explicit ``bm`` draws a fresh stream (the reference consumes torch's global generator)."""

    def __init__(self, dt: float | None = 0.01, steps: int | None = None, rescale_t: str | None = None, eps: float = 1e-8,
                 seed: int = 0):
        self.dt, self.steps, self.rescale_t, self.eps = dt, steps, rescale_t, eps
        self.seed, self.calls, self.particle0 = seed, 0, 0

    def integrate(self, sde, ts, x_init, timesteps=None, bm=None, snr_adapted: bool = False):
        if timesteps is None:
            timesteps = get_timesteps(ts[0], ts[-1], dt=self.dt, steps=self.steps, rescale_t=self.rescale_t, device=ts.device,
                                      sde=sde if snr_adapted else None)
        increments = None
        if bm is not None:  # bm(s, t): the Brownian increment over [s, t]
            increments = torch.stack([bm(s, t) for s, t in zip(timesteps[:-1], timesteps[1:])])
        seed = (int(self.seed) + 0x9E3779B97F4A7C15 * self.calls) & 0xFFFFFFFFFFFFFFFF
        self.calls += 1
        states = E.euler_states(sde, timesteps, x_init, increments=increments, seed=seed, particle0=self.particle0)
        return interpolate_states(ts, timesteps, states, eps=self.eps)
