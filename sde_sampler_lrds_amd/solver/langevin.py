"""``sde_sampler/solver/langevin.py:15-66`` (LangevinSolver) built from objects instead of a Hydra config: sample the
prior, integrate the Langevin SDE of the target with the Euler integrator (one launch), report the expectation
predictions over the trajectory after the burn-in.  Runs on an MI355X only."""
from __future__ import annotations

import time

import torch

from ..distr.base import EXPECTATION_FNS
from ..eq.integrator import EulerIntegrator, Integrator
from ..eq.sdes import LangevinSDE
from ..utils.common import make_results


class LangevinSolver:
    def __init__(self, target, prior, eval_timesteps, integrator: Integrator | None = None, diff_coeff: float = 1.0,
                 clip_score: float | None = None, eval_batch_size: int = 2000, eval_expectation_burn: int = 0, device="cuda"):
        self.device = torch.device(device)
        self.target, self.prior = target.to(self.device), prior.to(self.device)
        self.integrator = integrator or EulerIntegrator()
        self.sde = LangevinSDE(target_score=self.target.score, diff_coeff=diff_coeff, clip_score=clip_score).to(self.device)
        self.eval_timesteps = eval_timesteps  # callable(device=...) -> ts, like the reference's partial(get_timesteps, ...)
        self.eval_batch_size = eval_batch_size
        self.burn_steps = eval_expectation_burn
        if self.burn_steps >= len(self.eval_timesteps()):
            raise ValueError("Specify more eval_steps than burn_steps.")

    def run(self):
        start_time = time.time()
        x = self.prior.sample((self.eval_batch_size,))
        ts = self.eval_timesteps(device=self.device)
        xs = self.integrator.integrate(self.sde, ts=ts, x_init=x)
        torch.cuda.synchronize(self.device)
        metrics = {"eval/sample_time": time.time() - start_time}
        exp_samples = xs[self.burn_steps:].reshape(-1, self.target.dim)
        expectation_preds = {name: fn(exp_samples).mean() for name, fn in EXPECTATION_FNS.items()}
        return make_results(samples=xs[-1], weights=None, log_norm_const_preds=None, ts=ts, xs=xs, metrics=metrics,
                            expectation_preds=expectation_preds)
