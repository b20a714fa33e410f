"""Oracle runs of the BASELINE.json workloads (sde_sampler_lrds_amd/experiments/baseline_configs.py) on a block of particles.

TEST INFRASTRUCTURE, like everything under oracle/: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline / log-Z-error leg only -- as the checker, never as the thing measured or shipped.  The functions below only assemble
the restated step loops of oracle/sde_oracle.py (each of which cites the reference lines it follows) with the parameters of a
workload built by baseline_configs: cfg 2 = EIReferenceSDELoss.simulate (losses/oc.py:444-510), cfg 3 =
EMReferenceSDELoss.simulate without reference (:218-296), cfg 4 = ControlledLangevinSDELoss.simulate (:666-755).
"""
from __future__ import annotations

import math

import torch

from . import sde_oracle as orc


def _sd(mod):
    return {k: v.detach().cpu() for k, v in mod.state_dict().items()}


class PerturbedNoise:
    """A noise source moved by +-eps per normal (independent random signs, seeded per step): measures how far a workload amplifies
    the 1.2e-6 difference between the kernel's hardware Box-Muller and libm's (tests/test_gpu_parity.py).  A structured sign
    pattern underestimates it (neighbouring features cancel); random signs are what the hardware error looks like."""

    def __init__(self, base, eps=1.2e-6, salt=0):
        self.base, self.eps, self.salt = base, eps, salt

    def __call__(self, k, x):
        z = self.base(k, x)
        g = torch.Generator().manual_seed(7919 * self.salt + k + 1)
        return z + self.eps * (2.0 * torch.randint(0, 2, z.shape, generator=g).float() - 1.0)


def runner(cfg: str, info: dict, ts: torch.Tensor):
    """-> run(x0_block [b,d] cpu, noise) -> (x_N, rnd, scale) with scale = the largest summand of the log-weights (their error is
    judged relative to it, tests/test_gpu_parity.py rnd_scale)."""
    ts = ts.detach().cpu()
    if cfg == "rds_gmm":
        sde = orc.VP(0.1, 10.0, 1.0, 1.0)
        tgt = orc.GMMDiag(info["target"].loc.cpu(), info["target"].scale.cpu(), info["target"].mixture_weights.cpu())
        ctrl = orc.Ctrl(_sd(info["ctrl"]), "clipped", clip_model=1e4)
        means, var, w = info["means"].cpu(), info["ref_var"].cpu(), info["ref_weights"].cpu()
        loc0, v0 = sde.marginal_diag(torch.tensor(0.0), means, var)
        refd = orc.GMMDiag(loc0, v0.sqrt(), w)

        def run(x0, noise, return_traj=False):
            with torch.no_grad():
                x, rnd, xs = orc.simulate_ei_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp,
                                                 lambda t, xx: orc.mog_score(xx, w, *sde.marginal_diag(t, means, var)), noise,
                                                 return_traj=return_traj)
                scale = max(1.0, float(tgt.logp(x).abs().max()))
                return (x, rnd, scale, xs) if return_traj else (x, rnd, scale)
        return run
    if cfg == "pis_phi4":
        g, T, d = math.sqrt(0.2), 5.0, info["d"]
        sde = orc.ScaledBM(g, T)
        tgt = orc.PhiFour(0.1, 0.0, d, 20.0)
        ctrl = orc.Ctrl(_sd(info["ctrl"]), "score", clip_model=1e4, target_score=tgt.score, clip_score=1e4, scale_score=1.0)
        refd = orc.GaussDiag(torch.zeros(d), torch.full((d,), g * math.sqrt(T)))

        def run(x0, noise):
            with torch.no_grad():
                x, rnd, _ = orc.simulate_em_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, None, noise)
                return x, rnd, max(1.0, float(rnd.abs().max()))
        return run
    if cfg == "cmcd_logreg":
        tgt = orc.LogReg(info["X"], info["y"], 4.5, -2.5, 0.5)
        prior = orc.GaussFull(info["mean"], info["cov"])
        ctrl = orc.Ctrl(_sd(info["ctrl"]), "score", clip_model=1e4, target_score=tgt.score, clip_score=1e4, scale_score=1.0)

        def run(x0, noise):
            with torch.no_grad():
                x, rnd, _ = orc.simulate_cmcd(ts, x0, ctrl, tgt.score, prior.score, 1.0, 1.0, 1e5, tgt.logp, prior.logp, noise)
                return x, rnd, max(1.0, float(rnd.abs().max()))
        return run
    raise ValueError(cfg)


def initial_particles(cfg: str, info: dict, seed: int, particle0: int, b: int) -> torch.Tensor:
    """The x0 the engine draws for this workload's prior: loc + scale * philox_normal(seed, 0, particle0, b, d, stream=1)
    -- IsotropicGauss(scale 1) for cfg 2, Delta(0) for cfg 3, GaussFull(mean, cov) for cfg 4 (MultivariateNormal: loc + L z)."""
    d = info["d"]
    if cfg == "pis_phi4":
        return torch.zeros(b, d)
    z = orc.philox_normal(seed, 0, particle0, b, d, stream=1)
    if cfg == "rds_gmm":
        return z
    return info["mean"] + z @ torch.linalg.cholesky(info["cov"]).T


def log_z(rnd: torch.Tensor) -> float:
    """log_norm_const_is of BaseOCLoss.compute_results (losses/oc.py:150-161) in fp64."""
    r = -rnd.double().flatten()
    return float(torch.logsumexp(r, 0) - math.log(r.numel()))
