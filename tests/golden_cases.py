"""Rebuild the golden cases of tests/golden/*.npz on top of the CPU oracle.

Shared by the oracle-vs-golden tests (CPU) and the HIP-vs-oracle/golden tests (GPU).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from oracle import sde_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SIM_CASES = [
    "rds_ei_gmm_d128_k4", "rds_ei_gmm_d128_k4_n256", "rds_ei_gmm_d128_k16", "rds_ei_gmm_d8_k4",
    "rds_ddpm_gmm_d16_snr", "rds_em_gmm_d16", "rds_em_vp_default_d16", "rds_ei_vp_default_d16",
    "rds_ei_pbm_default_d16", "pis_em_phi4_d100", "dds_two_modes_d2", "dds_rings_d2", "cmcd_logreg_d61", "cmcd_gmm_iso_d16", "cmcd_gmm_diag_d40", "cmcd_phi4_d100", "pis_logreg_d61", "dds_logreg_d61", "dis_ei_d8",
    "dis_orig_lerp_d8", "rds_ei_gmm_fullcov_d128_k4", "rds_em_gmm_fullcov_d40_k3", "rds_ei_gmm_eigen_d16_k3",
    "rds_ei_gauss_fullcov_d40", "dis_ei_cancel_drift_d8", "rds_em_remove_ref_d16", "rds_ei_remove_ref_d40",
    "dds_two_modes_full_d5", "pis_two_modes_full_d20", "pis_gmm_full_d128_k3",
]


EUBO_CASES = ["eubo_ei_gmm_d128_k4", "eubo_ei_gmm_d16_k4", "eubo_em_gmm_d16_k4", "eubo_dis_ei_d8", "eubo_cmcd_gmm_d16",
              "eubo_ei_gmm_fullcov_d40_k3"]  # compute_eubo (noising direction)


EULER_CASES = ["euler_langevin_gmm_d16", "euler_langevin_phi4_d100", "euler_langevin_rings_d2", "euler_ou_vp_d40",
               "euler_controlled_vp_d16"]  # EulerIntegrator.integrate (eq/integrator.py)


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.a = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}

    def params(self, prefix):
        return {k[len(prefix):]: v for k, v in self.a.items() if k.startswith(prefix)}

    def __getitem__(self, k):
        return self.a[k]


def load(name) -> Case:
    return Case(name)


def make_sde(m):
    if m.get("sde", "vp") == "pbm":
        return orc.PinnedBM(m["diff_coeff"], m["T"])
    return orc.VP(m["beta_min"], m["beta_max"], m["sigma"], m["T"])


def run_oracle_eubo(c: Case, noise=None):
    """Oracle restatement of ``compute_eubo`` for an EUBO case -> (noised x, rnd)."""
    m = c.meta
    tgt = orc.GMMDiag(c["tgt_loc"], c["tgt_scale"], c["tgt_w"])
    if m["kind"] == "eubo_cmcd":
        prior = orc.IsoGauss(m["d"], 0.0, m["prior_scale"])
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score, clip_score=m["clip_score"],
                        scale_score=m["scale_score"])
        with torch.no_grad():
            return orc.eubo_cmcd(c["ts"], c["x0"], ctrl, tgt.score, prior.score, m["diff_coeff"], m["T"], m["clip_langevin"], tgt.logp,
                                 prior.logp, noise or orc.PhiloxNoise(m["seed"]))
    sde = make_sde(m)
    if m["kind"] == "eubo_dis":
        prior = orc.IsoGauss(m["d"], 0.0, 1.0)
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score, clip_score=m["clip_score"],
                        scale_score=m["scale_score"])
        with torch.no_grad():
            return orc.eubo_dis_ei(c["ts"], c["x0"], ctrl, sde, tgt.logp, prior.logp, noise or orc.PhiloxNoise(m["seed"]))
    ctrl = orc.Ctrl(c.params("ctrl."), "clipped", clip_model=m["clip_model"])
    means, w = c["ref_means"], c["ref_w"]
    fn = orc.eubo_ei_ref if m["integrator"] == "ei" else orc.eubo_em_ref
    if m.get("cov", "diag") == "full":
        cov = c["ref_cov"]
        refd = orc.GMMFullCov(*sde.marginal_full(torch.tensor(0.0), means, cov), w)
        ref_score = lambda t, x: orc.mog_score_full(x, w, *sde.marginal_full(t, means, cov))  # noqa: E731
    else:
        var = c["ref_vars"]
        loc0, v0 = sde.marginal_diag(torch.tensor(0.0), means, var)
        refd = orc.GMMDiag(loc0, torch.sqrt(v0), w)
        ref_score = lambda t, x: orc.mog_score(x, w, *sde.marginal_diag(t, means, var))  # noqa: E731
    with torch.no_grad():
        return fn(c["ts"], c["x0"], ctrl, sde, tgt.logp, refd.logp, ref_score, noise or orc.PhiloxNoise(m["seed"]))


def run_oracle(c: Case, noise=None, B=None):
    """Run the oracle's restatement of case ``c`` -> (x_N, rnd).  ``noise`` defaults to the
    counter-based replay the fixture was generated with."""
    m = c.meta
    x0 = c["x0"] if B is None else c["x0"][:B]
    ts = c["ts"]
    noise = noise or orc.PhiloxNoise(m["seed"])
    kind = m["kind"]
    if kind == "rds_gmm":
        sde = make_sde(m)
        tgt = orc.GMMDiag(c["tgt_loc"], c["tgt_scale"], c["tgt_w"])
        ctrl = orc.Ctrl(c.params("ctrl."), "clipped", clip_model=m["clip_model"])
        means, w = c["ref_means"], c["ref_w"]
        cov_kind = m.get("cov", "diag")
        if cov_kind == "full":  # covariance matrices [K,d,d] (score_mog_full with linalg.solve)
            cov = c["ref_cov"]
            ref_score = lambda t, x: orc.mog_score_full(x, w, *sde.marginal_full(t, means, cov))  # noqa: E731
            refd = orc.GMMFullCov(*sde.marginal_full(torch.tensor(0.0), means, cov), w)
        elif cov_kind == "eigen":  # (D, P) form: precisions and log-determinants
            D, P = c["ref_D"], c["ref_P"]
            ref_score = lambda t, x: orc.mog_score_full_prec(x, w, *sde.marginal_eigen(t, means, D, P))  # noqa: E731
            refd = orc.GMMFullPrec(*sde.marginal_eigen(torch.tensor(0.0), means, D, P), w)
        else:
            var = c["ref_vars"]

            def ref_score(t, x):
                loc, v = sde.marginal_diag(t, means, var)
                return orc.mog_score(x, w, loc, v)

            loc0, v0 = sde.marginal_diag(torch.tensor(0.0), means, var)
            refd = orc.GMMDiag(loc0, torch.sqrt(v0), w)
        if m.get("remove_ref"):  # RemoveReferenceCtrl(CancelDriftCtrl, ref_score, use_rescaling=False)
            inner = orc.Ctrl(c.params("ctrl."), "cancel_drift", clip_model=m["clip_model"], target_score=tgt.score,
                             clip_score=m["clip_score"], scale_score=m["scale_score"], sde=sde)
            ctrl = orc.RemoveReference(inner, ref_score)
        if m["integrator"] == "em":
            out = orc.simulate_em_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, ref_score, noise)
        else:
            out = orc.simulate_ei_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, ref_score, noise,
                                      ddpm=(m["integrator"] == "ddpm_like"))
    elif kind == "rds_default":
        sde = make_sde(m)
        tgt = orc.GMMDiag(c["tgt_loc"], c["tgt_scale"], c["tgt_w"])
        ctrl = orc.Ctrl(c.params("ctrl."), "clipped", clip_model=m["clip_model"])
        xi, vi = c["ref_x_init"], c["ref_var_init"]
        if m.get("cov", "diag") == "full":  # covariance matrix [d,d]: score_gauss_full (distr/gauss.py:129-135) with linalg.solve
            def full_t(t):
                loc, cv = sde.marginal_full(t, xi.unsqueeze(0), vi.unsqueeze(0))
                return loc[0], cv[0]

            def ref_score(t, x):
                loc, cv = full_t(t)
                return -torch.linalg.solve(cv.unsqueeze(0), (x - loc.unsqueeze(0)).unsqueeze(-1)).squeeze(-1)

            refd = orc.GaussFull(*full_t(torch.tensor(0.0)))
        else:
            def ref_score(t, x):
                loc, v = sde.marginal_diag(t, xi, vi)
                return orc.gauss_score(x, loc, v)

            loc0, v0 = sde.marginal_diag(torch.tensor(0.0), xi, vi)
            refd = orc.GaussDiag(loc0, v0.sqrt())
        if m["integrator"] == "em":
            out = orc.simulate_em_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, ref_score, noise)
        else:
            out = orc.simulate_ei_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, ref_score, noise,
                                      ddpm=(m["integrator"] == "ddpm_like"))
    elif kind == "pis_phi4":
        sde = orc.ScaledBM(m["diff_coeff"], m["T"])
        tgt = orc.PhiFour(m["a"], m["b"], m["d"], m["beta"])
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score,
                        clip_score=m["clip_score"], scale_score=m["scale_score"])
        refd = orc.GaussDiag(c["ref_loc"], c["ref_scale"])
        out = orc.simulate_em_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, None, noise)
    elif kind == "pis_full":  # PIS with a ScoreCtrl on a full-covariance mixture target
        sde = orc.ScaledBM(m["diff_coeff"], m["T"])
        tgt = orc.GMMFullTarget(c["tgt_loc"], c["tgt_cov"], c["tgt_w"])
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score,
                        clip_score=m["clip_score"], scale_score=m["scale_score"])
        refd = orc.GaussDiag(c["ref_loc"], c["ref_scale"])
        out = orc.simulate_em_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, None, noise)
    elif kind in ("dds", "dds_rings", "dds_full"):
        if kind == "dds_rings":
            tgt = orc.Rings(m["lower_rad"], m["upper_rad"], m["num_rad"], m["scale"])
        elif kind == "dds_full":
            tgt = orc.GMMFullTarget(c["tgt_loc"], c["tgt_cov"], c["tgt_w"])
        else:
            tgt = orc.GMMDiag(c["tgt_loc"], c["tgt_scale"], c["tgt_w"])
        prior = orc.IsoGauss(m["d"], 0.0, m["sigma"])
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score,
                        clip_score=m["clip_score"], scale_score=m["scale_score"])
        out = orc.simulate_dds(ts, x0, ctrl, m["alpha"], m["sigma"], tgt.logp, prior.logp, noise)
    elif kind == "cmcd_logreg":
        tgt = orc.LogReg(c["X"], c["y"], m["weight_scale"], m["intercept_mean"], m["intercept_scale"])
        prior = orc.GaussFull(c["prior_loc"], c["prior_cov"])
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score,
                        clip_score=m["clip_score"], scale_score=m["scale_score"])
        out = orc.simulate_cmcd(ts, x0, ctrl, tgt.score, prior.score, m["diff_coeff"], m["T"], m["clip_langevin"],
                                tgt.logp, prior.logp, noise)
    elif kind in ("logreg_pis", "logreg_dds"):
        tgt = orc.LogReg(c["X"], c["y"], m["weight_scale"], m["intercept_mean"], m["intercept_scale"])
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score, clip_score=m["clip_score"],
                        scale_score=m["scale_score"])
        if kind == "logreg_pis":
            sde = orc.ScaledBM(m["diff_coeff"], m["T"])
            refd = orc.GaussDiag(c["ref_loc"], c["ref_scale"])
            out = orc.simulate_em_ref(ts, x0, ctrl, sde, tgt.logp, refd.logp, None, noise)
        else:
            prior = orc.IsoGauss(m["d"], 0.0, m["sigma"])
            out = orc.simulate_dds(ts, x0, ctrl, m["alpha"], m["sigma"], tgt.logp, prior.logp, noise)
    elif kind in ("cmcd_gmm", "cmcd_phi4"):
        if kind == "cmcd_phi4":
            tgt = orc.PhiFour(m["a"], m["b"], m["d"], m["beta"])
            prior = orc.IsoGauss(m["d"], 0.0, m["prior_scale"])
        else:
            tgt = orc.GMMDiag(c["tgt_loc"], c["tgt_scale"], c["tgt_w"])
            prior = orc.IsoGauss(m["d"], 0.0, m["prior_scale"]) if m["prior_kind"] == "iso" else orc.GaussDiag(c["prior_loc"], c["prior_scale_vec"])
        ctrl = orc.Ctrl(c.params("ctrl."), "score", clip_model=m["clip_model"], target_score=tgt.score,
                        clip_score=m["clip_score"], scale_score=m["scale_score"])
        out = orc.simulate_cmcd(ts, x0, ctrl, tgt.score, prior.score, m["diff_coeff"], m["T"], m["clip_langevin"],
                                tgt.logp, prior.logp, noise)
    elif kind in ("dis_ei", "dis_orig"):
        sde = make_sde(m)
        tgt = orc.GMMDiag(c["tgt_loc"], c["tgt_scale"], c["tgt_w"])
        prior = orc.IsoGauss(m["d"], 0.0, 1.0)
        if kind == "dis_ei":
            ctrl = orc.Ctrl(c.params("ctrl."), "cancel_drift" if m.get("cancel_drift") else "score", clip_model=m["clip_model"],
                            target_score=tgt.score, clip_score=m["clip_score"], scale_score=m["scale_score"], sde=sde)
            out = orc.simulate_dis_ei(ts, x0, ctrl, sde, tgt.logp, prior.logp, noise)
        else:
            ctrl = orc.Ctrl(c.params("ctrl."), "lerp", clip_model=m["clip_model"], target_score=tgt.score,
                            clip_score=m["clip_score"], scale_score=m["scale_score"], sde=sde, prior_score=prior.score)
            out = orc.simulate_time_reversal(ts, x0, ctrl, sde, tgt.logp, prior.logp, noise)
    else:
        raise KeyError(kind)
    return out[0], out[1]


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max(1, |b|) elementwise-scaled: relative for large values, absolute near zero."""
    a, b = a.double(), b.double()
    return float(((a - b).abs() / b.abs().clamp(min=1.0)).max())


def euler_increment(c: Case):
    """The Brownian increments an Euler fixture was generated with: bm(s_k, t_k) = philox_normal(seed, k) sqrt(t_k - s_k)."""
    m = c.meta
    return lambda k, s, t, x: orc.philox_normal(m["seed"], k, 0, x.shape[0], x.shape[1]) * torch.sqrt(t - s)


def run_oracle_euler(c: Case, ts=None, increment=None):
    """Oracle restatement of EulerIntegrator.integrate for an Euler case -> states on ``ts`` (default: the fixture's)."""
    m = c.meta
    kind = m["sde_kind"]
    if kind.startswith("langevin"):
        if kind == "langevin_gmm":
            tgt = orc.GMMDiag(c["tgt_loc"], c["tgt_scale"], c["tgt_w"])
        elif kind == "langevin_phi4":
            tgt = orc.PhiFour(m["phi_a"], m["phi_b"], m["d"], m["phi_beta"])
        else:
            tgt = orc.Rings(m["lower_rad"], m["upper_rad"], m["num_rad"], m["scale"])
        g = torch.tensor(m["diff_coeff"], dtype=torch.float32)
        drift = lambda s, x: orc.langevin_sde_drift(x, tgt.score, g, m["clip_score"])  # noqa: E731
        diff = lambda s: g  # noqa: E731
    else:
        sde = make_sde(m)
        ctrl = orc.Ctrl(c.params("ctrl."), "clipped", clip_model=m["clip_model"]) if kind == "controlled_vp" else None
        drift, diff = orc.controlled_sde_drift(sde, ctrl), sde.diff
    with torch.no_grad():
        return orc.euler_integrate(drift, diff, c["ts"] if ts is None else ts, c["x0"], c["timesteps"], increment or euler_increment(c))


SAMPLER_CASES = ["smc_tempered_d3", "smc_annealed_langevin_d3", "re_tempered_d3", "re_ula_d3", "smc_precond_d3", "smc_pdds_d3",
                 "re_precond_d3"]  # additions/ebm_mle.py samplers


def tempered_log_prob_and_grads(t, x):
    """Closed-form annealing path used as the INPUT of the sampler fixtures: N(0, 9 I)^(1-t) * (two Gaussian modes)^t,
    one t per row.  Returns (log-density [B], gradient [B,d])."""
    m, v = 2.0, 0.3
    lp0, g0 = -0.5 * (x ** 2).sum(-1) / 9.0, -x / 9.0
    la, lb = -0.5 * ((x - m) ** 2).sum(-1) / v, -0.5 * ((x + m) ** 2).sum(-1) / v
    lp1 = torch.logsumexp(torch.stack([la, lb]), dim=0)
    ra = torch.exp(la - lp1).unsqueeze(-1)
    g1 = -(ra * (x - m) + (1.0 - ra) * (x + m)) / v
    w = t.reshape(-1, 1)
    return ((1.0 - w[:, 0]) * lp0 + w[:, 0] * lp1), (1.0 - w) * g0 + w * g1


def sampler_inputs(m):
    """x_init, times [n_levels,B,1], step sizes [n_levels,B,1] of a sampler fixture (regenerated from its meta)."""
    g = torch.Generator().manual_seed(m["seed"])
    x_init = 3.0 * torch.randn(m["B"], m["d"], generator=g)
    times = torch.linspace(1.0, 0.0, m["n_levels"]).view(-1, 1, 1).repeat(1, m["B"], 1)  # visited last -> first: prior first
    steps = torch.full((m["n_levels"], m["B"], 1), m["step"])
    if m.get("pdds"):  # PDDS: noise levels increase with the level index, strictly inside (0, T)
        times = torch.linspace(0.05, 0.95, m["n_levels"]).view(-1, 1, 1).repeat(1, m["B"], 1)
    return x_init, times, steps


def sampler_fn(m):
    """The annealing path of a sampler fixture: tempering weight t, or 1 - t when the levels are noise levels (PDDS)."""
    if m.get("pdds"):
        return lambda t, x: tempered_log_prob_and_grads(1.0 - t, x)
    return tempered_log_prob_and_grads


def sampler_precond(m):
    """SPD preconditioners (and their Cholesky factors) of a preconditioned sampler fixture: one per level (SMC, [L,d,d]) or
    one per level and chain (replica exchange, [L,B,d,d]); None for the plain fixtures."""
    if not m.get("precond"):
        return {}
    g = torch.Generator().manual_seed(m["seed"] + 1000)
    lead = (m["n_levels"],) if m["sampler"] == "smc" else (m["n_levels"], m["B"])
    a = torch.randn(*lead, m["d"], m["d"], generator=g)
    P = 0.3 * a @ a.transpose(-1, -2) + torch.eye(m["d"])
    return dict(precond_matrix_per_noise=P, precond_matrix_chol_per_noise=torch.linalg.cholesky(P))
