"""The BASELINE.json configurations as engine workloads: synthetic data of the named shape, random-init weights of
the reference architecture with the last layer made non-trivial (the reference initialises it at ~1e-6,
models/utils.py:7-22, which would make the drift net a no-op).  Shared by bench.py, tools/ and the full-size GPU tests.

Each builder returns ``(loss, ts, x0, args, kwargs, info)``; ``loss.simulate(ts, x0, *args, **kwargs)`` is the pass.
``info['flops']`` = drift-net FLOP per particle-step, 2*(2*64*d + 2*64^2).
"""
from __future__ import annotations

import math

import torch

from ..distr.gauss import Gauss, GaussFull, ManyModes
from ..distr.logistic_regression import LogisticRegression
from ..distr.phi_four import PhiFour
from ..eq.sdes import VP, ControlledLangevinSDE, ScaledBM
from ..losses import oc
from ..models.mlp import FourierMLP, TimeEmbed
from ..models.reparam import ClippedCtrl, ScoreCtrl
from ..reference import MarginalReference
from ..utils.common import get_timesteps


def _net(d):
    net = FourierMLP(dim=d, activation=torch.nn.GELU(), num_layers=4, channels=64)
    with torch.no_grad():
        net.out_layer.weight.uniform_(-0.1, 0.1)
        net.out_layer.bias.uniform_(-0.1, 0.1)
    return net


def _score_ctrl(d, target):
    sm = TimeEmbed(dim_out=1, activation=torch.nn.GELU(), num_layers=4, channels=64)
    with torch.no_grad():
        sm.out_layer.weight.uniform_(-0.02, 0.02)
        sm.out_layer.bias.fill_(0.02)
    return ScoreCtrl(base_model=_net(d), score_model=sm, target_score=target.score, detach_score=False, clip_score=1e4,
                     clip_model=1e4, scale_score=1.0)


def _flops(d):
    return 2 * (2 * 64 * d + 2 * 64 * 64)


def build_rds_gmm(device, B, N, d=128, K=4, seed=1, x_seed=None, ref_var=None, ref_weights=None):
    """configs[1]: ManyModes d=128, RDS with a diagonal-GMM reference, VP(0.1, 10), exponential integrator.
    ``seed`` fixes the model (drift net, reference means); ``x_seed`` (default: ``seed``) the initial particles, so the ranks
    of a sharded run build the SAME sampler and draw different particles."""
    torch.manual_seed(seed)
    sde = VP(0.1, 10.0, 1.0, terminal_t=1.0)
    target = ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0, n_reference_samples=10)
    ctrl = ClippedCtrl(base_model=_net(d), clip_model=1e4)
    means = target.loc.clone() + 0.1 * torch.randn(K, d)
    # reference mixture: SURVEY 8d's spec (variance 0.5, equal weights) unless a test asks for other [K,d] variances / [K] weights
    ref_var = 0.5 * torch.ones(K, d) if ref_var is None else ref_var.clone()
    ref_weights = torch.ones(K) if ref_weights is None else ref_weights.clone()
    ref = MarginalReference(sde, "gmm", means_init=means, variances_init=ref_var.clone(), weights_init=ref_weights.clone())
    for m in (sde, target, ctrl, ref):
        m.to(device)
    loss = oc.EIReferenceSDELoss(ctrl, ctrl, sde=sde, method="kl", reference_ctrl=ref)
    ts = get_timesteps(0.0, 1.0, steps=N).to(device)
    x0 = torch.randn(B, d, generator=torch.Generator().manual_seed(seed if x_seed is None else x_seed)).to(device)
    args = (target.unnorm_log_prob, ref.reference_distr.to(device).log_prob)
    info = dict(sde=sde, target=target, ctrl=ctrl, means=means, ref_var=ref_var, ref_weights=ref_weights, K=K, d=d, flops=_flops(d),
                workload=f"ManyModes d={d} K={K}, RDS gmm-ref, VP(0.1,10), EI integrator")
    return loss, ts, x0, args, {}, info


def build_pis_phi4(device, B, N, d=100, seed=3):
    """configs[2]: PhiFour d=100, PIS (ScoreCtrl with the target score), Euler-Maruyama, ScaledBM(sqrt .2, T=5), x0 = 0."""
    torch.manual_seed(seed)
    g, T = math.sqrt(0.2), 5.0
    sde = ScaledBM(diff_coeff=g, terminal_t=T)
    target = PhiFour(a=0.1, b=0.0, dim=d, beta=20.0)
    ctrl = _score_ctrl(d, target)
    refd = Gauss(dim=d, loc=torch.zeros(d), scale=torch.full((d,), g * math.sqrt(T)))
    for m in (sde, target, ctrl, refd):
        m.to(device)
    loss = oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, T, N + 1, device=device)
    x0 = torch.zeros(B, d, device=device)
    info = dict(sde=sde, target=target, ctrl=ctrl, refd=refd, d=d, flops=_flops(d), workload=f"PhiFour d={d}, PIS, EM integrator")
    return loss, ts, x0, (target.unnorm_log_prob, refd.log_prob), {}, info


def synthetic_sonar(seed=7):
    """Design matrix of the sonar data's shape and range (data/sonar.pkl is a pickle and is not loaded)."""
    gen = torch.Generator().manual_seed(seed)
    X = (1e-4 + (1 - 1e-4) * torch.rand(166, 60, generator=gen) ** 2).float()
    y = (torch.rand(166, generator=gen) < 0.47).float()
    return X, y


def build_cmcd_logreg(device, B, N, seed=4):
    """configs[3], one GPU's shard: logistic regression d=61, CMCD, GaussFull prior, g=1, T=1."""
    torch.manual_seed(seed)
    X, y = synthetic_sonar()
    target = LogisticRegression(X, y, intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5)
    d = target.dim
    A = torch.randn(d, d)
    cov = 0.01 * A @ A.T + 0.5 * torch.eye(d)
    mean = 0.1 * torch.randn(d)
    prior = GaussFull(dim=d, loc=mean, cov=cov)
    sde = ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0, terminal_t=1.0, clip_score=1e5)
    ctrl = _score_ctrl(d, target)
    for m in (target, prior, sde, ctrl):
        m.to(device)
    loss = oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, 1.0, N + 1, device=device)
    x0 = (mean + torch.randn(B, d, generator=torch.Generator().manual_seed(seed)) @ torch.linalg.cholesky(cov).T).to(device)
    info = dict(target=target, prior=prior, ctrl=ctrl, X=X, y=y, mean=mean, cov=cov, d=d, flops=_flops(d),
                workload="LogisticRegression d=61 (sonar-shaped synthetic data), CMCD, GaussFull prior")
    return loss, ts, x0, (target.unnorm_log_prob,), dict(initial_log_prob=prior.log_prob, train=False), info


BUILDERS = {"rds_gmm": build_rds_gmm, "pis_phi4": build_pis_phi4, "cmcd_logreg": build_cmcd_logreg}
FULL_SIZE = {"rds_gmm": (65536, 256), "pis_phi4": (131072, 512), "cmcd_logreg": (65536, 256)}  # BASELINE.json (cfg 4: one GPU's shard of 262 144)


def prior_of(cfg, info, device):
    """The prior the reference's solver samples x0 from for this workload (solver/oc.py:132): IsotropicGauss(scale 1) for the VP
    RDS solver (conf/solver/vp_rds.yaml), Delta for PIS (conf/solver/pis.yaml), the GaussFull of CMCD.update_prior (solver/oc.py:291-303)."""
    from ..distr.delta import Delta
    from ..distr.gauss import IsotropicGauss
    if cfg == "rds_gmm":
        return IsotropicGauss(dim=info["d"], scale=1.0).to(device)
    if cfg == "pis_phi4":
        return Delta(dim=info["d"]).to(device)
    return info["prior"]
