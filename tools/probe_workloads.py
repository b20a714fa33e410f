"""Step-loop kernel time on the other BASELINE.json configurations (random-init weights, synthetic data):
   cfg 3: PhiFour d=100, PIS (ScoreCtrl with the target score), Euler-Maruyama, 131 072 particles x 512 steps
   cfg 4: logistic regression d=61 (sonar-shaped synthetic design matrix), CMCD, 65 536 particles x 256 steps (one GPU's shard)
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd.distr.gauss import Gauss, GaussFull  # noqa: E402
from sde_sampler_lrds_amd.distr.logistic_regression import LogisticRegression  # noqa: E402
from sde_sampler_lrds_amd.distr.phi_four import PhiFour  # noqa: E402
from sde_sampler_lrds_amd.eq.sdes import ControlledLangevinSDE, ScaledBM  # noqa: E402
from sde_sampler_lrds_amd.losses import oc  # noqa: E402
from sde_sampler_lrds_amd.models.mlp import FourierMLP, TimeEmbed  # noqa: E402
from sde_sampler_lrds_amd.models.reparam import ScoreCtrl  # noqa: E402


def score_ctrl(d, target):
    net = FourierMLP(dim=d, activation=torch.nn.GELU(), num_layers=4, channels=64)
    sm = TimeEmbed(dim_out=1, activation=torch.nn.GELU(), num_layers=4, channels=64)
    with torch.no_grad():
        net.out_layer.weight.uniform_(-0.1, 0.1)
        net.out_layer.bias.uniform_(-0.1, 0.1)
        sm.out_layer.weight.uniform_(-0.02, 0.02)
        sm.out_layer.bias.fill_(0.02)
    return ScoreCtrl(base_model=net, score_model=sm, target_score=target.score, detach_score=False, clip_score=1e4,
                     clip_model=1e4, scale_score=1.0)


def build_pis_phi4(dev, B, N, d=100):
    torch.manual_seed(3)
    g, T = math.sqrt(0.2), 5.0
    sde = ScaledBM(diff_coeff=g, terminal_t=T)
    target = PhiFour(a=0.1, b=0.0, dim=d, beta=20.0)
    ctrl = score_ctrl(d, target)
    refd = Gauss(dim=d, loc=torch.zeros(d), scale=torch.full((d,), g * math.sqrt(T)))
    for m in (sde, target, ctrl, refd):
        m.to(dev)
    loss = oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, T, N + 1, device=dev)
    x0 = torch.zeros(B, d, device=dev)
    return loss, ts, x0, (target.unnorm_log_prob, refd.log_prob), {}


def build_cmcd_logreg(dev, B, N):
    torch.manual_seed(4)
    gen = torch.Generator().manual_seed(7)
    X = (1e-4 + (1 - 1e-4) * torch.rand(166, 60, generator=gen) ** 2).float()
    y = (torch.rand(166, generator=gen) < 0.47).float()
    target = LogisticRegression(X, y, intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5)
    d = target.dim
    A = torch.randn(d, d)
    cov = 0.01 * A @ A.T + 0.5 * torch.eye(d)
    mean = 0.1 * torch.randn(d)
    prior = GaussFull(dim=d, loc=mean, cov=cov)
    sde = ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0, terminal_t=1.0, clip_score=1e5)
    ctrl = score_ctrl(d, target)
    for m in (target, prior, sde, ctrl):
        m.to(dev)
    loss = oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, 1.0, N + 1, device=dev)
    x0 = (mean + torch.randn(B, d) @ torch.linalg.cholesky(cov).T).to(dev)
    return loss, ts, x0, (target.unnorm_log_prob,), dict(initial_log_prob=prior.log_prob, train=False)


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["pis", "cmcd"]
    for name in which:
        if name == "pis":
            B, N, d = 131072, 512, 100
            loss, ts, x0, args, kw = build_pis_phi4(dev, B, N)
        else:
            B, N, d = 65536, 256, 61
            loss, ts, x0, args, kw = build_cmcd_logreg(dev, B, N)
        loss.seed = 1
        ev = L.HipEvents()
        loss.timing_events = ev
        for rep in range(3):
            x, rnd, _ = loss.simulate(ts, x0, *args, **kw)
            torch.cuda.synchronize()
            ms = ev.elapsed_ms()
        fl = 2 * (2 * 64 * d + 2 * 64 * 64)
        print(f"{name}: B={B} N={N} d={d}: kernel {ms:.2f} ms -> {B*N/(ms*1e-3):.3e} p-steps/s ({fl*B*N/(ms*1e-3)/1e12:.1f} TFLOP/s alg.)  "
              f"rnd mean {rnd.mean().item():.4f} finite {bool(torch.isfinite(rnd).all())}", flush=True)
