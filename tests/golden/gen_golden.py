"""Generate golden input/output vectors by running the REAL reference
(``/root/reference``, vanilladucky/sde_sampler_lrds) in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Only this script touches the reference; the resulting ``*.npz`` fixtures are data
(inputs + expected outputs) and are what travels to the GPU box.

How the reference is brought up: ``wandb``, ``torchquad`` and
``torchsde`` are imported by reference modules but never used on this path, and are not
installed here, so empty stand-in modules are registered for those three names.
``sde_sampler.solver.*`` needs hydra/omegaconf (absent) and is not imported: each solver
is assembled by hand exactly as its ``setup_models`` does (solver/oc.py:43-78, 358-378,
438-452, 273-289, 499-511) with the constants of ``conf/``.

Noise: the reference draws one ``torch.randn_like(x)`` per step.  While a case runs,
``torch.randn_like`` is replaced by a replay of the engine's counter-based noise
definition (``oracle.sde_oracle.philox_normal(seed, step, ...)``), so a fixture only has to
store the seed, and every implementation consumes bit-identical normals.

``data/sonar.pkl`` is a pickle and ``torch.load(weights_only=True)`` refuses it, so it is
NOT loaded: the logistic-regression case uses a synthetic design matrix of the same shape
and value range (X [166,60] in [1e-4,1], y in {0,1}) and the reference's
``LogisticRegression`` arithmetic methods on it (the constructor, which unpickles, is bypassed).
"""
from __future__ import annotations

import json
import math
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
for _n in ("wandb", "torchquad", "torchsde"):
    sys.modules.setdefault(_n, types.ModuleType(_n))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import sde_oracle as orc  # noqa: E402  (only for the noise definition)

from sde_sampler.distr import base as r_base  # noqa: E402
from sde_sampler.distr import delta as r_delta  # noqa: E402
from sde_sampler.distr import gauss as r_gauss  # noqa: E402
from sde_sampler.distr import logistic_regression as r_lr  # noqa: E402
from sde_sampler.distr import phi_four as r_phi  # noqa: E402
from sde_sampler.distr import rings as r_rings  # noqa: E402
from sde_sampler.eq import integrator as r_int  # noqa: E402
from sde_sampler.eq import sdes as r_sdes  # noqa: E402
from sde_sampler.losses import oc as r_oc  # noqa: E402
from sde_sampler.models import mlp as r_mlp  # noqa: E402
from sde_sampler.models import reparam as r_rep  # noqa: E402
from sde_sampler.models import utils as r_mu  # noqa: E402
from sde_sampler.utils.common import get_timesteps as r_get_timesteps  # noqa: E402


# ----------------------------------------------------------------------------- helpers
class Replay:
    """Stand-in for torch.randn_like during a reference run: step k -> philox_normal(seed, k)."""

    def __init__(self, seed):
        self.seed, self.k = seed, 0

    def __call__(self, x, *a, **kw):
        z = orc.philox_normal(self.seed, self.k, 0, x.shape[0], x.shape[1])
        self.k += 1
        return z


def run_with_replay(seed, fn):
    orig = torch.randn_like
    rep = Replay(seed)
    torch.randn_like = rep
    try:
        with torch.no_grad():
            out = fn()
    finally:
        torch.randn_like = orig
    return out, rep.k


def fourier_mlp(dim):
    return r_mlp.FourierMLP(dim=dim, activation=torch.nn.GELU(), num_layers=4, channels=64,
                            last_bias_init=r_mu.init_bias_uniform_zeros,
                            last_weight_init=r_mu.kaiming_uniform_zeros_)


def score_time_embed(bias=0.0):
    m = r_mlp.TimeEmbed(dim_out=1, activation=torch.nn.GELU(), num_layers=4, channels=64,
                        last_bias_init=r_mu.init_bias_uniform_zeros,
                        last_weight_init=r_mu.kaiming_uniform_zeros_)
    with torch.no_grad():
        m.out_layer.weight.uniform_(-0.02, 0.02)
        m.out_layer.bias.fill_(bias)
    return m


def liven(net, scale=0.1):
    """The reference initialises the last layer at ~1e-6 so the net is a no-op at init;
    re-randomise it so that the drift net matters in the fixture."""
    with torch.no_grad():
        net.out_layer.weight.uniform_(-scale, scale)
        net.out_layer.bias.uniform_(-scale, scale)
    return net


def sd(module):
    return {k: v.detach().clone().numpy() for k, v in module.state_dict().items()}


def save(name, meta, arrays):
    path = os.path.join(HERE, name + ".npz")
    arrays = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KB  logZ={meta.get('log_norm_const_is')}")


def finish(name, meta, arrays, res, nsteps_drawn):
    meta = dict(meta)
    meta["log_norm_const_is"] = res.log_norm_const_preds["log_norm_const_is"]
    meta["elbo"] = res.metrics["eval/elbo"]
    meta["lv_loss"] = res.metrics["eval/lv_loss"]
    meta["draws"] = nsteps_drawn
    arrays = dict(arrays)
    arrays["out_x"] = res.samples
    arrays["out_weights"] = res.weights
    save(name, meta, arrays)


def pack_params(prefix, state):
    return {f"{prefix}{k}": v for k, v in state.items()}


# ----------------------------------------------------------------------------- cases
def case_rds_gmm(name, d, K, B, N, seed, integrator="ei", time_type="uniform", beta_max=10.0, t_end=None, cov="diag", remove_ref=False):
    """RDS with a diagonal-GMM reference (solver/oc.py:563-576), VP noising, basic model
    (conf/solver/vp_rds.yaml, conf/model/basic.yaml)."""
    torch.manual_seed(seed)
    sde = r_sdes.VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=beta_max, scale_diff_coeff=1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0,
                               n_reference_samples=10)
    ctrl = r_rep.ClippedCtrl(base_model=liven(fourier_mlp(d)), clip_model=1e4)
    if remove_ref:  # Langevin init on a reference solver: CancelDriftCtrl (conf/model/langevin_init.yaml) under RemoveReferenceCtrl (below)
        ctrl = r_rep.CancelDriftCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=1.0), target_score=target.score,
                                     detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0, sde=sde, langevin_init=True)
    means = target.loc.clone() + 0.1 * torch.randn(K, d)
    variances = 0.5 * torch.ones(K, d) * (1.0 + 0.2 * torch.rand(K, d))
    weights = torch.ones(K)
    cov_arrays = dict(ref_vars=variances)
    if cov != "diag":  # full covariance matrices (score_mog_full, distr/gauss.py:110-121), or their (D, P) eigen form (eq/sdes.py:228-238)
        A = torch.randn(K, d, d) / d ** 0.5
        full = 0.3 * A @ A.transpose(-1, -2) + 0.4 * torch.eye(d)
        weights = torch.rand(K) + 0.5
        if cov == "full":
            variances, cov_arrays = full, dict(ref_cov=full)
        else:
            D, P = torch.linalg.eigh(full)
            variances, cov_arrays = (D, P), dict(ref_D=D, ref_P=P)
    ref_utils = dict(means_init=means, variances_init=variances, weights_init=weights)
    ref_distr = sde.marginal_gmm_distr(t=torch.tensor(0.0), **ref_utils)
    ref_ctrl = lambda t, x: sde.marginal_gmm_score(t=t, x=x, **ref_utils)  # noqa: E731
    cls = {"ei": r_oc.EIReferenceSDELoss, "ddpm_like": r_oc.DDPMLikeReferenceSDELoss,
           "em": r_oc.EMReferenceSDELoss}[integrator]
    inner = ctrl
    if remove_ref:  # models/reparam.py:46-64 in the form its forward can evaluate (use_rescaling=False): ctrl - ref_score
        ctrl = r_rep.RemoveReferenceCtrl(inner, ref_ctrl, use_rescaling=False)
    loss = cls(ctrl, ctrl, sde=sde, method="kl", reference_ctrl=ref_ctrl)
    if time_type == "snr":
        ts = r_get_timesteps(1e-4, 1.0 - 1e-4, steps=N, sde=sde)
    else:
        ts = r_get_timesteps(0.0, 1.0 if t_end is None else t_end, steps=N)
    x0 = orc.philox_normal(seed, 0, 0, B, d, stream=1)  # IsotropicGauss prior, scale 1
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob, ref_distr.log_prob,
                                                         compute_weights=True, return_traj=True, use_ema=False))
    # step-0 intermediates
    with torch.no_grad():
        T = ts[-1]
        u0 = ctrl(T - ts[0], x0)
        r0 = ref_ctrl(T - ts[0], x0)
    with torch.no_grad():
        (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(
            ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob, reference_log_prob=ref_distr.log_prob))
    meta = dict(kind="rds_gmm", integrator=integrator, d=d, K=K, B=B, N=N, seed=seed, beta_min=0.1, beta_max=beta_max,
                sigma=1.0, T=1.0, clip_model=1e4, time_type=time_type, cov=cov)
    if remove_ref:
        meta.update(remove_ref=True, clip_score=1e4, scale_score=1.0)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], u0=u0, ref0=r0,
                  tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights,
                  ref_means=means, ref_w=weights, **cov_arrays, **pack_params("ctrl.", sd(inner)))
    finish(name, meta, arrays, res, draws)


def case_eubo_gmm(name, d, K, B, N, seed, integrator, cov="diag"):
    """compute_eubo of the RDS losses (losses/oc.py:298-362 EM, :512-568 EI): noising trajectories started at target
    samples, diagonal-GMM reference, same set-up as case_rds_gmm."""
    torch.manual_seed(seed)
    sde = r_sdes.VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, scale_diff_coeff=1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0, n_reference_samples=10)
    ctrl = r_rep.ClippedCtrl(base_model=liven(fourier_mlp(d)), clip_model=1e4)
    means = target.loc.clone() + 0.1 * torch.randn(K, d)
    variances, weights = 0.5 * torch.ones(K, d), torch.ones(K)
    cov_arrays = dict(ref_vars=variances)
    if cov == "full":  # full covariance matrices (score_mog_full)
        A = torch.randn(K, d, d) / d ** 0.5
        variances = 0.3 * A @ A.transpose(-1, -2) + 0.4 * torch.eye(d)
        weights = torch.rand(K) + 0.5
        cov_arrays = dict(ref_cov=variances)

    def reference_ctrl(t, x):
        return sde.marginal_gmm_score(t, x, means, variances, weights)

    ref_distr = sde.marginal_gmm_distr(torch.tensor(0.0), means, variances, weights)
    cls = {"ei": r_oc.EIReferenceSDELoss, "em": r_oc.EMReferenceSDELoss}[integrator]
    loss = cls(ctrl, ctrl, sde=sde, method="kl", reference_ctrl=reference_ctrl)
    ts = r_get_timesteps(0.0, 1.0, steps=N)
    # "samples from the target": mode centres + noise, from the engine's counter-based stream (stream 1)
    comp = torch.arange(B) % K
    x0 = target.loc[comp] + math.sqrt(0.5) * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    rnd, draws = run_with_replay(seed, lambda: loss.compute_eubo(ts, x0.clone(), target.unnorm_log_prob, ref_distr.log_prob))
    # compute_eubo noises x in place: rerun on a copy to record the end point
    xc = x0.clone()
    run_with_replay(seed, lambda: loss.compute_eubo(ts, xc, target.unnorm_log_prob, ref_distr.log_prob))
    meta = dict(kind="eubo_gmm", d=d, K=K, B=B, N=N, seed=seed, integrator=integrator, beta_min=0.1, beta_max=10.0, sigma=1.0, T=1.0,
                clip_model=1e4, draws=draws, cov=cov)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, out_x=xc, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights,
                  ref_means=means, ref_w=weights, **cov_arrays, **pack_params("ctrl.", sd(ctrl)))
    save(name, meta, arrays)


def case_eubo_dis(name, d, K, B, N, seed):
    """DiscreteTimeReversalLossEI.compute_eubo (losses/oc.py:980-1036), same set-up as case_dis(kind='ei')."""
    torch.manual_seed(seed)
    sde = r_sdes.VP(0.1, 10.0, 1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=1.0)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.2),
                           target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    loss = r_oc.DiscreteTimeReversalLossEI(ctrl, ctrl, sde=sde, method="kl")
    ts = r_get_timesteps(0.0, 1.0, steps=N)
    comp = torch.arange(B) % K
    x0 = target.loc[comp] + math.sqrt(0.5) * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    xc = x0.clone()
    rnd, draws = run_with_replay(seed, lambda: loss.compute_eubo(ts, xc, target.unnorm_log_prob, initial_log_prob=prior.log_prob))
    meta = dict(kind="eubo_dis", d=d, K=K, B=B, N=N, seed=seed, beta_min=0.1, beta_max=10.0, sigma=1.0, T=1.0, clip_model=1e4,
                clip_score=1e4, scale_score=1.0, draws=draws)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, out_x=xc, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights,
                  **pack_params("ctrl.", sd(ctrl)))
    save(name, meta, arrays)


def case_train_lv(name, d, K, B, N, seed, integrator, method="lv"):
    """One log-variance training evaluation of the RDS losses (losses/oc.py:364-394 with method='lv'): loss value and
    the gradient w.r.t. every drift-net parameter under the replayed noise."""
    torch.manual_seed(seed)
    sde = r_sdes.VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, scale_diff_coeff=1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0, n_reference_samples=10)
    ctrl = r_rep.ClippedCtrl(base_model=liven(fourier_mlp(d)), clip_model=1e4)
    means = target.loc.clone() + 0.1 * torch.randn(K, d)
    variances, weights = 0.5 * torch.ones(K, d), torch.ones(K)

    def reference_ctrl(t, x):
        return sde.marginal_gmm_score(t, x, means, variances, weights)

    ref_distr = sde.marginal_gmm_distr(torch.tensor(0.0), means, variances, weights)
    cls = {"ei": r_oc.EIReferenceSDELoss, "em": r_oc.EMReferenceSDELoss, "ddpm_like": r_oc.DDPMLikeReferenceSDELoss}[integrator]
    loss = cls(ctrl, ctrl, sde=sde, method=method, reference_ctrl=reference_ctrl)
    ts = r_get_timesteps(0.0, 1.0, steps=N)
    x0 = orc.philox_normal(seed, 0, 0, B, d, stream=1)
    orig = torch.randn_like
    rep = Replay(seed)
    torch.randn_like = rep
    try:
        value, _ = loss(ts, x0.clone(), target.unnorm_log_prob, ref_distr.log_prob)
    finally:
        torch.randn_like = orig
    value.backward()
    grads = {f"grad.{k}": p.grad.detach().clone() for k, p in ctrl.named_parameters()}
    meta = dict(kind="train_lv", d=d, K=K, B=B, N=N, seed=seed, integrator=integrator, beta_min=0.1, beta_max=10.0, sigma=1.0, T=1.0,
                clip_model=1e4, loss=float(value), draws=rep.k, method=method)
    arrays = dict(ts=ts, x0=x0, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights, ref_means=means,
                  ref_vars=variances, ref_w=weights, **pack_params("ctrl.", sd(ctrl)), **grads)
    save(name, meta, arrays)


def _train_fixture(name, meta, arrays, ctrl, call):
    """Run ``call()`` (a reference loss __call__) under the replayed noise, back-propagate, save loss + gradients."""
    orig = torch.randn_like
    rep = Replay(meta["seed"])
    torch.randn_like = rep
    try:
        value, _ = call()
    finally:
        torch.randn_like = orig
    value.backward()
    grads = {f"grad.{k}": p.grad.detach().clone() for k, p in ctrl.named_parameters() if p.grad is not None}
    # How well-conditioned is this gradient?  The same call with every normal moved by +-1.2e-6 (what separates the kernel's hardware
    # Box-Muller from libm's; random signs): the largest relative change of any parameter's gradient, in the REFERENCE's own arithmetic.
    # Tests of an fp32 implementation assert max(5e-5, 10 x this).
    from oracle.baseline_oracles import PerturbedNoise
    sens = 0.0
    for salt in range(2):
        for p in ctrl.parameters():
            p.grad = None
        base = Replay(meta["seed"])
        pert = PerturbedNoise(lambda k, x: orc.philox_normal(meta["seed"], k, 0, x.shape[0], x.shape[1]), salt=salt)
        state = {"k": 0}

        def draw(x, *a, **kw):
            zz = pert(state["k"], x)
            state["k"] += 1
            return zz
        torch.randn_like = draw
        try:
            v2, _ = call()
        finally:
            torch.randn_like = orig
        v2.backward()
        for k, p in ctrl.named_parameters():
            if p.grad is not None and f"grad.{k}" in grads:
                g0 = grads[f"grad.{k}"]
                sens = max(sens, float((p.grad - g0).abs().max()) / max(float(g0.abs().max()), 1e-6))
    save(name, dict(meta, loss=float(value), draws=rep.k, grad_sensitivity=sens), dict(arrays, **pack_params("ctrl.", sd(ctrl)), **grads))


def case_train_lv_dis(name, d, K, B, N, seed, method="lv"):
    """DiscreteTimeReversalLossEI.__call__ (losses/oc.py:1038-1066), method='lv', ScoreCtrl on a mixture target."""
    torch.manual_seed(seed)
    sde = r_sdes.VP(0.1, 10.0, 1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=1.0)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.2), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    loss = r_oc.DiscreteTimeReversalLossEI(ctrl, ctrl, sde=sde, method=method)
    ts = r_get_timesteps(0.0, 1.0, steps=N)
    x0 = orc.philox_normal(seed, 0, 0, B, d, stream=1)
    meta = dict(method=method, kind="train_lv_dis", d=d, K=K, B=B, N=N, seed=seed, beta_min=0.1, beta_max=10.0, sigma=1.0, T=1.0, clip_model=1e4,
                clip_score=1e4, scale_score=1.0)
    arrays = dict(ts=ts, x0=x0, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights)
    _train_fixture(name, meta, arrays, ctrl, lambda: loss(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob))


def case_train_lv_dis_orig(name, d, K, B, N, seed, method="lv"):
    """TimeReversalLoss.__call__ (losses/oc.py:1240-1272), method='lv', LerpCtrl, no inference control (dis_orig)."""
    torch.manual_seed(seed)
    sde = r_sdes.VP(0.1, 10.0, 1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=1.0)
    ctrl = r_rep.LerpCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=1.0), target_score=target.score,
                          detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0, sde=sde, prior_score=prior.score)
    loss = r_oc.TimeReversalLoss(ctrl, ctrl, sde=sde, method=method, inference_ctrl=None)
    ts = r_get_timesteps(0.0, 1.0, steps=N)
    x0 = orc.philox_normal(seed, 0, 0, B, d, stream=1)
    meta = dict(method=method, kind="train_lv_dis_orig", d=d, K=K, B=B, N=N, seed=seed, beta_min=0.1, beta_max=10.0, sigma=1.0, T=1.0, clip_model=1e4,
                clip_score=1e4, scale_score=1.0)
    arrays = dict(ts=ts, x0=x0, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights)
    _train_fixture(name, meta, arrays, ctrl, lambda: loss(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob))


def case_train_lv_cmcd(name, d, K, B, N, seed, method="lv"):
    """ControlledLangevinSDELoss.__call__ (losses/oc.py:830-857), method='lv' (or 'kl': back-propagation through simulate(train=True),
    rnd0 = 0, :695-699), mixture target, IsotropicGauss prior."""
    torch.manual_seed(seed)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=2.0)
    sde = r_sdes.ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0, terminal_t=1.0, clip_score=1e5)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.01), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    loss = r_oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method=method, max_rnd=1e8)
    ts = torch.linspace(0.0, 1.0, N + 1)
    x0 = 2.0 * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    meta = dict(kind="train_lv_cmcd", d=d, K=K, B=B, N=N, seed=seed, diff_coeff=1.0, T=1.0, clip_langevin=1e5, clip_model=1e4,
                clip_score=1e4, scale_score=1.0, prior_kind="iso", prior_scale=2.0, method=method)
    arrays = dict(ts=ts, x0=x0, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights)
    _train_fixture(name, meta, arrays, ctrl, lambda: loss(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob))


def case_train_lv_dds(name, d, B, seed, dt=0.4, end=6.4, sigma=1.0, method="lv"):
    """ExponentialIntegratorSDELoss.__call__ (losses/oc.py:1399-1428), method='lv', TwoModes target."""
    torch.manual_seed(seed)
    target = r_gauss.TwoModes(dim=d, a=1.0, ill_conditioned="not", n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=sigma)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.01), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    loss = r_oc.ExponentialIntegratorSDELoss(ctrl, ctrl, sde=None, method=method, alpha=1.0, sigma=sigma)
    ts = r_get_timesteps(0.0, end, dt=dt, rescale_t="cosine")
    x0 = sigma * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    meta = dict(method=method, kind="train_lv_dds", d=d, B=B, N=len(ts) - 1, seed=seed, alpha=1.0, sigma=sigma, clip_model=1e4, clip_score=1e4,
                scale_score=1.0, dt=dt, end=end)
    arrays = dict(ts=ts, x0=x0, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights)
    _train_fixture(name, meta, arrays, ctrl, lambda: loss(ts, x0.clone(), target.unnorm_log_prob, prior.log_prob))


def case_train_lv_pis(name, d, B, N, seed, dt, method="lv"):
    """EMReferenceSDELoss.__call__ without a reference (PIS, losses/oc.py:364-394), method='lv', phi^4 target."""
    torch.manual_seed(seed)
    g, Tstar = math.sqrt(0.2), 5.0
    sde = r_sdes.ScaledBM(diff_coeff=g, terminal_t=Tstar)
    target = r_phi.PhiFour(a=0.1, b=0.0, dim=d, dim_phys=1, beta=20.0)
    prior = r_delta.Delta(dim=d)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.02), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    ref_distr = sde.marginal_distr(t=sde.terminal_t, x_init=prior.loc)
    loss = r_oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method=method, max_rnd=1e8 if method == "lv" else None)
    ts = torch.linspace(0.0, N * dt, N + 1)
    x0 = prior.sample((B,))
    meta = dict(method=method, kind="train_lv_pis", d=d, B=B, N=N, seed=seed, diff_coeff=g, T=Tstar, a=0.1, b=0.0, beta=20.0, clip_model=1e4,
                clip_score=1e4, scale_score=1.0)
    arrays = dict(ts=ts, x0=x0, ref_loc=ref_distr.loc, ref_scale=ref_distr.scale)
    _train_fixture(name, meta, arrays, ctrl, lambda: loss(ts, x0.clone(), target.unnorm_log_prob, ref_distr.log_prob))


def case_rds_default(name, d, K, B, N, seed, sde_kind="vp", integrator="em", full=False):
    """RDS with the default Gaussian reference (solver/oc.py:535-551): VP + IsotropicGauss prior, or
    PinnedBM + Delta prior (conf/solver/vp_rds.yaml, pbm_rds.yaml)."""
    torch.manual_seed(seed)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    ctrl = r_rep.ClippedCtrl(base_model=liven(fourier_mlp(d)), clip_model=1e4)
    if sde_kind == "vp":
        sde = r_sdes.VP(0.1, 10.0, 1.0, terminal_t=1.0)
        x_init, var_init = torch.zeros(d), torch.ones(d)
        if full:  # ref_type='gaussian' with a covariance matrix (solver/oc.py:551-562, score_gauss_full distr/gauss.py:129-135)
            A = torch.randn(d, d) / d ** 0.5
            x_init, var_init = 0.5 * torch.randn(d), 0.4 * A @ A.T + 0.6 * torch.eye(d)
        ts = r_get_timesteps(0.0, 1.0, steps=N)
        x0 = orc.philox_normal(seed, 0, 0, B, d, stream=1)
        sde_meta = dict(sde="vp", beta_min=0.1, beta_max=10.0, sigma=1.0, T=1.0)
    else:
        g, T = math.sqrt(0.2), 5.0
        sde = r_sdes.PinnedBM(diff_coeff=g, terminal_t=T)
        x_init = torch.zeros(d)
        var_init = sde.terminal_t * sde.diff_coeff ** 2 * torch.ones(d)
        ts = r_get_timesteps(1e-4, T - 1e-4, steps=N, sde=sde)
        x0 = torch.zeros(B, d)
        sde_meta = dict(sde="pbm", diff_coeff=g, T=T)
    ref_distr = sde.marginal_distr(t=torch.tensor(0.0), x_init=x_init, var_init=var_init)
    ref_ctrl = lambda t, x: sde.marginal_score(t=t, x=x, x_init=x_init, var_init=var_init)  # noqa: E731
    cls = {"ei": r_oc.EIReferenceSDELoss, "ddpm_like": r_oc.DDPMLikeReferenceSDELoss,
           "em": r_oc.EMReferenceSDELoss}[integrator]
    loss = cls(ctrl, ctrl, sde=sde, method="kl", reference_ctrl=ref_ctrl)
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob, ref_distr.log_prob,
                                                         compute_weights=True, return_traj=True, use_ema=False))
    (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(
        ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob, reference_log_prob=ref_distr.log_prob))
    meta = dict(kind="rds_default", integrator=integrator, d=d, K=K, B=B, N=N, seed=seed, clip_model=1e4, cov="full" if full else "diag",
                **sde_meta)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], tgt_loc=target.loc, tgt_scale=target.scale,
                  tgt_w=target.mixture_weights, ref_x_init=x_init, ref_var_init=var_init,
                  **pack_params("ctrl.", sd(ctrl)))
    finish(name, meta, arrays, res, draws)


def case_pis_phi4(name, d, B, N, seed, dt):
    """PIS on PhiFour (conf/solver/pis.yaml, conf/model/score.yaml, conf/target/phi_four.yaml,
    solver/oc.py:358-378): ScaledBM(sqrt .2, T=5), Delta prior, ScoreCtrl."""
    torch.manual_seed(seed)
    g, Tstar = math.sqrt(0.2), 5.0
    sde = r_sdes.ScaledBM(diff_coeff=g, terminal_t=Tstar)
    target = r_phi.PhiFour(a=0.1, b=0.0, dim=d, dim_phys=1, beta=20.0)
    prior = r_delta.Delta(dim=d)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.02),
                           target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                           scale_score=1.0)
    ref_distr = sde.marginal_distr(t=sde.terminal_t, x_init=prior.loc)
    loss = r_oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, N * dt, N + 1)
    x0 = prior.sample((B,))
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob, ref_distr.log_prob,
                                                         compute_weights=True, return_traj=True, use_ema=False))
    (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(
        ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob, reference_log_prob=ref_distr.log_prob))
    with torch.no_grad():
        xm = res.xs[N // 2]
        u_mid = ctrl(ts[-1] - ts[N // 2], xm)
    meta = dict(kind="pis_phi4", d=d, B=B, N=N, seed=seed, diff_coeff=g, T=Tstar, a=0.1, b=0.0, beta=20.0,
                clip_model=1e4, clip_score=1e4, scale_score=1.0)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], x_mid=xm, u_mid=u_mid,
                  ref_loc=ref_distr.loc, ref_scale=ref_distr.scale, **pack_params("ctrl.", sd(ctrl)))
    finish(name, meta, arrays, res, draws)


def _full_cov_target(d, K, seed):
    """GMMFull / TwoModesFull target (distr/gauss.py:310-520).  K = 0: the reference's own TwoModesFull(dim=d) (conf/target/two_modes_full.yaml);
    K > 0: a GMMFull with K random means and well-conditioned random covariances."""
    if K == 0:
        return r_gauss.TwoModesFull(dim=d, a=1.0, ill_conditioned="medium", n_reference_samples=10)
    g = torch.Generator().manual_seed(seed + 1000)
    loc = 1.5 * torch.randn(K, d, generator=g)
    a = torch.randn(K, d, d, generator=g) / math.sqrt(d)
    cov = 0.6 * torch.matmul(a, a.transpose(1, 2)) + 0.4 * torch.eye(d)
    return r_gauss.GMMFull(dim=d, loc=loc, cov=cov, mixture_weights=torch.rand(K, generator=g) + 0.5, n_reference_samples=10)


def case_pis_full(name, d, K, B, N, seed, dt):
    """PIS (EMReferenceSDELoss without a reference drift, solver/oc.py:358-378) with a target-informed ScoreCtrl on a FULL-covariance
    mixture target: score_mog_full (distr/gauss.py:110-121) inside the step loop."""
    torch.manual_seed(seed)
    g, Tstar = math.sqrt(0.2), N * dt
    sde = r_sdes.ScaledBM(diff_coeff=g, terminal_t=Tstar)
    target = _full_cov_target(d, K, seed)
    prior = r_delta.Delta(dim=d)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.02),
                           target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                           scale_score=1.0)
    ref_distr = sde.marginal_distr(t=sde.terminal_t, x_init=prior.loc)
    loss = r_oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, N * dt, N + 1)
    x0 = prior.sample((B,))
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob, ref_distr.log_prob,
                                                         compute_weights=True, return_traj=True, use_ema=False))
    (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(
        ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob, reference_log_prob=ref_distr.log_prob))
    meta = dict(kind="pis_full", d=d, K=int(target.loc.shape[0]), B=B, N=N, seed=seed, diff_coeff=g, T=Tstar, clip_model=1e4, clip_score=1e4,
                scale_score=1.0)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], ref_loc=ref_distr.loc, ref_scale=ref_distr.scale,
                  tgt_loc=target.loc, tgt_cov=target.cov, tgt_w=target.mixture_weights, **pack_params("ctrl.", sd(ctrl)))
    finish(name, meta, arrays, res, draws)


def case_dds(name, d, B, seed, dt=0.4, end=6.4, sigma=1.0, rings=False, full=False):
    """DDS on TwoModes or Rings (conf/solver/dds.yaml, conf/loss/exponential_sde.yaml, conf/target/rings.yaml,
    solver/oc.py:438-452)."""
    torch.manual_seed(seed)
    if rings:
        target = r_rings.Rings(dim=2, n_reference_samples=10)
    elif full:
        target = _full_cov_target(d, 0, seed)
    else:
        target = r_gauss.TwoModes(dim=d, a=1.0, ill_conditioned="not", n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=sigma)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.01),
                           target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                           scale_score=1.0)
    loss = r_oc.ExponentialIntegratorSDELoss(ctrl, ctrl, sde=None, method="kl", alpha=1.0, sigma=sigma)
    ts = r_get_timesteps(0.0, end, dt=dt, rescale_t="cosine")
    x0 = sigma * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob, prior.log_prob,
                                                         compute_weights=True, return_traj=True, use_ema=False))
    (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(
        ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob, reference_log_prob=prior.log_prob,
        compute_ito_int=True))
    meta = dict(kind="dds_rings" if rings else ("dds_full" if full else "dds"), d=d, B=B, N=len(ts) - 1, seed=seed, alpha=1.0, sigma=sigma,
                clip_model=1e4, clip_score=1e4, scale_score=1.0, dt=dt, end=end)
    if rings:
        meta.update(lower_rad=1.0, upper_rad=5.0, num_rad=3, scale=0.1)
        tgt_arrays = dict(rings_rad=target.radiuses, rings_w=target.radius_dist.mixture_distribution.probs)
    elif full:
        tgt_arrays = dict(tgt_loc=target.loc, tgt_cov=target.cov, tgt_w=target.mixture_weights)
    else:
        tgt_arrays = dict(tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], **tgt_arrays, **pack_params("ctrl.", sd(ctrl)))
    finish(name, meta, arrays, res, draws)


class SyntheticLogReg(r_lr.LogisticRegression):
    """The reference's LogisticRegression with its file-reading constructor bypassed
    (distr/logistic_regression.py:14-39 restated for in-memory data); every arithmetic
    method (posterior_log_prob, the autograd score of distr/base.py:146-154) is the reference's own."""

    def __init__(self, X, y, intercept_mean, intercept_scale, weight_scale):
        r_base.Distribution.__init__(self, dim=X.shape[1] + 1)
        self.X_train, self.y_train = X.float(), y.float().flatten()
        self.X_test, self.y_test = self.X_train, self.y_train
        dw = X.shape[1]
        self.threshold = 1e-8
        self.register_buffer("weight_scale", torch.tensor(weight_scale), persistent=False)
        self.weights_prior = torch.distributions.Independent(
            torch.distributions.Normal(loc=torch.zeros((dw,)), scale=self.weight_scale * torch.ones((dw,))), 1)
        self.use_intercept = True
        self.register_buffer("intercept_mean", torch.tensor(intercept_mean), persistent=False)
        self.register_buffer("intercept_scale", torch.tensor(intercept_scale), persistent=False)
        self.intercept_prior = torch.distributions.Normal(loc=self.intercept_mean, scale=self.intercept_scale)


def synthetic_sonar(seed=7):
    g = torch.Generator().manual_seed(seed)
    X = (1e-4 + (1 - 1e-4) * torch.rand(166, 60, generator=g) ** 2).float()
    y = (torch.rand(166, generator=g) < 0.47).float()
    return X, y


def case_cmcd_logreg(name, B, N, seed, dt):
    """CMCD on (synthetic-)sonar logistic regression (conf/solver/cmcd.yaml, conf/target/sonar.yaml,
    solver/oc.py:273-303): ControlledLangevinSDE(g=1, T=1, clip 1e5), GaussFull prior, ScoreCtrl."""
    torch.manual_seed(seed)
    X, y = synthetic_sonar()
    target = SyntheticLogReg(X, y, intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5)
    d = target.dim
    A = torch.randn(d, d)
    cov = 0.01 * A @ A.T + 0.5 * torch.eye(d)
    mean = 0.1 * torch.randn(d)
    prior = r_gauss.GaussFull(dim=d, loc=mean, cov=cov)
    sde = r_sdes.ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0,
                                       terminal_t=1.0, clip_score=1e5)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.01),
                           target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                           scale_score=1.0)
    loss = r_oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, N * dt, N + 1)
    L = torch.linalg.cholesky(cov)
    x0 = mean + orc.philox_normal(seed, 0, 0, B, d, stream=1) @ L.T

    def ev():
        return loss.eval(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob,
                         compute_weights=True, return_traj=True, use_ema=False)

    # Distribution.score needs autograd, so this case runs without the no_grad wrapper of run_with_replay
    orig = torch.randn_like
    rep = Replay(seed)
    torch.randn_like = rep
    try:
        res = ev()
        draws = rep.k
        rep.k = 0
        x_n, rnd, _ = loss.simulate(ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob,
                                    initial_log_prob=prior.log_prob, train=False)
    finally:
        torch.randn_like = orig
    xq = torch.cat([x0[:16], 5.0 * x0[:8], 30.0 * x0[:8]])
    sc = target.score(xq.clone())
    lp = target.unnorm_log_prob(xq).detach()
    meta = dict(kind="cmcd_logreg", d=d, B=B, N=N, seed=seed, diff_coeff=1.0, T=1.0, clip_langevin=1e5, clip_model=1e4,
                clip_score=1e4, scale_score=1.0, intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5,
                data="synthetic sonar-shaped (sonar.pkl not loadable with a safe loader)")
    arrays = dict(ts=ts, x0=x0, rnd=rnd.detach(), xs_last2=res.xs[-2:].detach(), X=X, y=y, prior_loc=mean,
                  prior_cov=cov, score_x=xq, score_out=sc.detach(), logp_out=lp,
                  prior_logp_x0=prior.log_prob(x0).detach(), prior_score_x0=prior.score(x0).detach(),
                  **pack_params("ctrl.", sd(ctrl)))
    res = res._replace(samples=res.samples.detach(), weights=res.weights.detach())
    finish(name, meta, arrays, res, draws)


def case_eubo_cmcd(name, d, K, B, N, seed):
    """ControlledLangevinSDELoss.compute_eubo (losses/oc.py:757-828) on a mixture target, IsotropicGauss prior."""
    torch.manual_seed(seed)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=2.0)
    sde = r_sdes.ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0, terminal_t=1.0, clip_score=1e5)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.01), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    loss = r_oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, 1.0, N + 1)
    comp = torch.arange(B) % K
    x0 = target.loc[comp] + math.sqrt(0.5) * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    rnd, draws = run_with_replay(seed, lambda: loss.compute_eubo(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob))
    meta = dict(kind="eubo_cmcd", d=d, K=K, B=B, N=N, seed=seed, diff_coeff=1.0, T=1.0, clip_langevin=1e5, clip_model=1e4, clip_score=1e4,
                scale_score=1.0, prior_kind="iso", prior_scale=2.0, draws=draws)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights,
                  **pack_params("ctrl.", sd(ctrl)))
    save(name, meta, arrays)


def case_cmcd_phi4(name, d, B, N, seed):
    """CMCD on the phi^4 lattice (conf/solver/cmcd.yaml with conf/target/phi_four.yaml), IsotropicGauss prior."""
    torch.manual_seed(seed)
    target = r_phi.PhiFour(a=0.1, b=0.0, dim=d, dim_phys=1, beta=20.0)
    prior = r_gauss.IsotropicGauss(dim=d, scale=0.5)
    x0 = 0.5 * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    sde = r_sdes.ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0, terminal_t=1.0, clip_score=1e5)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d), scale=0.02), score_model=score_time_embed(bias=0.001), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    loss = r_oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, 1.0, N + 1)
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob,
                                                         compute_weights=True, return_traj=True, use_ema=False))
    (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob,
                                                                   train=False))
    meta = dict(kind="cmcd_phi4", d=d, B=B, N=N, seed=seed, diff_coeff=1.0, T=1.0, clip_langevin=1e5, clip_model=1e4, clip_score=1e4,
                scale_score=1.0, a=0.1, b=0.0, beta=20.0, prior_scale=0.5)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], **pack_params("ctrl.", sd(ctrl)))
    finish(name, meta, arrays, res, draws)


def case_cmcd_gmm(name, d, K, B, N, seed, prior_kind="iso"):
    """CMCD on a Gaussian-mixture target (conf/solver/cmcd.yaml with conf/target/many_modes.yaml): IsotropicGauss(5) prior
    (benchmark_utils.py cmcd defaults) or a diagonal Gauss prior (update_prior with a variance vector, solver/oc.py:291-303)."""
    torch.manual_seed(seed)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    if prior_kind == "iso":
        prior = r_gauss.IsotropicGauss(dim=d, scale=5.0)
        x0 = 5.0 * orc.philox_normal(seed, 0, 0, B, d, stream=1)
        extra = dict(prior_scale=5.0)
    else:
        mean, var = 0.3 * torch.randn(d), 2.0 + torch.rand(d)
        prior = r_gauss.Gauss(dim=d, loc=mean, scale=var.sqrt())
        x0 = mean + var.sqrt() * orc.philox_normal(seed, 0, 0, B, d, stream=1)
        extra = {}
    sde = r_sdes.ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0, terminal_t=1.0, clip_score=1e5)
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.01), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    loss = r_oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    ts = torch.linspace(0.0, 1.0, N + 1)
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob,
                                                         compute_weights=True, return_traj=True, use_ema=False))
    (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(ts, x0.clone(), target.unnorm_log_prob, initial_log_prob=prior.log_prob,
                                                                   train=False))
    meta = dict(kind="cmcd_gmm", d=d, K=K, B=B, N=N, seed=seed, diff_coeff=1.0, T=1.0, clip_langevin=1e5, clip_model=1e4, clip_score=1e4,
                scale_score=1.0, prior_kind=prior_kind, **extra)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights,
                  **pack_params("ctrl.", sd(ctrl)))
    if prior_kind != "iso":
        arrays.update(prior_loc=prior.loc.flatten(), prior_scale_vec=prior.scale.flatten())
    finish(name, meta, arrays, res, draws)


def case_logreg_ctrl(name, B, N, seed, solver):
    """ScoreCtrl on the (synthetic-)sonar logistic regression with the non-CMCD solvers the reference benchmarks run on the
    Bayesian targets: solver='pis' (EMReferenceSDELoss, ScaledBM, Delta prior) or 'dds' (ExponentialIntegratorSDELoss)."""
    torch.manual_seed(seed)
    X, y = synthetic_sonar()
    target = SyntheticLogReg(X, y, intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5)
    d = target.dim
    ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d), scale=0.05), score_model=score_time_embed(bias=0.01), target_score=target.score,
                           detach_score=False, clip_score=1e4, clip_model=1e4, scale_score=1.0)
    if solver == "pis":
        g, Tstar = 1.0, 1.0
        sde = r_sdes.ScaledBM(diff_coeff=g, terminal_t=Tstar)
        prior = r_delta.Delta(dim=d)
        ref_distr = sde.marginal_distr(t=sde.terminal_t, x_init=prior.loc)
        loss = r_oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
        ts = torch.linspace(0.0, Tstar, N + 1)
        x0 = prior.sample((B,))
        meta_extra = dict(diff_coeff=g, T=Tstar)
        arr_extra = dict(ref_loc=ref_distr.loc, ref_scale=ref_distr.scale)
        kw = dict()
        ref_logp = ref_distr.log_prob
    else:
        sigma = 1.0
        prior = r_gauss.IsotropicGauss(dim=d, scale=sigma)
        loss = r_oc.ExponentialIntegratorSDELoss(ctrl, ctrl, sde=None, method="kl", alpha=1.0, sigma=sigma)
        ts = r_get_timesteps(0.0, 3.2, dt=0.1, rescale_t="cosine")
        x0 = sigma * orc.philox_normal(seed, 0, 0, B, d, stream=1)
        meta_extra = dict(alpha=1.0, sigma=sigma, dt=0.1, end=3.2)
        arr_extra = dict()
        kw = dict(compute_ito_int=True)
        ref_logp = prior.log_prob
    # Distribution.score needs autograd: no no_grad wrapper
    orig = torch.randn_like
    rep = Replay(seed)
    torch.randn_like = rep
    try:
        res = loss.eval(ts, x0.clone(), target.unnorm_log_prob, ref_logp, compute_weights=True, return_traj=True, use_ema=False)
    finally:
        torch.randn_like = orig
    draws = rep.k
    rep = Replay(seed)
    torch.randn_like = rep
    try:
        x_n, rnd, _ = loss.simulate(ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob, reference_log_prob=ref_logp, **kw)
    finally:
        torch.randn_like = orig
    meta = dict(kind="logreg_" + solver, d=d, B=B, N=len(ts) - 1, seed=seed, clip_model=1e4, clip_score=1e4, scale_score=1.0,
                intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5, **meta_extra)
    arrays = dict(ts=ts, x0=x0, rnd=rnd.detach(), xs_last2=res.xs[-2:].detach(), X=X, y=y, **arr_extra, **pack_params("ctrl.", sd(ctrl)))
    res = res._replace(samples=res.samples.detach(), weights=res.weights.detach())
    finish(name, meta, arrays, res, draws)


def case_dis(name, d, K, B, N, seed, kind, cancel_drift=False):
    """DIS: kind='ei' -> DiscreteTimeReversalLossEI with ScoreCtrl; kind='orig' -> TimeReversalLoss with
    LerpCtrl (conf/solver/dis.yaml, conf/model/lerp.yaml, solver/oc.py:185-261)."""
    torch.manual_seed(seed)
    sde = r_sdes.VP(0.1, 10.0, 1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=1.0)
    if kind == "ei" and cancel_drift:  # conf/model/langevin_init.yaml: CancelDriftCtrl (models/reparam.py:120-145)
        ctrl = r_rep.CancelDriftCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=1.0),
                                     target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                                     scale_score=1.0, sde=sde, langevin_init=True)
        loss = r_oc.DiscreteTimeReversalLossEI(ctrl, ctrl, sde=sde, method="kl")
    elif kind == "ei":
        ctrl = r_rep.ScoreCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=0.2),
                               target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                               scale_score=1.0)
        loss = r_oc.DiscreteTimeReversalLossEI(ctrl, ctrl, sde=sde, method="kl")
    else:
        ctrl = r_rep.LerpCtrl(base_model=liven(fourier_mlp(d)), score_model=score_time_embed(bias=1.0),
                              target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                              scale_score=1.0, sde=sde, prior_score=prior.score)
        loss = r_oc.TimeReversalLoss(ctrl, ctrl, sde=sde, method="kl", inference_ctrl=None)
    ts = r_get_timesteps(0.0, 1.0, steps=N)
    x0 = orc.philox_normal(seed, 0, 0, B, d, stream=1)
    res, draws = run_with_replay(seed, lambda: loss.eval(ts, x0.clone(), target.unnorm_log_prob,
                                                         initial_log_prob=prior.log_prob, compute_weights=True,
                                                         return_traj=True, use_ema=False))
    kw = dict(train=False) if kind == "ei" else dict(train=False, compute_ito_int=True)
    (x_n, rnd, _), _ = run_with_replay(seed, lambda: loss.simulate(
        ts, x0.clone(), terminal_unnorm_log_prob=target.unnorm_log_prob, initial_log_prob=prior.log_prob, **kw))
    meta = dict(kind="dis_" + kind, d=d, K=K, B=B, N=N, seed=seed, beta_min=0.1, beta_max=10.0, sigma=1.0, T=1.0,
                clip_model=1e4, clip_score=1e4, scale_score=1.0, cancel_drift=cancel_drift)
    arrays = dict(ts=ts, x0=x0, rnd=rnd, xs_last2=res.xs[-2:], tgt_loc=target.loc, tgt_scale=target.scale,
                  tgt_w=target.mixture_weights, **pack_params("ctrl.", sd(ctrl)))
    finish(name, meta, arrays, res, draws)


def unit_vectors():
    """Isolated known-answer vectors: scores / log-probs / SDE scalars / time grids / net forward."""
    torch.manual_seed(123)
    out, meta = {}, {}
    # GMM score + log-prob (distr/gauss.py:97-107, 217-221)
    tgt = r_gauss.ManyModes(n_modes=5, dim=12, var=0.5, n_reference_samples=10)
    x = 3.0 * torch.randn(40, 12)
    out.update(gmm_loc=tgt.loc, gmm_scale=tgt.scale, gmm_w=tgt.mixture_weights.clone(), gmm_x=x,
               gmm_score=tgt.score(x.clone()), gmm_logp=tgt.unnorm_log_prob(x))
    # PhiFour (distr/phi_four.py:54-96)
    phi = r_phi.PhiFour(a=0.1, b=0.3, dim=20, beta=20.0)
    xp = 0.8 * torch.randn(16, 20)
    out.update(phi_x=xp, phi_score=phi.score(xp), phi_logp=phi.unnorm_log_prob(xp))
    meta["phi"] = dict(a=0.1, b=0.3, dim=20, beta=20.0)
    # Rings (distr/rings.py:93-109)
    rg = r_rings.Rings()
    xr = 3.0 * torch.randn(32, 2)
    out.update(rings_x=xr, rings_score=rg.score(xr), rings_logp=rg.unnorm_log_prob(xr))
    # IsotropicGauss / Gauss / GaussFull
    iso = r_gauss.IsotropicGauss(dim=12, loc=0.5, scale=1.7)
    out.update(iso_logp=iso.log_prob(x), iso_score=iso.score(x))
    meta["iso"] = dict(dim=12, loc=0.5, scale=1.7)
    gd = r_gauss.Gauss(dim=12, loc=torch.linspace(-1, 1, 12), scale=torch.linspace(0.5, 2.0, 12))
    out.update(gd_loc=gd.loc, gd_scale=gd.scale, gd_logp=gd.log_prob(x), gd_score=gd.score(x))
    A = torch.randn(12, 12)
    cov = 0.1 * A @ A.T + 0.5 * torch.eye(12)
    gf = r_gauss.GaussFull(dim=12, loc=torch.linspace(-1, 1, 12), cov=cov)
    out.update(gf_loc=gf.loc, gf_cov=cov, gf_logp=gf.log_prob(x), gf_score=gf.score(x))
    # SDE scalars
    ts = torch.linspace(0.0, 1.0, 9)
    ts_in = torch.linspace(0.01, 0.99, 9)
    vp = r_sdes.VP(0.1, 10.0, 1.3, terminal_t=1.0)
    a, b = ts_in[:-1], ts_in[1:]
    out.update(sc_ts=ts_in, vp_alpha=vp.alpha_(ts_in), vp_s=vp.s(ts_in), vp_sigma_sq=vp.sigma_sq(ts_in),
               vp_diff=vp.diff_coeff_t(ts_in), vp_drift=vp.drift_coeff_t(ts_in), vp_omega=vp.omega(a, b),
               vp_lambda=vp.lambda_(a, b), vp_omega_ddpm=vp.omega_ddpm(a, b), vp_int_drift=vp.int_drift_coeff_t(a, b),
               vp_log_snr=vp.log_snr(ts_in))
    meta["vp"] = dict(beta_min=0.1, beta_max=10.0, sigma=1.3, T=1.0)
    pbm = r_sdes.PinnedBM(diff_coeff=math.sqrt(0.2), terminal_t=5.0)
    tp = torch.linspace(0.05, 4.9, 9)
    ap, bp = tp[:-1], tp[1:]
    out.update(pbm_ts=tp, pbm_s=pbm.s(tp), pbm_sigma_sq=pbm.sigma_sq(tp), pbm_omega=pbm.omega(ap, bp),
               pbm_omega_ddpm=pbm.omega_ddpm(ap, bp), pbm_drift=pbm.drift_coeff_t(tp), pbm_log_snr=pbm.log_snr(tp))
    xx = torch.randn(4, 6)
    sc = torch.randn(4, 6)
    z = torch.randn(4, 6)
    out.update(step_x=xx, step_sc=sc, step_z=z,
               vp_ei=vp.ei_integration_step(xx, a[3], b[3], sc, z=z)[0],
               vp_ddpm=vp.ddpm_integration_step(xx, a[3], b[3], sc, z=z)[0],
               pbm_ei=pbm.ei_integration_step(xx, ap[3], bp[3], sc, z=z)[0],
               pbm_ddpm=pbm.ddpm_integration_step(xx, ap[3], bp[3], sc, z=z)[0])
    # time grids (utils/common.py:30-82)
    out.update(ts_uniform=r_get_timesteps(0.0, 1.0, steps=10), ts_cosine=r_get_timesteps(0.0, 6.4, dt=0.05, rescale_t="cosine"),
               ts_quad=r_get_timesteps(0.0, torch.tensor(2.0), steps=10, rescale_t="quad"),
               ts_snr_vp=r_get_timesteps(1e-4, 1.0 - 1e-4, steps=12, sde=r_sdes.VP(0.1, 10.0, 1.0, terminal_t=1.0)),
               ts_snr_pbm=r_get_timesteps(1e-4, 5.0 - 1e-4, steps=12, sde=pbm))
    # net forward (models/mlp.py:135-143, :85-96)
    net = liven(fourier_mlp(12))
    tt = torch.tensor(0.37)
    with torch.no_grad():
        out.update(net_t=tt, net_x=x, net_out=net(tt, x), temb_out=net.timestep_embed(tt.view(1, 1)))
    out.update(**pack_params("net.", sd(net)))
    sm = score_time_embed(bias=0.2)
    with torch.no_grad():
        out.update(sm_out=sm(tt), **pack_params("sm.", sd(sm)))
    # marginal GMM score at a time (eq/sdes.py:329-345)
    vp1 = r_sdes.VP(0.1, 10.0, 1.0, terminal_t=1.0)
    mv = 0.5 * torch.ones(5, 12) * (1 + 0.3 * torch.rand(5, 12))
    out.update(mg_vars=mv, mg_score=vp1.marginal_gmm_score(torch.tensor(0.41), x, tgt.loc, mv, torch.ones(5)),
               mg_logp0=vp1.marginal_gmm_distr(torch.tensor(0.0), tgt.loc, mv, torch.ones(5)).log_prob(x))
    save("unit_vectors", dict(kind="unit", **meta), out)


def case_euler(name, kind, d, B, N, seed, n_out=7):
    """EulerIntegrator.integrate (eq/integrator.py:93-129) on the SDEs the solvers hand it:
    ``langevin_*``: LangevinSDE of a target (eq/sdes.py:46-76; solver/langevin.py:36-66); ``ou_vp``: the uncontrolled
    inference process (solver/oc.py:162-180, :502); ``controlled_vp``: ControlledSDE(VP, ClippedCtrl) (solver/oc.py:205, :378).
    The Brownian increments come through the integrator's own ``bm`` hook: bm(s_k, t_k) = philox_normal(seed, k) sqrt(t_k - s_k).
    The grid is non-uniform and ``ts`` (n_out points) is coarser than it, so the interpolation (:66-77, :120-122) is exercised."""
    torch.manual_seed(seed)
    T = 1.0
    meta = dict(kind="euler", sde_kind=kind, d=d, B=B, N=N, seed=seed, T=T)
    arrays = {}
    if kind.startswith("langevin"):
        g, clip_score = 1.3, 40.0
        if kind == "langevin_gmm":
            target = r_gauss.ManyModes(n_modes=4, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0, n_reference_samples=10)
            arrays.update(tgt_loc=target.loc, tgt_scale=target.scale, tgt_w=target.mixture_weights)
        elif kind == "langevin_phi4":
            target = r_phi.PhiFour(a=0.1, b=0.0, dim=d, dim_phys=1, beta=20.0)
            meta.update(phi_a=0.1, phi_b=0.0, phi_beta=20.0)
            T, g = 0.05, 1.0
        else:
            target = r_rings.Rings(dim=2, n_reference_samples=10)
            meta.update(lower_rad=1.0, upper_rad=5.0, num_rad=3, scale=0.1)
            arrays.update(rings_rad=target.radiuses, rings_w=target.radius_dist.mixture_distribution.probs)
            T, g, clip_score = 0.5, 1.0, 25.0
        sde = r_sdes.LangevinSDE(target_score=target.score, diff_coeff=g, clip_score=clip_score, terminal_t=T)
        meta.update(diff_coeff=g, clip_score=clip_score, T=T)
    else:
        base = r_sdes.VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, scale_diff_coeff=1.0, terminal_t=T)
        meta.update(sde="vp", beta_min=0.1, beta_max=10.0, sigma=1.0)
        if kind == "controlled_vp":
            ctrl = r_rep.ClippedCtrl(base_model=liven(fourier_mlp(d)), clip_model=1e4)
            sde = r_sdes.ControlledSDE(sde=base, ctrl=ctrl)
            meta.update(clip_model=1e4)
            arrays.update(pack_params("ctrl.", sd(ctrl)))
        else:
            sde = base
    inner = torch.sort(torch.rand(N - 1) * T).values
    timesteps = torch.cat([torch.zeros(1), inner, torch.tensor([T])]).float()
    ts = torch.linspace(0.0, T, n_out)
    x0 = (0.3 if kind == "langevin_phi4" else 1.5) * orc.philox_normal(seed, 0, 0, B, d, stream=1)
    grid = [float(v) for v in timesteps]

    def bm(s, t):
        k = grid.index(float(s))
        return orc.philox_normal(seed, k, 0, B, d) * torch.sqrt(t - s)

    with torch.no_grad():
        out = r_int.EulerIntegrator().integrate(sde, ts=ts, x_init=x0.clone(), timesteps=timesteps, bm=bm)
        full = r_int.EulerIntegrator().integrate(sde, ts=timesteps, x_init=x0.clone(), timesteps=timesteps, bm=bm)
    assert out.shape == (n_out, B, d) and bool(torch.isfinite(out).all())
    arrays.update(ts=ts, timesteps=timesteps, x0=x0, out_xs=out, out_last=full[-1])
    save(name, meta, arrays)
    print(f"   |x_T| max {float(full[-1].abs().max()):.3f}")


def case_sampler(name, kind, seed, B=32, d=3, n_levels=5, n_warm=4, n_steps=3, precond=False, pdds=False, **kw):
    """additions/ebm_mle.py: smc_sampler (:11-195) and re_sampler (:269-400) run on CPU with torch's global generator
    seeded; the annealing path is the closed form of tests/golden_cases.py (an INPUT of the fixture), optionally with
    preconditioned moves or the PDDS transition (VP(0.1, 10) EI kernel).  The product's samplers consume random numbers in
    the same order, so the outputs are compared entry by entry."""
    from sde_sampler.additions import ebm_mle as r_ebm
    from tests import golden_cases as gc
    meta = dict(kind="sampler", sampler=kind, seed=seed, B=B, d=d, n_levels=n_levels, n_warm=n_warm, n_steps=n_steps, step=0.05,
                precond=precond, pdds=pdds, kw=dict(kw))
    x_init, times, steps = gc.sampler_inputs(meta)
    extra = dict(gc.sampler_precond(meta))
    if pdds:
        extra.update(use_pdds_weights=True, sde=r_sdes.VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, scale_diff_coeff=1.0, terminal_t=1.0))
    fn = gc.sampler_fn(meta)
    torch.manual_seed(seed)
    with torch.no_grad():
        if kind == "smc":
            samples, steps_out, diags = r_ebm.smc_sampler(x_init, times, fn, n_warm, n_steps, steps.clone(), verbose=False, **kw, **extra)
        else:
            kw = dict(kw)
            samples, steps_out, diags = r_ebm.re_sampler(x_init, times, fn, kw.pop("swap_frequency"), n_warm, n_steps, steps.clone(),
                                                         verbose=False, **kw, **extra)
    arrays = dict(samples=samples, steps_out=steps_out.reshape(n_levels, B, 1))
    for k, v in diags.items():
        arrays["diag_" + k] = torch.as_tensor(v).float()
    save(name, meta, arrays)


CASES = {
    "unit_vectors": unit_vectors,
    # config 2 family (ManyModes d=128, RDS gmm-ref, VP, EI)
    "rds_ei_gmm_d128_k4": lambda n: case_rds_gmm(n, d=128, K=4, B=64, N=16, seed=11),
    "rds_ei_gmm_d128_k4_n256": lambda n: case_rds_gmm(n, d=128, K=4, B=32, N=256, seed=12),
    "rds_ei_gmm_d128_k16": lambda n: case_rds_gmm(n, d=128, K=16, B=32, N=32, seed=13),
    "rds_ei_gmm_d8_k4": lambda n: case_rds_gmm(n, d=8, K=4, B=96, N=100, seed=14),
    "rds_ddpm_gmm_d16_snr": lambda n: case_rds_gmm(n, d=16, K=4, B=64, N=32, seed=15, integrator="ddpm_like", time_type="snr"),
    "rds_em_gmm_d16": lambda n: case_rds_gmm(n, d=16, K=4, B=64, N=64, seed=16, integrator="em"),
    # RemoveReferenceCtrl(CancelDriftCtrl) -- Langevin init on a reference solver (benchmark_utils.py:260-262), EM and EI
    "rds_em_remove_ref_d16": lambda n: case_rds_gmm(n, d=16, K=4, B=64, N=48, seed=111, integrator="em", remove_ref=True),
    "rds_ei_remove_ref_d40": lambda n: case_rds_gmm(n, d=40, K=3, B=48, N=32, seed=112, integrator="ei", remove_ref=True),
    "rds_em_vp_default_d16": lambda n: case_rds_default(n, d=16, K=4, B=64, N=64, seed=17, sde_kind="vp", integrator="em"),
    "rds_ei_vp_default_d16": lambda n: case_rds_default(n, d=16, K=4, B=64, N=32, seed=18, sde_kind="vp", integrator="ei"),
    "rds_ei_pbm_default_d16": lambda n: case_rds_default(n, d=16, K=4, B=64, N=32, seed=19, sde_kind="pbm", integrator="ei"),
    "rds_ei_gauss_fullcov_d40": lambda n: case_rds_default(n, d=40, K=4, B=48, N=32, seed=104, sde_kind="vp", integrator="ei", full=True),
    # full-covariance mixture references (score_mog_full): covariance matrices, and the (D, P) eigen form
    "rds_ei_gmm_fullcov_d128_k4": lambda n: case_rds_gmm(n, d=128, K=4, B=32, N=32, seed=101, cov="full"),
    "rds_em_gmm_fullcov_d40_k3": lambda n: case_rds_gmm(n, d=40, K=3, B=48, N=48, seed=102, integrator="em", cov="full"),
    "rds_ei_gmm_eigen_d16_k3": lambda n: case_rds_gmm(n, d=16, K=3, B=64, N=32, seed=103, cov="eigen"),
    # compute_eubo (noising direction) of the RDS losses
    "eubo_ei_gmm_d128_k4": lambda n: case_eubo_gmm(n, d=128, K=4, B=64, N=16, seed=61, integrator="ei"),
    "eubo_ei_gmm_d16_k4": lambda n: case_eubo_gmm(n, d=16, K=4, B=64, N=64, seed=62, integrator="ei"),
    "eubo_em_gmm_d16_k4": lambda n: case_eubo_gmm(n, d=16, K=4, B=64, N=64, seed=63, integrator="em"),
    "eubo_ei_gmm_fullcov_d40_k3": lambda n: case_eubo_gmm(n, d=40, K=3, B=48, N=32, seed=66, integrator="ei", cov="full"),
    "eubo_dis_ei_d8": lambda n: case_eubo_dis(n, d=8, K=4, B=64, N=32, seed=64),
    "eubo_cmcd_gmm_d16": lambda n: case_eubo_cmcd(n, d=16, K=4, B=64, N=32, seed=65),
    # log-variance training evaluation (loss + gradients) of the RDS losses
    "train_lv_ei_gmm_d16": lambda n: case_train_lv(n, d=16, K=4, B=64, N=32, seed=71, integrator="ei"),
    "train_lv_em_gmm_d16": lambda n: case_train_lv(n, d=16, K=4, B=64, N=32, seed=72, integrator="em"),
    "train_lv_dis_ei_d8": lambda n: case_train_lv_dis(n, d=8, K=4, B=64, N=32, seed=73),
    "train_lv_dds_d2": lambda n: case_train_lv_dds(n, d=2, B=128, seed=74),
    "train_lv_dis_orig_d8": lambda n: case_train_lv_dis_orig(n, d=8, K=4, B=64, N=64, seed=76),
    "train_lv_cmcd_gmm_d16": lambda n: case_train_lv_cmcd(n, d=16, K=4, B=64, N=32, seed=77),
    "train_lv_pis_phi4_d100": lambda n: case_train_lv_pis(n, d=100, B=32, N=16, seed=75, dt=5.0 / 512),
    # KL training (method='kl': back-propagation through the whole trajectory, BaseOCLoss.compute_loss :105-131): loss + every gradient
    "train_kl_dds_d2": lambda n: case_train_lv_dds(n, d=2, B=128, seed=174, method="kl"),
    "train_kl_ei_gmm_d16": lambda n: case_train_lv(n, d=16, K=4, B=64, N=32, seed=171, integrator="ei", method="kl"),
    "train_kl_em_gmm_d16": lambda n: case_train_lv(n, d=16, K=4, B=64, N=32, seed=172, integrator="em", method="kl"),
    "train_kl_dis_ei_d8": lambda n: case_train_lv_dis(n, d=8, K=4, B=64, N=32, seed=173, method="kl"),
    "train_kl_dis_orig_d8": lambda n: case_train_lv_dis_orig(n, d=8, K=4, B=64, N=64, seed=176, method="kl"),
    "train_kl_pis_phi4_d100": lambda n: case_train_lv_pis(n, d=100, B=32, N=16, seed=175, dt=5.0 / 512, method="kl"),
    "train_kl_cmcd_gmm_d16": lambda n: case_train_lv_cmcd(n, d=16, K=4, B=64, N=32, seed=177, method="kl"),
    # config 3 (PhiFour d=100, PIS, EM), at the real step size 5/512
    "pis_em_phi4_d100": lambda n: case_pis_phi4(n, d=100, B=64, N=32, seed=21, dt=5.0 / 512),
    # config 1 (TwoModes d=2, DDS) and the Rings target on the same solver
    "dds_two_modes_d2": lambda n: case_dds(n, d=2, B=128, seed=31),
    "dds_rings_d2": lambda n: case_dds(n, d=2, B=128, seed=32, rings=True),
    # target-informed controls on FULL-covariance mixture targets (score_mog_full in the step loop): the reference's TwoModesFull, a GMMFull
    "dds_two_modes_full_d5": lambda n: case_dds(n, d=5, B=128, seed=33, full=True),
    "pis_two_modes_full_d20": lambda n: case_pis_full(n, d=20, K=0, B=64, N=40, seed=34, dt=0.05),
    "pis_gmm_full_d128_k3": lambda n: case_pis_full(n, d=128, K=3, B=32, N=32, seed=35, dt=0.05),
    # config 4 (logreg d=61, CMCD), at the real step size 1/256
    "cmcd_logreg_d61": lambda n: case_cmcd_logreg(n, B=64, N=16, seed=41, dt=1.0 / 256),
    "cmcd_gmm_iso_d16": lambda n: case_cmcd_gmm(n, d=16, K=4, B=64, N=32, seed=42, prior_kind="iso"),
    "cmcd_gmm_diag_d40": lambda n: case_cmcd_gmm(n, d=40, K=4, B=64, N=32, seed=43, prior_kind="diag"),
    "cmcd_phi4_d100": lambda n: case_cmcd_phi4(n, d=100, B=16, N=256, seed=44),
    "pis_logreg_d61": lambda n: case_logreg_ctrl(n, B=32, N=32, seed=45, solver="pis"),
    "dds_logreg_d61": lambda n: case_logreg_ctrl(n, B=32, N=32, seed=46, solver="dds"),
    # EulerIntegrator (eq/integrator.py) on Langevin / uncontrolled / controlled SDEs
    "euler_langevin_gmm_d16": lambda n: case_euler(n, "langevin_gmm", d=16, B=48, N=40, seed=81),
    "euler_langevin_phi4_d100": lambda n: case_euler(n, "langevin_phi4", d=100, B=24, N=32, seed=82),
    "euler_langevin_rings_d2": lambda n: case_euler(n, "langevin_rings", d=2, B=96, N=48, seed=83),
    "euler_ou_vp_d40": lambda n: case_euler(n, "ou_vp", d=40, B=32, N=24, seed=84),
    "euler_controlled_vp_d16": lambda n: case_euler(n, "controlled_vp", d=16, B=48, N=32, seed=85),
    # SMC / replica-exchange samplers (additions/ebm_mle.py)
    "smc_tempered_d3": lambda n: case_sampler(n, "smc", seed=91, reweight_threshold=0.7),
    "smc_annealed_langevin_d3": lambda n: case_sampler(n, "smc", seed=92, reweight_threshold=0.0, use_ula=True),
    "re_tempered_d3": lambda n: case_sampler(n, "re", seed=93, n_steps=9, swap_frequency=3),
    "re_ula_d3": lambda n: case_sampler(n, "re", seed=94, n_steps=7, swap_frequency=3, use_ula=True),
    "smc_precond_d3": lambda n: case_sampler(n, "smc", seed=95, precond=True, reweight_threshold=0.8),
    "smc_pdds_d3": lambda n: case_sampler(n, "smc", seed=96, pdds=True, reweight_threshold=0.9),
    "re_precond_d3": lambda n: case_sampler(n, "re", seed=97, n_steps=8, precond=True, swap_frequency=3),
    # DIS variants
    "dis_ei_d8": lambda n: case_dis(n, d=8, K=4, B=64, N=32, seed=51, kind="ei"),
    "dis_orig_lerp_d8": lambda n: case_dis(n, d=8, K=4, B=64, N=64, seed=52, kind="orig"),
    "dis_ei_cancel_drift_d8": lambda n: case_dis(n, d=8, K=4, B=64, N=32, seed=53, kind="ei", cancel_drift=True),
}


def main():
    """python tests/golden/gen_golden.py [case ...]   (default: every case)"""
    for name in (sys.argv[1:] or list(CASES)):
        if name == "unit_vectors":
            unit_vectors()
        else:
            CASES[name](name)


if __name__ == "__main__":
    main()
