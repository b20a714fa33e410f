"""``make_model`` with the reference's signature (``experiments/benchmark_utils.py:96-265``), without Hydra:
the YAML constants of ``conf/`` it composes are restated here as dicts, then the same overrides are applied
and the solver is built (``sde_sampler_lrds_amd.solver.oc``).  Returns an object whose ``evaluate()`` runs the
HIP engine."""
from __future__ import annotations

import math
from functools import partial

import torch

from ..distr.gauss import BracketTwoModes, ManyModes, TwoModes, TwoModesFull
from ..distr.logistic_regression import LogisticRegression
from ..distr.phi_four import PhiFour
from ..distr.rings import Rings
from ..engine import UnsupportedByEngine
from ..models.reparam import RemoveReferenceCtrl
from ..solver import oc
from ..utils.common import get_timesteps

solver_types = {"dds_orig": "dds", "pis_orig": "pis", "dis_orig": "dis", "cmcd": "cmcd", "vp-ref": "vp_rds", "pbm-ref": "pbm_rds"}
model_types = {"target_informed_zero_init": "score", "target_informed_unet_zero_init": "score_unet",
               "target_informed_langevin_init": "langevin_init", "target_informed_lerp_tempering": "lerp", "base_zero_init": "basic",
               "unet_zero_init": "basic_unet"}

# conf/solver/*.yaml -> (solver class, prior, sde, default model, default loss)
_SOLVERS = {
    "pis": (oc.PIS, dict(kind="delta"), dict(kind="ScaledBM", diff_coeff=math.sqrt(0.2), terminal_t=5.0), "EMReferenceSDELoss"),
    "dds": (oc.DDS, dict(kind="gauss"), None, "ExponentialIntegratorSDELoss"),
    "dis": (oc.Bridge, dict(kind="gauss"), dict(kind="VP", diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, scale_diff_coeff=1.0, terminal_t=1.0), "TimeReversalLoss"),
    "cmcd": (oc.CMCD, dict(kind="gauss", scale=5.0), dict(kind="ControlledLangevinSDE", diff_coeff=1.0, terminal_t=1.0, clip_score=1e5), "ControlledLangevinSDELoss"),
    "vp_rds": (oc.RDS, dict(kind="gauss"), dict(kind="VP", diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, scale_diff_coeff=1.0, terminal_t=1.0), "EMReferenceSDELoss"),
    "pbm_rds": (oc.RDS, dict(kind="delta"), dict(kind="PinnedBM", diff_coeff=math.sqrt(0.2), terminal_t=5.0), "EMReferenceSDELoss"),
}


# conf/target/{sonar,ionosphere,cancer,credit}.yaml
LOGREG_TARGETS = {
    "sonar": dict(dim=61, data_type="sonar", intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5),
    "ionosphere": dict(dim=34, data_type="ionosphere", intercept_mean=4.25, intercept_scale=0.25, weight_scale=5.25),
    "cancer": dict(dim=31, data_type="cancer", intercept_mean=31.0, intercept_scale=2.0, weight_scale=3.75),
    "credit": dict(dim=25, data_type="credit", intercept_mean=3.25, intercept_scale=0.5, weight_scale=1.25),
}


def make_target_details(target_name, **kwargs):
    """experiments/benchmark_utils.py:41-93 (targets with a HIP kernel)."""
    if target_name == "two_modes":
        return dict(name=target_name, dim=kwargs.get("dim", 5), ill_conditioned=kwargs.get("ill_conditioned", "medium"), a=kwargs.get("a", 1.0))
    if target_name == "two_modes_full":  # log-density only (terminal cost): solvers with a ClippedCtrl
        return dict(name=target_name, dim=kwargs.get("dim", 5), ill_conditioned=kwargs.get("ill_conditioned", "medium"), a=kwargs.get("a", 1.0))
    if target_name == "bracket_two_modes":
        return dict(name=target_name, dim=kwargs.get("dim", 5), a=kwargs.get("a", 0.75))
    if target_name == "many_modes":
        return dict(name="many_modes", dim=kwargs.get("dim", 5), n_modes=kwargs.get("n_modes", 4),
                    mixture_weight_factor=kwargs.get("mixture_weight_factor", 3.0), var=kwargs.get("var", 0.5))
    if target_name == "phi_four":
        return dict(name="phi_four", dim=kwargs.get("dim", 100), b=kwargs.get("b", 0.0))
    if target_name == "rings":
        return dict(name="rings")
    if target_name in LOGREG_TARGETS:  # benchmark_utils.py:84-91
        return dict(name=target_name)
    raise UnsupportedByEngine(f"Target {target_name} has no HIP log-density / score kernel.")


def _make_target(details):
    d = dict(details)
    name = d.pop("name")
    if name == "many_modes":  # conf/target/many_modes.yaml
        return ManyModes(**{"n_modes": 4, "dim": 8, "seed_loc": 42, "mixture_weight_factor": 3.0, "var": 0.5, "n_reference_samples": 10000, **d})
    if name == "two_modes":  # conf/target/two_modes.yaml
        return TwoModes(**{"dim": 2, "a": 1.0, "n_reference_samples": 10000, **d})
    if name == "two_modes_full":  # conf/target/two_modes_full.yaml
        return TwoModesFull(**{"dim": 5, "a": 1.0, "n_reference_samples": 16384, **d})
    if name == "bracket_two_modes":  # conf/target/bracket_two_modes.yaml
        return BracketTwoModes(**{"dim": 5, "a": 1.0, "n_reference_samples": 16384, **d})
    if name == "phi_four":  # conf/target/phi_four.yaml
        return PhiFour(**{"dim": 100, "a": 0.1, "b": 0.0, "dim_phys": 1, "beta": 20.0, **d})
    if name == "rings":  # conf/target/rings.yaml
        return Rings(**{"dim": 2, "n_reference_samples": 10000, **d})
    if name in LOGREG_TARGETS:  # the design matrix: register_dataset / $SDENG_DATA_DIR/<name>.pt (distr/logistic_regression.py)
        return LogisticRegression(**{**LOGREG_TARGETS[name], **d})
    if "object" in details:
        return details["object"]
    raise NotImplementedError(name)


def validate_make_model_args(solver_type, ref_type, loss_type, integrator_type, model_type, time_type, solver_details=None,
                             target_details=None, training_details=None, force_base_zero_init=False, force_vp20=False,
                             force_vp_cosine=False):
    """The argument checks of the reference's ``make_model`` (experiments/benchmark_utils.py:100-160), in its order, with its
    exception types and messages: the same combinations are rejected, for the same stated reason.  ``tests/golden/make_model_grid.json``
    holds the outcome of the reference's own lines for the whole (solver x ref x loss x integrator x model x time x force_*) grid;
    ``tests/test_make_model_contract.py`` compares this function with it entry by entry."""
    # :101-109
    assert solver_type in solver_types
    assert ref_type in ["default", "gaussian", "gmm", "nn"]
    assert loss_type in ["kl", "lv"]
    assert integrator_type in ["em", "ei", "ddpm_like"]
    assert model_type in model_types
    assert time_type in ["uniform", "snr"]
    if solver_details is not None:
        assert isinstance(solver_details, dict)
    if target_details is not None:
        assert isinstance(target_details, dict) and ("name" in target_details)
    if training_details is not None:
        assert isinstance(training_details, dict)
    zero_init_models = ["target_informed_zero_init", "target_informed_unet_zero_init"]
    # :112-131 the original samplers (PIS / DDS / DIS) and CMCD
    if ("orig" in solver_type) or ("dis" in solver_type) or ("cmcd" in solver_type):
        if not ((model_type == "base_zero_init") and force_base_zero_init):
            if (solver_type == "dds_orig") and (model_type not in zero_init_models):
                raise ValueError("Only target_informed_zero_init model is supported.")
            if (solver_type == "pis_orig") and (model_type not in zero_init_models):
                raise ValueError("Only target_informed_zero_init model is supported.")
            if ("dis" in solver_type) and (model_type == "base_zero_init"):
                raise ValueError("Model base_zero_init is not supported.")
            if (solver_type == "cmcd") and (model_type == "base_zero_init"):  # (the upstream message says the opposite of the test)
                raise ValueError("Only base_zero_init is supported for CMCD.")
        if not (time_type == "uniform"):
            raise ValueError("Only uniform time discretisation is supported for orig/cmcd models.")
        if not (integrator_type == "em"):
            raise ValueError("Can't use EI or DDPM-like discretization with orig models.")
        if force_vp20 and (solver_type != "dis_orig"):
            raise ValueError("Can't use vp_20 for orig models other than DIS.")
        if force_vp_cosine:
            raise ValueError("Can't use vp_cosine for orig models.")
    # :134-144 the reference-based samplers
    if "ref" in solver_type:
        if model_type == "target_informed_lerp_tempering":
            raise ValueError("Model target_informed_lerp_tempering is not supported.")
        if (solver_type == "pbm-ref") and (time_type == "uniform"):
            raise ValueError("PBM schedule is unstable with uniform time discretization.")
        if (integrator_type == "ddpm_like") and (time_type == "uniform"):
            raise ValueError("Using the integration scheme from DDPM with uniform times is unstable.")
    # :147-150
    if force_vp20 and force_vp_cosine:
        raise ValueError("Can't use vp_20 and vp_cosine at the same time.")
    if (solver_type == "pbm-ref") and (force_vp20 or force_vp_cosine):
        raise ValueError("Can't use vp_20 or vp_cosine with PBM.")
    # :153-156
    if ((ref_type != "default") and ("ref" not in solver_type)) and (solver_type != "cmcd"):
        raise ValueError("Only ref models can use a non-default ref.")
    if (solver_type == "cmcd") and (ref_type not in ["default", "gaussian"]):
        raise ValueError("Can't use ref other than gaussian for CMCD.")
    # :159-160
    if (model_type == "target_informed_langevin_init") and (integrator_type in ["ei", "ddpm_like"]):
        raise ValueError("Can't use EI or DDPM-like with Langevin score.")


def make_model(solver_type, ref_type, loss_type, integrator_type, model_type, time_type, solver_details, target_details,
               training_details, optim_details=None, n_steps=100, force_base_zero_init=False, use_ema=False, force_vp20=False,
               force_vp_cosine=False, compute_samples_based_metrics=True, force_T_cosine=None, device="cuda"):
    """experiments/benchmark_utils.py:96-265.  Every combination the reference rejects is rejected here with the same exception
    (``validate_make_model_args``); of those it accepts, the ones without a HIP kernel -- UNet drift nets, 'nn' references -- raise
    ``UnsupportedByEngine`` (a NotImplementedError) AFTER that validation, never a silently different sampler."""
    validate_make_model_args(solver_type, ref_type, loss_type, integrator_type, model_type, time_type, solver_details, target_details,
                             training_details, force_base_zero_init, force_vp20, force_vp_cosine)
    if model_types[model_type] in ("score_unet", "basic_unet"):
        raise UnsupportedByEngine(f"model_type {model_type}: the UNet drift net (models/mnist_unet.py) has no HIP kernel")
    if ref_type == "nn" and "ref" in solver_type:
        raise UnsupportedByEngine("'nn' references need autograd of an energy net inside every step (models/reparam.py:478-482): no HIP kernel")

    cls, prior, sde, loss_kind = _SOLVERS[solver_types[solver_type]]
    prior, sde = dict(prior), (dict(sde) if sde else None)
    if force_vp20 and sde and sde["kind"] == "VP":
        sde["diff_coeff_sq_max"] = 20.0
    if force_vp_cosine and sde and sde["kind"] == "VP":
        sde = dict(kind="CosineVP", c=0.008, scale_diff_coeff=1.0, terminal_t=1.0)
    loss = dict(kind=loss_kind, method=loss_type, traj_per_sample=1, max_rnd=1e8 if loss_type == "lv" else None)
    if "ref" in solver_type and integrator_type == "ei":
        loss["kind"] = "EIReferenceSDELoss"
    if "ref" in solver_type and integrator_type == "ddpm_like":
        loss["kind"] = "DDPMLikeReferenceSDELoss"
    ts = dict(start=0.0, end=sde["terminal_t"] if sde else 6.4, steps=n_steps)
    if solver_type == "dds_orig":
        loss.update(alpha=1.0, sigma=solver_details["sigma"])
        ts = dict(start=0.0, end=force_T_cosine or 6.4, dt=0.05, rescale_t="cosine")
        prior["scale"] = solver_details["sigma"]
    elif solver_type == "pis_orig":
        sde["diff_coeff"] = solver_details["sigma"]
    elif solver_type == "dis_orig":
        sde["scale_diff_coeff"] = solver_details["sigma"]
        prior["scale"] = solver_details["sigma"]
    elif "ref" in solver_type and ref_type == "default":
        if "pbm" in solver_type:
            sde["diff_coeff"] = solver_details["sigma"]
        else:
            sde["scale_diff_coeff"] = solver_details["sigma"]
            prior["scale"] = solver_details["sigma"]
    if time_type == "snr":
        ts["start"], ts["end"] = 1e-4, sde["terminal_t"] - 1e-4
    if force_vp_cosine:
        ts["start"] = 1e-3
    cfg = dict(prior=prior, sde=sde, model=model_types[model_type], loss=loss, timesteps=ts,
               eval_batch_size=training_details["eval_batch_size"], train_batch_size=training_details["train_batch_size"],
               train_steps=training_details.get("train_steps", 0), optim=dict(optim_details or {}), use_ema=use_ema)  # benchmark_utils.py:181-183, 215-217
    model = cls(cfg, _make_target(target_details), device=device)
    if "ref" in solver_type:
        if ref_type == "gaussian":
            model.change_reference_type(ref_type="gaussian", mean=solver_details["mean_ref"], var=solver_details["var_ref"])
        elif ref_type == "gmm":
            model.change_reference_type(ref_type="gmm", weights=solver_details["weights_ref"], means=solver_details["means_ref"],
                                        variances=solver_details["variances_ref"])
    if "cmcd" in solver_type and ref_type == "gaussian":
        model.update_prior(mean=solver_details["mean"], var=solver_details["var"])
    if time_type == "snr":
        model.train_timesteps = partial(get_timesteps, **model.train_timesteps.keywords, sde=model.sde)
        model.eval_timesteps = model.train_timesteps
    # benchmark_utils.py:260-262.  Upstream this rebinds ``model.generative_ctrl`` AFTER the loss was built (solver/oc.py:504-511 hands the
    # loss the control object at construction), so the loss -- every simulate / eval / training call -- keeps driving the un-wrapped
    # CancelDriftCtrl; the wrapper only changes what ``model.generative_ctrl`` / ``state_dict()['generative_ctrl']`` (keys gain a
    # ``score.`` prefix) and the EMA update see.  Reproduced as is: same object graph, same sampler.  (Called directly, the wrapper built
    # with upstream's defaults -- use_rescaling=True, sde=None -- raises AttributeError there and here, models/reparam.py:58-60.)
    if model_type == "target_informed_langevin_init" and "ref" in solver_type:
        model.generative_ctrl = RemoveReferenceCtrl(model.generative_ctrl, model.reference_score_t).to(model.device)
    return model


def fit_gmm(n_components, dataset, means_init=None, em_type="diag", max_iter=1000):
    """experiments/benchmark_utils.py:336-361: fit the mixture that becomes the learned reference of (L)RDS
    (``model.change_reference_type('gmm', weights=..., means=..., variances=...)``).  Host-side data preparation with
    scikit-learn, like upstream; ``em_type='diag'`` gives [K,d] variances, ``em_type='full'`` covariance matrices [K,d,d]
    (both have step-loop kernels)."""
    from sklearn.mixture import GaussianMixture
    if em_type not in ("diag", "full"):
        raise NotImplementedError(f"covariance_type '{em_type}': the engine's reference kernels take 'diag' or 'full' mixtures")
    data = dataset.reshape(-1, dataset.shape[-1]).cpu().numpy()
    last = None
    for reg_covar in [1e-6, 5e-5, 1e-5, 5e-4, 1e-4, 5e-3, 1e-3, 5e-2, 1e-2]:
        try:
            gmm = GaussianMixture(n_components=n_components, covariance_type=em_type, reg_covar=reg_covar, max_iter=max_iter,
                                  means_init=means_init.cpu().numpy() if means_init is not None else None).fit(data)
            weights = torch.from_numpy(gmm.weights_).float()
            means = torch.from_numpy(gmm.means_).float()
            variances = torch.from_numpy(gmm.covariances_).float()
            if em_type == "full":  # upstream validates by constructing GMMFull: positive definite covariances
                torch.linalg.cholesky(variances)
                return weights, means, variances
            if bool(torch.isfinite(variances).all()) and bool((variances > 0).all()):
                return weights, means, variances
        except Exception as e:  # noqa: BLE001  (upstream retries with the next regulariser on any failure)
            last = e
    raise ValueError(f"Couldn't fit a GMM on this dataset. ({last})")


def mcmc_sample(device, target, x_init, mcmc_type="mala", step_size=1e-3, n_chains_per_mode=4, dataset_length=50000,
                n_warmup_steps=512, skip_chain_per_mode=False, target_log_prob_and_grad=None, adapt_step_size=True, shuffle=True,
                verbose=False):
    """experiments/benchmark_utils.py:268-333: chains started at ``x_init`` (one group per mode) producing the data set the
    reference mixture is fitted to.  The target's log-density and score come from the HIP distribution kernels."""
    from ..additions import mcmc
    target = target.to(device)
    if target_log_prob_and_grad is None:
        target_log_prob_and_grad = mcmc.hip_log_prob_and_grad(target)
    x_init = x_init.to(device)
    y = x_init.clone() if skip_chain_per_mode else x_init.repeat_interleave(n_chains_per_mode, dim=0)
    n_chains = y.shape[0]
    n_mcmc_steps = int(dataset_length / n_chains)
    step = step_size * torch.ones((n_chains, 1), device=device)
    lp, grad = target_log_prob_and_grad(y)
    hip_target = getattr(target_log_prob_and_grad, "hip_target", None)
    from ..additions import ebm_mle
    if mcmc_type == "mala" and hip_target is not None and ebm_mle.NATIVE_MOVES and y.is_cuda and y.dim() == 2:
        # all warm-up and sampling steps of all chains in ONE launch (k_langevin_moves); random numbers as configured in ebm_mle
        y, lp, grad = y.contiguous().float().clone(), lp.contiguous().float().clone(), grad.contiguous().float().clone()
        got, _, _ = ebm_mle._native_run((hip_target, None), None, y, lp, grad, step.reshape(-1).contiguous().clone(), n_warmup_steps + n_mcmc_steps,
                                        n_warmup_steps, False, 0.75 if adapt_step_size else 0.0)
        ret = got.cpu().view((-1, *x_init.shape[1:]))
        return ret[torch.randperm(ret.shape[0])] if shuffle else ret
    ys = torch.empty((n_mcmc_steps, *y.shape))
    for step_id in range(n_warmup_steps + n_mcmc_steps):
        if mcmc_type == "mala":
            y, lp, grad, log_acc = mcmc.mala_step(y, lp, grad, target_log_prob_and_grad, step)
        else:
            y, lp, log_acc = mcmc.rwmh_step(y, lp, lambda v: target_log_prob_and_grad(v)[0], step)
        if adapt_step_size:
            step = mcmc.heuristics_step_size(step, log_acc)
        if step_id >= n_warmup_steps:
            ys[step_id - n_warmup_steps] = y.detach().cpu()
    ret = ys.view((-1, *x_init.shape[1:]))
    return ret[torch.randperm(ret.shape[0])] if shuffle else ret

