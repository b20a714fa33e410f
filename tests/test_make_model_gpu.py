"""The drop-in driver surface: make_model(...) -> solver.evaluate() on the HIP engine (GPU box)."""
import math

import pytest
import torch

from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details


def _train(n):
    return dict(train_steps=1, train_batch_size=64, eval_batch_size=n)


@pytest.mark.gpu
def test_make_model_rds_gmm_evaluate(gpu):
    tgt = make_target_details("many_modes", dim=16, n_modes=4)
    K, d = 4, 16
    g = torch.Generator().manual_seed(0)
    model = make_model("vp-ref", "gmm", "kl", "ei", "base_zero_init", "uniform",
                       dict(means_ref=4 * torch.rand(K, d, generator=g) - 2, variances_ref=0.5 * torch.ones(K, d),
                            weights_ref=torch.ones(K)), tgt, _train(2048), n_steps=32)
    res = model.evaluate()
    assert res.samples.shape == (2048, 16) and res.weights.shape == (2048, 1)
    assert res.xs.shape == (33, 2048, 16)
    assert abs(res.weights.sum().item() - 1.0) < 1e-4
    for key in ("eval/elbo", "eval/lv_loss", "eval/sample_time", "eval/norm_effective_sample_size"):
        assert key in res.metrics
    assert torch.isfinite(res.samples).all()
    assert "log_norm_const_is" in res.log_norm_const_preds
    sd = model.state_dict()
    assert sd["ref_type"] == "gmm" and "ref_means_init" in sd


@pytest.mark.gpu
@pytest.mark.parametrize("solver,ref,integ,mtype,time_type,tname,details", [
    ("pis_orig", "default", "em", "target_informed_zero_init", "uniform", "phi_four", dict(sigma=0.4472135954999579)),
    ("dds_orig", "default", "em", "target_informed_zero_init", "uniform", "two_modes", dict(sigma=1.0)),
    ("dis_orig", "default", "em", "target_informed_lerp_tempering", "uniform", "many_modes", dict(sigma=1.0)),
    ("vp-ref", "default", "ddpm_like", "base_zero_init", "snr", "many_modes", dict(sigma=1.0)),
    ("pbm-ref", "default", "ei", "base_zero_init", "snr", "many_modes", dict(sigma=0.4472135954999579)),
    ("cmcd", "default", "em", "target_informed_zero_init", "uniform", "many_modes", dict()),
    ("cmcd", "gaussian", "em", "target_informed_zero_init", "uniform", "many_modes", dict(mean=torch.zeros(8), var=3.0 * torch.ones(8))),
])
def test_make_model_variants_run(gpu, solver, ref, integ, mtype, time_type, tname, details):
    tgt = make_target_details(tname, dim=100 if tname == "phi_four" else 8)
    model = make_model(solver, ref, "lv", integ, mtype, time_type, details, tgt, _train(512), n_steps=16)
    res = model.evaluate()
    assert torch.isfinite(res.samples).all() and torch.isfinite(res.weights).all()
    assert res.samples.shape[0] == 512


@pytest.mark.gpu
def test_trainable_wrapper_evaluate_with_eubo_metrics(gpu):
    """The notebook flow (additions/hacking.py:36-102): make_model -> TrainableWrapper(model).evaluate(): sampling metrics
    plus the EUBO-side metrics from noising trajectories started at target samples; both passes on the HIP engine."""
    from sde_sampler_lrds_amd.additions.hacking import TrainableWrapper
    tgt = make_target_details("many_modes", dim=16, n_modes=4)
    K, d = 4, 16
    g = torch.Generator().manual_seed(0)
    model = make_model("vp-ref", "gmm", "kl", "ei", "base_zero_init", "uniform",
                       dict(means_ref=4 * torch.rand(K, d, generator=g) - 2, variances_ref=0.5 * torch.ones(K, d),
                            weights_ref=torch.ones(K)), tgt, _train(2048), n_steps=32)
    res = TrainableWrapper(model, verbose=False).evaluate()
    for key in ("eval/elbo", "eval/eubo", "eval/log_norm_const_is_f", "eval/norm_effective_sample_size_f"):
        assert key in res.metrics and math.isfinite(res.metrics[key]), key
    assert 0.0 < res.metrics["eval/norm_effective_sample_size_f"] <= 1.0 + 1e-6
    # ELBO <= log Z <= EUBO  (log Z = 0 for the normalised mixture target); sampling noise leaves slack
    assert res.metrics["eval/elbo"] <= res.metrics["eval/eubo"] + 1e-3


@pytest.mark.gpu
def test_learned_reference_pipeline(gpu):
    """The LRDS recipe end to end on the engine: MALA chains on the target (HIP log-density / score) -> fit_gmm -> RDS with
    the fitted mixture as reference -> a few log-variance training steps -> evaluation with EUBO metrics."""
    from sde_sampler_lrds_amd.additions.hacking import TrainableWrapper
    from sde_sampler_lrds_amd.experiments.benchmark_utils import _make_target, fit_gmm, mcmc_sample
    details = make_target_details("many_modes", dim=8, n_modes=4)
    target = _make_target(details)
    data = mcmc_sample(gpu, target, target.loc.clone(), step_size=5e-2, n_chains_per_mode=32, dataset_length=8192, n_warmup_steps=64)
    assert data.shape == (8192, 8) and torch.isfinite(data).all()
    weights, means, variances = fit_gmm(4, data, means_init=target.loc.cpu())
    assert float((means - target.loc.cpu()).abs().max()) < 0.5  # the chains stayed on their modes
    model = make_model("vp-ref", "gmm", "lv", "ei", "base_zero_init", "uniform",
                       dict(means_ref=means, variances_ref=variances, weights_ref=weights), details,
                       dict(train_steps=30, train_batch_size=256, eval_batch_size=2048), optim_details=dict(lr=1e-3), n_steps=32)
    res = TrainableWrapper(model, verbose=False).run()
    assert math.isfinite(res.metrics["eval/elbo"]) and math.isfinite(res.metrics["eval/eubo"])
    assert res.metrics["eval/norm_effective_sample_size"] > 0.2  # a good reference makes the sampler nearly exact already
