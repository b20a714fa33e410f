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
    # Langevin init on a reference solver (benchmark_utils.py:260-262): the loss keeps the CancelDriftCtrl, the solver's attribute is wrapped
    ("vp-ref", "gmm", "em", "target_informed_langevin_init", "uniform", "bracket_two_modes",
     dict(means_ref=torch.tensor([[1.0] * 8, [-1.0] * 8]), variances_ref=0.5 * torch.ones(2, 8), weights_ref=torch.ones(2))),
    ("vp-ref", "default", "em", "target_informed_langevin_init", "snr", "many_modes", dict(sigma=1.0)),
    ("dis_orig", "default", "em", "target_informed_langevin_init", "uniform", "many_modes", dict(sigma=1.0)),
    ("vp-ref", "default", "ei", "base_zero_init", "uniform", "two_modes_full", dict(sigma=1.0)),  # full-covariance target: terminal cost via torch
    # ... and inside a target-informed control (score_mog_full in the step loop, the mixture held in the reference slot)
    ("pis_orig", "default", "em", "target_informed_zero_init", "uniform", "two_modes_full", dict(sigma=0.4472135954999579)),
    ("dds_orig", "default", "em", "target_informed_zero_init", "uniform", "two_modes_full", dict(sigma=1.0)),
    ("dis_orig", "default", "em", "target_informed_lerp_tempering", "uniform", "two_modes_full", dict(sigma=1.0)),
    ("cmcd", "default", "em", "target_informed_zero_init", "uniform", "many_modes", dict()),
    ("cmcd", "gaussian", "em", "target_informed_zero_init", "uniform", "many_modes", dict(mean=torch.zeros(8), var=3.0 * torch.ones(8))),
])
def test_make_model_variants_run(gpu, solver, ref, integ, mtype, time_type, tname, details):
    tgt = make_target_details(tname, dim=100 if tname == "phi_four" else 8)
    model = make_model(solver, ref, "lv", integ, mtype, time_type, details, tgt, _train(512), n_steps=16)
    res = model.evaluate()
    assert torch.isfinite(res.samples).all() and torch.isfinite(res.weights).all()
    assert res.samples.shape[0] == 512
    if mtype == "target_informed_langevin_init" and "ref" in solver:
        assert type(model.generative_ctrl).__name__ == "RemoveReferenceCtrl" and model.loss.generative_ctrl is model.generative_ctrl.score


@pytest.mark.gpu
def test_make_model_rejects_what_the_reference_rejects(gpu):
    """(the full accept / reject table is tests/test_make_model_contract.py, on the CPU)"""
    tgt = make_target_details("bracket_two_modes", dim=8)
    with pytest.raises(ValueError, match="Only target_informed_zero_init model is supported."):
        make_model("pis_orig", "default", "lv", "em", "target_informed_langevin_init", "uniform", dict(sigma=0.4472135954999579), tgt, _train(512))


@pytest.mark.gpu
def test_full_covariance_target_limits_are_loud(gpu):
    """A full-covariance mixture target has a score kernel only inside a control WITHOUT a reference drift (PIS / DDS / DIS); with a
    reference (the slot is taken) or as a stand-alone HIP control evaluation it raises -- never a silent fallback."""
    from sde_sampler_lrds_amd import engine as E
    tgt = make_target_details("two_modes_full", dim=8)
    model = make_model("vp-ref", "default", "lv", "em", "target_informed_langevin_init", "uniform", dict(sigma=1.0), tgt, _train(256), n_steps=8)
    with pytest.raises(E.UnsupportedByEngine, match="reference drift"):
        model.evaluate()
    pis = make_model("pis_orig", "default", "lv", "em", "target_informed_zero_init", "uniform", dict(sigma=0.4472135954999579), tgt, _train(256), n_steps=8)
    from sde_sampler_lrds_amd import _lib as L
    with pytest.raises((E.UnsupportedByEngine, L.EngineError), match="TwoModesFull|target kind 8"):  # no stand-alone HIP evaluation of this control (sdeng_ctrl_forward has no staged-precision path)
        E.ctrl_forward(pis.generative_ctrl, 0.5, torch.zeros(4, 8, device=gpu))
    u = pis.generative_ctrl(torch.tensor(0.5, device=gpu), torch.zeros(4, 8, device=gpu))  # (the module's own torch forward, as upstream)
    assert u.shape == (4, 8) and torch.isfinite(u).all()


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
@pytest.mark.parametrize("em_type", ["diag", "full"])
def test_learned_reference_pipeline(gpu, em_type):
    """The LRDS recipe end to end on the engine: MALA chains on the target (HIP log-density / score) -> fit_gmm -> RDS with
    the fitted mixture as reference -> a few log-variance training steps -> evaluation with EUBO metrics."""
    from sde_sampler_lrds_amd.additions.hacking import TrainableWrapper
    from sde_sampler_lrds_amd.experiments.benchmark_utils import _make_target, fit_gmm, mcmc_sample
    details = make_target_details("many_modes", dim=8, n_modes=4)
    target = _make_target(details)
    data = mcmc_sample(gpu, target, target.loc.clone(), step_size=5e-2, n_chains_per_mode=32, dataset_length=8192, n_warmup_steps=64)
    assert data.shape == (8192, 8) and torch.isfinite(data).all()
    weights, means, variances = fit_gmm(4, data, means_init=target.loc.cpu(), em_type=em_type)
    assert variances.dim() == (3 if em_type == "full" else 2)
    assert float((means - target.loc.cpu()).abs().max()) < 0.5  # the chains stayed on their modes
    model = make_model("vp-ref", "gmm", "lv", "ei", "base_zero_init", "uniform",
                       dict(means_ref=means, variances_ref=variances, weights_ref=weights), details,
                       dict(train_steps=30, train_batch_size=256, eval_batch_size=2048), optim_details=dict(lr=1e-3), n_steps=32)
    res = TrainableWrapper(model, verbose=False).run()
    assert math.isfinite(res.metrics["eval/elbo"]) and math.isfinite(res.metrics["eval/eubo"])
    assert res.metrics["eval/norm_effective_sample_size"] > 0.2  # a good reference makes the sampler nearly exact already


@pytest.mark.gpu
def test_smc_and_replica_exchange_on_hip_densities(gpu):
    """additions/ebm_mle.py samplers driven by the HIP distribution kernels (hip_tempered_log_prob_and_grads): the tempered
    density/gradient equal the torch formulas, SMC moves a wide Gaussian onto a 3-mode mixture with the right mode weights, and
    replica exchange keeps the target level on it."""
    from sde_sampler_lrds_amd.additions import ebm_mle
    from sde_sampler_lrds_amd.distr.gauss import GMM, IsotropicGauss
    torch.manual_seed(0)
    d, B, n_levels = 4, 4096, 12
    loc = torch.tensor([[3.0, 0, 0, 0], [-3.0, 0, 0, 0], [0, 3.0, 0, 0]])
    wts = torch.tensor([0.5, 0.3, 0.2])
    target = GMM(dim=d, loc=loc, scale=0.6 * torch.ones(3, d), mixture_weights=wts.clone()).to(gpu)
    prior = IsotropicGauss(dim=d, scale=3.0).to(gpu)
    f = ebm_mle.hip_tempered_log_prob_and_grads(target, prior)
    x = 2.0 * torch.randn(257, d, device=gpu)
    t = torch.rand(257, 1, device=gpu)
    lp, g = f(t, x)
    lp_t = ((1 - t) * prior.unnorm_log_prob(x) + t * target.unnorm_log_prob(x)).flatten()
    g_t = (1 - t) * prior.score(x) + t * target.score(x)
    assert float(((lp - lp_t).abs() / lp_t.abs().clamp(min=1)).max()) < 1e-5 and float((g - g_t).abs().max()) < 1e-4

    times = torch.linspace(1.0, 0.0, n_levels, device=gpu).view(-1, 1, 1).repeat(1, B, 1)
    steps = torch.full((n_levels, B, 1), 0.05, device=gpu)
    samples, steps, diags = ebm_mle.smc_sampler(prior.sample((B,)), times, f, 8, 4, steps, reweight_threshold=1.0)  # resample at every level: the returned particles are then unweighted
    final = samples[0, -1]
    mode = torch.cdist(final, loc.to(gpu)).argmin(dim=1)
    freq = torch.bincount(mode, minlength=3).float().cpu() / B
    print("smc mode frequencies", freq.tolist(), "ess", [round(float(v), 2) for v in diags["ess"]], "acc", float(diags["local_acc"].mean()))
    assert float((freq - wts).abs().max()) < 0.06
    assert 0.4 < float(diags["local_acc"].mean()) <= 1.0 and float(diags["ess"].min()) > 0.0

    steps = torch.full((n_levels, 512, 1), 0.05, device=gpu)
    rs, _, rd = ebm_mle.re_sampler(final[:512].clone(), times[:, :512], f, 4, 20, 8, steps)
    assert rs.shape == (n_levels, 8, 512, d) and bool(torch.isfinite(rs).all()) and 0.0 <= float(rd["swap_acc"]) <= 1.0
    near = torch.cdist(rs[0, -1], loc.to(gpu)).min(dim=1).values
    assert float(near.mean()) < 2.0  # the t = 1 level stays on the modes


# ---- Bayesian logistic-regression targets through the drop-in make_model (BASELINE.json configs[3]) -----------------------------
# The reference's data/*.pkl are pickles and are never loaded: the design matrices here are synthetic, of each data set's
# feature count (conf/target/*.yaml: dim - 1) and a plausible number of rows; parity on the REAL data is unpinned.
LOGREG_SHAPES = {"sonar": (166, 60), "ionosphere": (280, 33), "cancer": (455, 30), "credit": (800, 24)}


def _synthetic_logreg(name, seed=7):
    n, f = LOGREG_SHAPES[name]
    g = torch.Generator().manual_seed(seed + n)
    X = (1e-4 + (1 - 1e-4) * torch.rand(n, f, generator=g) ** 2).float()
    y = (torch.rand(n, generator=g) < 0.47).float()
    return X, y


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sonar", "ionosphere", "cancer", "credit"])
def test_make_model_cmcd_on_logistic_regression_targets(gpu, name):
    """make_model('cmcd', 'gaussian', 'lv', 'em', 'target_informed_zero_init', 'uniform', ...) -- experiments/benchmark_utils.py:96-265
    with target 'sonar' | 'ionosphere' | 'cancer' | 'credit' (:84-91) -- builds, evaluates and takes a training step on the HIP engine.
    sonar's design matrix sits in LDS; the larger ones are read through L2 (sdeng_api.hip prepare_logreg)."""
    from sde_sampler_lrds_amd.distr.logistic_regression import LogisticRegression, register_dataset
    from sde_sampler_lrds_amd.experiments.benchmark_utils import LOGREG_TARGETS
    X, y = _synthetic_logreg(name)
    register_dataset(name, X, y)
    d = LOGREG_TARGETS[name]["dim"]
    assert X.shape[1] + 1 == d
    g = torch.Generator().manual_seed(1)
    A = torch.randn(d, d, generator=g)
    details = dict(mean=0.1 * torch.randn(d, generator=g), var=0.01 * A @ A.T + 0.5 * torch.eye(d))
    model = make_model("cmcd", "gaussian", "lv", "em", "target_informed_zero_init", "uniform", details, make_target_details(name),
                       dict(train_steps=2, train_batch_size=128, eval_batch_size=1024), optim_details=dict(lr=1e-3), n_steps=16)
    assert isinstance(model.target, LogisticRegression) and model.target.dim == d and model.target.data_type == name
    res = model.evaluate()
    assert res.samples.shape == (1024, d) and torch.isfinite(res.samples).all() and torch.isfinite(res.weights).all()
    assert math.isfinite(res.log_norm_const_preds["log_norm_const_is"])
    m = model.step(0)
    assert math.isfinite(m["train/loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ionosphere", "credit"])
def test_logreg_design_matrix_through_l2_matches_oracle(gpu, name):
    """The CMCD step loop with a design matrix too large for LDS (images read through L2) against the oracle, identical seeds."""
    from oracle import sde_oracle as orc
    from sde_sampler_lrds_amd.distr.gauss import GaussFull
    from sde_sampler_lrds_amd.distr.logistic_regression import LogisticRegression
    from sde_sampler_lrds_amd.eq.sdes import ControlledLangevinSDE
    from sde_sampler_lrds_amd.losses import oc
    from sde_sampler_lrds_amd.experiments.baseline_configs import _score_ctrl
    from tests import golden_cases as gc
    X, y = _synthetic_logreg(name)
    torch.manual_seed(5)
    target = LogisticRegression(X_train=X, y_train=y, intercept_mean=-2.5, intercept_scale=0.5, weight_scale=4.5)
    d = target.dim
    A = torch.randn(d, d)
    cov, mean = 0.01 * A @ A.T + 0.5 * torch.eye(d), 0.1 * torch.randn(d)
    prior = GaussFull(dim=d, loc=mean, cov=cov)
    sde = ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=1.0, terminal_t=1.0, clip_score=1e5)
    ctrl = _score_ctrl(d, target)
    for mod in (target, prior, sde, ctrl):
        mod.to(gpu)
    loss = oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
    loss.seed = 9
    B, N = 96, 10
    ts = torch.linspace(0.0, 1.0, N + 1, device=gpu)
    x0 = (mean + torch.randn(B, d) @ torch.linalg.cholesky(cov).T).to(gpu)
    x, rnd, _ = loss.simulate(ts, x0, target.unnorm_log_prob, initial_log_prob=prior.log_prob, train=False)
    otgt = orc.LogReg(X, y, 4.5, -2.5, 0.5)
    oprior = orc.GaussFull(mean, cov)
    octrl = orc.Ctrl({k: v.detach().cpu() for k, v in ctrl.state_dict().items()}, "score", clip_model=1e4, target_score=otgt.score, clip_score=1e4,
                     scale_score=1.0)
    from oracle.baseline_oracles import PerturbedNoise

    def run(noise):
        with torch.no_grad():
            return orc.simulate_cmcd(ts.cpu(), x0.cpu(), octrl, otgt.score, oprior.score, 1.0, 1.0, 1e5, otgt.logp, oprior.logp, noise)
    ox, ornd, _ = run(orc.PhiloxNoise(9))
    scale = max(1.0, float(ornd.abs().max()))
    sens = 0.0
    for salt in range(2):  # what the kernel's 1.2e-6 Box-Muller error does to this case (tests/test_gpu_parity.py)
        px, prnd, _ = run(PerturbedNoise(orc.PhiloxNoise(9), salt=salt))
        sens = max(sens, gc.rel_err(px, ox), float((prnd - ornd).abs().max()) / scale)
    ex = gc.rel_err(x.cpu(), ox)
    er = float((rnd.cpu().flatten() - ornd.flatten()).abs().max()) / scale
    tol = max(1e-5, 10 * sens)
    print(f"cmcd logreg {name} ({X.shape[0]} rows, through L2): x_N {ex:.2e}, rnd {er:.2e}  (tolerance {tol:.1e})")
    assert ex < tol and er < tol


# ---- SURVEY 8f-4 on the HIP path: the reference's own sampler output, with the HIP kernels as the density ----------------------
class _CpuGenerator:
    """Draw every random number of the samplers from torch's CPU generator (the one the fixtures were generated with) and move it to
    the chains' device: same seed, same consumption order -> the reference's random numbers, on GPU tensors."""

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in ("randn", "rand", "randn_like", "rand_like", "multinomial")}
        s = self.saved

        def shaped(fn):
            def wrap(*size, device=None, dtype=None, generator=None, **kw):
                return fn(*size, dtype=dtype, **kw).to(device or "cpu")
            return wrap
        torch.randn, torch.rand = shaped(s["randn"]), shaped(s["rand"])
        torch.randn_like = lambda t, **kw: s["randn"](t.shape, dtype=t.dtype).to(t.device)
        torch.rand_like = lambda t, **kw: s["rand"](t.shape, dtype=t.dtype).to(t.device)
        torch.multinomial = lambda w, n, replacement=False, **kw: s["multinomial"](w.cpu(), n, replacement=replacement).to(w.device)
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(torch, n, f)


@pytest.mark.gpu
@pytest.mark.parametrize("name", __import__("tests.golden_cases", fromlist=["x"]).SAMPLER_CASES)
def test_annealed_samplers_on_hip_densities_match_reference_fixture(gpu, name):
    """additions/ebm_mle.py smc_sampler / re_sampler (MALA / ULA, preconditioned, PDDS) run on the GPU with
    ``hip_tempered_log_prob_and_grads`` -- log-densities and scores from ``sdeng_dist_eval`` -- as the annealing path, against the
    output of the reference's own samplers (tests/golden/smc_*.npz, re_*.npz): same inputs, same random numbers.  The fixtures' path
    is N(0, 9 I)^(1-t) (N(+2, .3 I) + N(-2, .3 I))^t in unnormalised form; the HIP densities are normalised, which shifts every level's
    log-density by a constant -- acceptance ratios, importance weights and swap ratios do not see it."""
    from sde_sampler_lrds_amd.additions import ebm_mle
    from sde_sampler_lrds_amd.distr.gauss import GMM, IsotropicGauss
    from tests import golden_cases as gc
    c = gc.load(name)
    m = c.meta
    d = m["d"]
    target = GMM(dim=d, loc=torch.stack([2.0 * torch.ones(d), -2.0 * torch.ones(d)]), scale=math.sqrt(0.3) * torch.ones(2, d),
                 mixture_weights=torch.ones(2)).to(gpu)
    prior = IsotropicGauss(dim=d, scale=3.0).to(gpu)
    hip = ebm_mle.hip_tempered_log_prob_and_grads(target, prior)
    fn = (lambda t, x: hip(1.0 - t, x)) if m.get("pdds") else hip
    # the path itself: HIP vs the fixtures' closed form (up to the per-level constant)
    xq, tq = 2.5 * torch.randn(300, d), torch.rand(300, 1)
    lp_h, g_h = hip(tq.to(gpu), xq.to(gpu))
    lp_c, g_c = gc.tempered_log_prob_and_grads(tq, xq)
    const = (1 - tq[:, 0]) * (-0.5 * d * math.log(2 * math.pi * 9.0)) + tq[:, 0] * (math.log(0.5) - 0.5 * d * math.log(2 * math.pi * 0.3))
    # (gradient: between the modes the two component terms, each ~ m / v = 6.7, cancel -- 1e-5 of THOSE is the resolution of either formula)
    assert gc.rel_err(lp_h.cpu(), lp_c + const) < 1e-5 and float((g_h.cpu() - g_c).abs().max()) < 1e-5 * (2.0 / 0.3) * 2
    x_init, times, steps = gc.sampler_inputs(m)
    kw = dict(m["kw"], **{k: v.to(gpu) for k, v in gc.sampler_precond(m).items()})
    if m.get("pdds"):
        from sde_sampler_lrds_amd.eq.sdes import VP
        kw.update(use_pdds_weights=True, sde=VP(0.1, 10.0, 1.0, terminal_t=1.0).to(gpu))
    torch.manual_seed(m["seed"])
    with _CpuGenerator():
        if m["sampler"] == "smc":
            samples, steps_out, diags = ebm_mle.smc_sampler(x_init.to(gpu), times.to(gpu), fn, m["n_warm"], m["n_steps"], steps.clone().to(gpu), **kw)
        else:
            samples, steps_out, diags = ebm_mle.re_sampler(x_init.to(gpu), times.to(gpu), fn, kw.pop("swap_frequency"), m["n_warm"], m["n_steps"],
                                                           steps.clone().to(gpu), **kw)
    assert samples.shape == c["samples"].shape
    err = (samples.cpu() - c["samples"]).abs().amax(dim=(0, 1, 3))  # per chain
    print(f"{name}: chains {err.numel()}, max |dx| {float(err.max()):.2e}, chains off by > 1e-4: {int((err > 1e-4).sum())}")
    assert float(err.max()) < 1e-4
    assert float((steps_out.reshape(m["n_levels"], m["B"], 1).cpu() - c["steps_out"]).abs().max()) < 1e-6
    for k, v in diags.items():
        assert float((torch.as_tensor(v).float().cpu() - c["diag_" + k]).abs().max()) < 1e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize("use_ula", [False, True])
def test_native_langevin_moves_equal_the_host_composition(gpu, use_ula):
    """sdeng_langevin_moves (one launch for all moves of a level) against the move-by-move composition of additions/mcmc.py over
    ``sdeng_dist_eval`` -- same random numbers (torch's generator in the reference's order) -- on a larger case than the fixtures:
    600 chains, d = 24, 12 moves, per-chain tempering weights and step sizes, step-size adaptation on."""
    from sde_sampler_lrds_amd import engine as E
    from sde_sampler_lrds_amd.additions import ebm_mle, mcmc
    from sde_sampler_lrds_amd.distr.gauss import GMM, IsotropicGauss
    torch.manual_seed(4)
    d, B, K = 24, 600, 12
    target = GMM(dim=d, loc=2.0 * torch.randn(3, d), scale=0.5 + 0.3 * torch.rand(3, d), mixture_weights=torch.tensor([0.5, 0.3, 0.2])).to(gpu)
    prior = IsotropicGauss(dim=d, scale=2.5).to(gpu)
    f = ebm_mle.hip_tempered_log_prob_and_grads(target, prior)
    t = torch.rand(B, 1, device=gpu)
    x0 = 2.0 * torch.randn(B, d, device=gpu)
    step0 = (0.02 + 0.05 * torch.rand(B, 1, device=gpu))
    # host composition
    torch.manual_seed(11)
    x, step = x0.clone(), step0.clone()
    lp, grad = f(t, x)
    xs_ref = []
    for _ in range(K):
        if use_ula:
            x, lp, grad = mcmc.ula_step(x, lp, grad, lambda y: f(t, y), step)
        else:
            x, lp, grad, log_acc = mcmc.mala_step(x, lp, grad, lambda y: f(t, y), step)
            step = mcmc.heuristics_step_size(step, log_acc, target_acceptance=0.75)
        xs_ref.append(x.clone())
    # one launch
    torch.manual_seed(11)
    xn = x0.clone()
    lpn, gn = f(t, xn)
    lpn, gn, stn = lpn.contiguous().clone(), gn.contiguous().clone(), step0.reshape(-1).clone()
    got, acc, last = E.langevin_moves(target, prior, xn, lpn, gn, stn, K, t=t.reshape(-1), unadjusted=use_ula, target_acceptance=0.0 if use_ula else 0.75)
    err = (got - torch.stack(xs_ref)).abs().amax(dim=(0, 2))
    flipped = int((err > 1e-3).sum())  # an accept / reject decision within round-off of its threshold flips the chain
    print(f"native moves ({'ULA' if use_ula else 'MALA'}): max |dx| over the chains that made the same decisions {float(err[err <= 1e-3].max()):.2e}; "
          f"{flipped} of {B} chains flipped a decision")
    assert flipped <= 2 and float(err[err <= 1e-3].max()) < 2e-5
    same = err <= 1e-3
    assert float((stn - step.reshape(-1))[same].abs().max()) < 1e-7 and float((lpn - lp)[same].abs().max()) < 1e-3
    assert torch.equal(got[-1], xn)


@pytest.mark.gpu
def test_native_moves_with_in_kernel_noise_sample_the_target(gpu):
    """NATIVE_NOISE (Philox draws inside the kernel): an SMC run moves a wide Gaussian onto a 3-mode mixture with the right mode weights."""
    from sde_sampler_lrds_amd.additions import ebm_mle
    from sde_sampler_lrds_amd.distr.gauss import GMM, IsotropicGauss
    torch.manual_seed(0)
    d, B, n_levels = 4, 8192, 12
    loc = torch.tensor([[3.0, 0, 0, 0], [-3.0, 0, 0, 0], [0, 3.0, 0, 0]])
    wts = torch.tensor([0.5, 0.3, 0.2])
    target = GMM(dim=d, loc=loc, scale=0.6 * torch.ones(3, d), mixture_weights=wts.clone()).to(gpu)
    prior = IsotropicGauss(dim=d, scale=3.0).to(gpu)
    f = ebm_mle.hip_tempered_log_prob_and_grads(target, prior)
    times = torch.linspace(1.0, 0.0, n_levels, device=gpu).view(-1, 1, 1).repeat(1, B, 1)
    steps = torch.full((n_levels, B, 1), 0.05, device=gpu)
    ebm_mle.NATIVE_NOISE = True
    try:
        samples, steps, diags = ebm_mle.smc_sampler(prior.sample((B,)), times, f, 8, 4, steps, reweight_threshold=1.0)
    finally:
        ebm_mle.NATIVE_NOISE = False
    final = samples[0, -1]
    mode = torch.cdist(final, loc.to(gpu)).argmin(dim=1)
    frac = torch.bincount(mode, minlength=3).float() / B
    print("native-noise SMC mode fractions", frac.tolist(), "local acc", diags["local_acc"][0].item())
    assert float((frac.cpu() - wts).abs().max()) < 0.05 and 0.3 < float(diags["local_acc"][0]) <= 1.0
