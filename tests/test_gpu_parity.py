"""HIP engine vs the reference's golden vectors and vs the CPU oracle (GPU box only).

Tolerance: BASELINE.json's north_star asks for 1e-5 relative on trajectory endpoints and log-weights
under identical noise.  `rel_err` is |a-b| / max(1,|b|).  The injected-noise runs consume bit-identical
normals to the ones the reference consumed when the fixture was generated."""
import pytest
import torch

from oracle import sde_oracle as orc
from tests import build_cases as bc
from tests import golden_cases as gc

TOL = 1e-5
HIP_CASES = list(gc.SIM_CASES)


def rnd_scale(c):
    """Log-weights are sums of terms as large as the terminal log-densities (|log p| ~ 150-260 at d=100-128,
    whose fp32 ulp alone is 1.5e-5): their error is judged relative to the largest summand, not to the
    (possibly cancelling) total."""
    b = bc.build(c, "cpu")
    x = c["out_x"]
    mags = [c["rnd"].abs()] + [fn(x).view(-1, 1).abs() for fn in b["args"]]
    if "initial_log_prob" in b["kwargs"]:
        mags.append(b["kwargs"]["initial_log_prob"](c["x0"]).view(-1, 1).abs())
    return torch.stack([m.float() for m in mags]).max(dim=0).values.clamp(min=1.0)


def rnd_err(got, c):
    return float(((got.cpu().double() - c["rnd"].double()).abs() / rnd_scale(c).double()).max())


_AMP = {}


def sensitivity(name):
    """How much a ONE-ulp relative perturbation of x0 moves x_N in the fp32 oracle: a lower bound on what any
    two correct fp32 implementations can differ by for this case (a few particles sit near a separatrix
    between mixture modes and amplify round-off ~100x)."""
    if name not in _AMP:
        c, c2 = gc.load(name), gc.load(name)
        c2.a["x0"] = c2.a["x0"] * (1 + 1.2e-7)
        x, r = gc.run_oracle(c)
        x2, r2 = gc.run_oracle(c2)
        _AMP[name] = max(gc.rel_err(x2, x), cov_sensitivity(name, x, r))
    return _AMP[name]


def cov_sensitivity(name, x, r):
    """Full-covariance TARGETS inside a score control (GMMFull.score: the reference inverts the covariances once, in fp32): how much a
    ONE-ulp relative perturbation of the covariance entries moves x_N and the log-weights in the fp32 oracle -- the conditioning of the
    reference's own `torch.linalg.inv(cov)`, which no implementation that forms the precisions another way can undercut.  0 for every other
    case.  (dds_two_modes_full_d5: the running cost sums |score|^2 ~ 1e4 per step; one ulp of the covariance moves rnd by 5-8e-3.)"""
    c = gc.load(name)
    if "tgt_cov" not in c.a:
        return 0.0
    scale = rnd_scale(c)
    worst = 0.0
    g = torch.Generator().manual_seed(7)
    for _ in range(2):
        c2 = gc.load(name)
        e = (torch.randint(0, 2, c2.a["tgt_cov"].shape, generator=g) * 2 - 1).float() * 1.2e-7
        c2.a["tgt_cov"] = c2.a["tgt_cov"] * (1 + (e + e.transpose(1, 2)) / 2)
        x2, r2 = gc.run_oracle(c2)
        worst = max(worst, gc.rel_err(x2, x), float(((r2.double() - r.double()).abs().view(-1, 1) / scale.double().view(-1, 1)).max()))
    return worst


# ---- 'identical seeds' (Philox) mode ----------------------------------------------------------------------------------
# The kernel evaluates Box-Muller with the hardware's v_log / v_sin / v_cos: its normals differ from the oracle's libm evaluation
# of the SAME counters by at most 1.2e-6 absolute (measured, test_philox_kernel_vs_oracle below; about
# 2 ulp at |z| ~ 5).  What that does to x_N and the log-weights is a property of the case, measured here in the oracle itself:
# the same trajectory with every normal moved by +-HW_NOISE_ERR (random signs; the worst of two patterns).  The Philox-mode tests assert max(1e-5, 10 x that) and print
# what they achieved -- 1e-5 wherever the case does not amplify (the north_star's bound), a stated larger bound where it does.
HW_NOISE_ERR = 1.2e-6
_NSENS = {}


from oracle.baseline_oracles import PerturbedNoise  # noqa: E402  (a noise source moved by +-eps per normal, fixed sign pattern)


def noise_sensitivity(name, eubo=False):
    key = (name, eubo)
    if key not in _NSENS:
        c = gc.load(name)
        run = gc.run_oracle_eubo if eubo else gc.run_oracle
        base = orc.PhiloxNoise(c.meta["seed"])
        x, r = run(c, noise=base)
        scale = rnd_scale(c) if not eubo else torch.ones(1)
        worst = 0.0
        for salt in range(2):  # two independent sign patterns: amplification near a separatrix is hit-or-miss
            x2, r2 = run(c, noise=PerturbedNoise(base, salt=salt))
            worst = max(worst, gc.rel_err(x2, x), float(((r2.double() - r.double()).abs().view(-1, 1) / scale.double().view(-1, 1)).max()))
        _NSENS[key] = worst
    return _NSENS[key]


def philox_tol(name, eubo=False):
    extra = 0.0
    if not eubo and "tgt_cov" in gc.load(name).a:
        extra = sensitivity(name)  # (includes the covariance-conditioning probe)
    return max(TOL, 10 * max(noise_sensitivity(name, eubo), extra))


def replay_noise(c, B=None):
    m = c.meta
    B = B or m["B"]
    return torch.stack([orc.philox_normal(m["seed"], k, 0, B, m["d"]) for k in range(m["N"])])


@pytest.mark.gpu
@pytest.mark.parametrize("name", HIP_CASES)
def test_injected_noise_matches_reference_fixture(gpu, name):
    c = gc.load(name)
    b = bc.build(c, gpu)
    x, rnd, xs = b["loss"].simulate(b["ts"], b["x0"], *b["args"], return_traj=True, noise=replay_noise(c).to(gpu), **b["kwargs"])
    torch.cuda.synchronize()
    ex, ernd = gc.rel_err(x.cpu(), c["out_x"]), rnd_err(rnd, c)
    print(f"{name}: max rel err x_N {ex:.2e}, rnd {ernd:.2e} (abs {float((rnd.cpu() - c['rnd']).abs().max()):.2e})")
    tol = max(TOL, 10 * sensitivity(name))  # 1e-5 unless the case itself amplifies one ulp beyond 1e-6
    print(f"   tolerance {tol:.1e}")
    assert ex < tol and ernd < tol
    assert xs.shape == (c.meta["N"] + 1, c.meta["B"], c.meta["d"])
    assert gc.rel_err(xs[-2:].cpu(), c["xs_last2"]) < tol
    assert torch.equal(xs[0].cpu(), c["x0"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", HIP_CASES)
def test_philox_mode_matches_oracle(gpu, name):
    """'Identical seeds': the kernel's in-register Philox stream against the oracle's CPU definition of it."""
    c = gc.load(name)
    b = bc.build(c, gpu)
    x, rnd, _ = b["loss"].simulate(b["ts"], b["x0"], *b["args"], **b["kwargs"])
    torch.cuda.synchronize()
    ex, ernd = gc.rel_err(x.cpu(), c["out_x"]), rnd_err(rnd, c)
    tol = philox_tol(name)
    print(f"{name}: philox-mode max rel err x_N {ex:.2e}, rnd {ernd:.2e}   (tolerance {tol:.1e}; the case moves {noise_sensitivity(name):.1e} "
          f"under a {HW_NOISE_ERR:.1e} perturbation of its normals)")
    assert ex < tol and ernd < tol


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.EUBO_CASES)
def test_compute_eubo_matches_reference_fixture(gpu, name):
    """compute_eubo (noising direction, SDENG_FORM_EUBO) against the reference's output: injected noise, then Philox."""
    c = gc.load(name)
    b = bc.build(c, gpu)
    tgt_scale = torch.stack([fn(c["x0"].to(gpu)).view(-1).abs().cpu() for fn in b["args"]] + [c["rnd"].view(-1).abs()]).max(0).values.clamp(min=1.0)
    for mode, noise, tol in (("injected", replay_noise(c).to(gpu), TOL), ("philox", None, philox_tol(name, eubo=True))):
        x = b["x0"].clone()
        extra = {k: v for k, v in b["kwargs"].items() if k == "initial_log_prob"}  # the DIS loss takes the prior log-density
        rnd = b["loss"].compute_eubo(b["ts"], x, *b["args"], noise=noise, **extra)
        torch.cuda.synchronize()
        ex = gc.rel_err(x.cpu(), c["out_x"]) if "out_x" in c.a else 0.0  # compute_eubo noises x in place, like the reference
        ernd = float(((rnd.cpu().view(-1) - c["rnd"].view(-1)).abs() / tgt_scale).max())
        print(f"{name} [{mode}]: max rel err noised x {ex:.2e}, rnd {ernd:.2e}   (tolerance {tol:.1e})")
        assert ex < tol and ernd < tol


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rds_ei_gmm_d128_k4", "pis_em_phi4_d100", "dds_two_modes_d2"])
def test_eval_results_match_reference(gpu, name):
    c = gc.load(name)
    b = bc.build(c, gpu)
    kw = {k: v for k, v in b["kwargs"].items() if k in ("initial_log_prob",)}
    res = b["loss"].eval(b["ts"], b["x0"], *b["args"], compute_weights=True, return_traj=True, use_ema=False,
                         noise=replay_noise(c).to(gpu), **kw)
    m = c.meta
    assert abs(res.log_norm_const_preds["log_norm_const_is"] - m["log_norm_const_is"]) < 1e-4 * max(1, abs(m["log_norm_const_is"]))
    assert abs(res.metrics["eval/elbo"] - m["elbo"]) < 1e-4 * max(1, abs(m["elbo"]))
    assert abs(res.metrics["eval/lv_loss"] - m["lv_loss"]) < 1e-3 * max(1, abs(m["lv_loss"]))
    assert gc.rel_err(res.weights.cpu(), c["out_weights"]) < 1e-4
    assert res.xs.shape == (len(b["ts"]), *res.samples.shape)  # solver/oc.py:145


@pytest.mark.gpu
def test_partial_tiles_and_sharding_independence(gpu):
    """B not a multiple of 32, and two shards [0,40) + [40,64) reproduce the single-shard run bit for bit."""
    c = gc.load("rds_ei_gmm_d128_k4")
    b = bc.build(c, gpu)
    loss = b["loss"]
    full = loss.simulate(b["ts"], b["x0"], *b["args"])
    loss.particle0 = 0
    a = loss.simulate(b["ts"], b["x0"][:40], *b["args"])
    loss.particle0 = 40
    z = loss.simulate(b["ts"], b["x0"][40:], *b["args"])
    loss.particle0 = 0
    assert torch.equal(torch.cat([a[0], z[0]]), full[0])
    assert torch.equal(torch.cat([a[1], z[1]]), full[1])


@pytest.mark.gpu
def test_philox_kernel_vs_oracle(gpu):
    from sde_sampler_lrds_amd import _lib as L
    out = torch.empty(300, 37, device=gpu)
    L.check(L.lib().sdeng_philox_normal(12345678901234, 7, 1000, 300, 37, 0, out.data_ptr(), None))
    torch.cuda.synchronize()
    ref = orc.philox_normal(12345678901234, 7, 1000, 300, 37)
    err = (out.cpu() - ref).abs().max().item()
    print("philox normal max abs err vs numpy:", err)
    assert err < 2 * HW_NOISE_ERR


@pytest.mark.gpu
def test_multi_step_philox_equals_single_steps(gpu):
    """sdeng_philox_normal_steps (one launch for N steps: the noise a training call keeps) == N calls of sdeng_philox_normal."""
    from sde_sampler_lrds_amd import _lib as L
    from sde_sampler_lrds_amd import engine as E
    N, B, d, seed, p0 = 7, 333, 29, 987654321987, 4096
    many = E.philox_noise(seed, N, B, d, p0, gpu)
    one = torch.empty(B, d, device=gpu)
    for k in range(N):
        L.check(L.lib().sdeng_philox_normal(seed, k, p0, B, d, 0, one.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(many[k], one)
    ref = orc.philox_normal(seed, 3, p0, B, d)
    assert float((many[3].cpu() - ref).abs().max()) < 5e-6


@pytest.mark.gpu
def test_empty_batch_and_bad_descriptor(gpu):
    from sde_sampler_lrds_amd import engine as E
    c = gc.load("rds_ei_gmm_d8_k4")
    b = bc.build(c, gpu)
    x, rnd, _ = b["loss"].simulate(b["ts"], b["x0"][:0], *b["args"])
    assert x.shape == (0, 8) and rnd.shape == (0, 1)
    with pytest.raises(RuntimeError):
        b["loss"].simulate(b["ts"].cpu(), b["x0"].cpu(), *b["args"])  # no CPU path
    b["loss"].reference_ctrl = lambda t, x: x  # opaque callable
    with pytest.raises(E.UnsupportedByEngine):
        b["loss"].simulate(b["ts"], b["x0"], *b["args"])


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,B", [(128, 4, 48), (20, 3, 37), (16, 2, 16)])
def test_shared_variance_reference_matches_oracle(gpu, d, K, B):
    """Mixture references whose components share one variance vector (the reference's default initialisation,
    solver/oc.py:563-576 with variances_init = const) take the (sum_k p_k m_k - x)/var form of the score in the step
    loop (detected on the device by k_ref_tables); the fixtures all have distinct variances, so this case is checked
    against the oracle directly, under identical injected noise, at the fixtures' tolerance."""
    from oracle import sde_oracle as orc
    from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
    N = 24
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N, d=d, K=K, seed=100 + d)
    z = torch.randn(N, B, d, generator=torch.Generator().manual_seed(d))
    x, rnd, _ = loss.simulate(ts, x0, *args, noise=z.to(gpu), **kw)
    sde = orc.VP(0.1, 10.0, 1.0, 1.0)
    tgt = orc.GMMDiag(info["target"].loc.cpu(), info["target"].scale.cpu(), info["target"].mixture_weights.cpu())
    sd = {k: v.detach().cpu() for k, v in info["ctrl"].state_dict().items()}
    ctrl = orc.Ctrl(sd, "clipped", clip_model=1e4)
    means, var, w = info["means"].cpu(), 0.5 * torch.ones(K, d), torch.ones(K)
    loc0, v0 = sde.marginal_diag(torch.tensor(0.0), means, var)
    refd = orc.GMMDiag(loc0, v0.sqrt(), w)
    with torch.no_grad():
        ox, ornd, _ = orc.simulate_ei_ref(ts.cpu(), x0.cpu(), ctrl, sde, tgt.logp, refd.logp,
                                          lambda t, xx: orc.mog_score(xx, w, *sde.marginal_diag(t, means, var)), orc.InjectedNoise(z))
    # log-weight error relative to the largest summand (the terminal log-densities), as rnd_scale() above
    scale = torch.stack([ornd.flatten().abs(), tgt.logp(ox).flatten().abs(), refd.logp(ox).flatten().abs()]).max(dim=0).values.clamp(min=1.0)
    ex, er = gc.rel_err(x.cpu(), ox), float(((rnd.cpu().flatten() - ornd.flatten()).abs() / scale).max())
    print(f"shared-variance reference d={d} K={K}: x_N {ex:.2e}, rnd {er:.2e}")
    assert ex < TOL and er < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.EULER_CASES)
def test_euler_integrator_matches_reference_fixture(gpu, name):
    """EulerIntegrator.integrate (one launch + interpolation) against the reference's own output under identical Brownian
    increments (through the integrator's ``bm`` hook), then with the kernel's Philox stream against the oracle."""
    from sde_sampler_lrds_amd.eq.integrator import EulerIntegrator
    c = gc.load(name)
    m = c.meta
    sde = bc.build_euler(c, gpu)
    grid_cpu = c["timesteps"]
    incs = torch.stack([orc.philox_normal(m["seed"], k, 0, m["B"], m["d"]) * torch.sqrt(grid_cpu[k + 1] - grid_cpu[k])
                        for k in range(m["N"])]).to(gpu)
    index = {float(v): k for k, v in enumerate(grid_cpu[:-1])}
    out = EulerIntegrator().integrate(sde, ts=c["ts"].to(gpu), x_init=c["x0"].to(gpu), timesteps=grid_cpu.to(gpu),
                                      bm=lambda s, t: incs[index[float(s)]])
    err = gc.rel_err(out.cpu(), c["out_xs"])
    full = EulerIntegrator().integrate(sde, ts=grid_cpu.to(gpu), x_init=c["x0"].to(gpu), timesteps=grid_cpu.to(gpu),
                                       bm=lambda s, t: incs[index[float(s)]])
    err_last = gc.rel_err(full[-1].cpu(), c["out_last"])
    print(f"{name}: injected increments: interpolated states {err:.2e}, x_T {err_last:.2e}")
    assert out.shape == c["out_xs"].shape and err < TOL and err_last < TOL
    assert torch.equal(full[0].cpu(), c["x0"])
    # Philox mode: the kernel's own stream (first call of an integrator with seed 5) vs the oracle's definition of it
    got = EulerIntegrator(seed=5).integrate(sde, ts=c["ts"].to(gpu), x_init=c["x0"].to(gpu), timesteps=grid_cpu.to(gpu))
    want = gc.run_oracle_euler(c, increment=lambda k, s, t, x: orc.philox_normal(5, k, 0, x.shape[0], x.shape[1]) * torch.sqrt(t - s))
    err_p = gc.rel_err(got.cpu(), want)
    print(f"{name}: philox mode {err_p:.2e}")
    assert err_p < TOL  # achieved <= 1.2e-6 on the five fixtures


@pytest.mark.gpu
def test_langevin_solver_reaches_the_target(gpu):
    """solver/langevin.py:36-66 on the engine: with diff_coeff = sqrt(2) the Langevin SDE is stationary at the target; a
    one-component Gaussian target is reached from a wide prior (mean / variance within Monte-Carlo error + O(dt) bias)."""
    from functools import partial

    from sde_sampler_lrds_amd.distr.gauss import GMM, IsotropicGauss
    from sde_sampler_lrds_amd.eq.integrator import EulerIntegrator
    from sde_sampler_lrds_amd.solver.langevin import LangevinSolver
    from sde_sampler_lrds_amd.utils.common import get_timesteps
    d = 6
    loc, scale = torch.linspace(-2.0, 2.0, d).view(1, d), torch.linspace(0.5, 1.0, d).view(1, d)
    target = GMM(dim=d, loc=loc, scale=scale, mixture_weights=torch.ones(1))
    prior = IsotropicGauss(dim=d, scale=3.0)
    solver = LangevinSolver(target, prior, eval_timesteps=partial(get_timesteps, 0.0, 8.0, steps=1600),
                            integrator=EulerIntegrator(dt=None, steps=1600, seed=9), diff_coeff=2.0 ** 0.5, eval_batch_size=16384,
                            eval_expectation_burn=1200, device=gpu)
    res = solver.run()
    assert res.xs.shape == (1601, 16384, d) and res.samples.shape == (16384, d) and bool(torch.isfinite(res.xs).all())
    mean, std = res.samples.mean(0).cpu(), res.samples.std(0).cpu()
    print("langevin solver: mean err", float((mean - loc[0]).abs().max()), "std err", float((std / scale[0] - 1).abs().max()))
    assert float((mean - loc[0]).abs().max()) < 0.06 and float((std / scale[0] - 1).abs().max()) < 0.05
    want_sq = float((loc ** 2 + scale ** 2).sum())
    assert abs(float(res.expectation_preds["square"]) - want_sq) < 0.03 * want_sq
    assert "eval/sample_time" in res.metrics


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,B,N", [(1, 2, 1, 3), (128, 4, 17, 1), (33, 3, 16, 2), (5, 4, 100, 0), (64, 2, 15, 7), (17, 9, 33, 5)])
def test_extreme_shapes_match_oracle(gpu, d, K, B, N):
    """Smallest and ragged shapes: one particle, one feature, one step, zero steps (x_N = x_0, the log-weight is the terminal
    cost alone), a batch one short of a tile, d one past a tile boundary; injected noise."""
    from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, max(N, 1), d=d, K=K, seed=300 + d)
    if N == 0:
        ts = ts[:1]
    z = torch.randn(N, B, d, generator=torch.Generator().manual_seed(d + N))
    x, rnd, xs = loss.simulate(ts, x0, *args, noise=z.to(gpu) if N else None, return_traj=True, **kw)
    sde = orc.VP(0.1, 10.0, 1.0, 1.0)
    tgt = orc.GMMDiag(info["target"].loc.cpu(), info["target"].scale.cpu(), info["target"].mixture_weights.cpu())
    ctrl = orc.Ctrl({k: v.detach().cpu() for k, v in info["ctrl"].state_dict().items()}, "clipped", clip_model=1e4)
    means, var, w = info["means"].cpu(), 0.5 * torch.ones(K, d), torch.ones(K)
    loc0, v0 = sde.marginal_diag(torch.tensor(0.0), means, var)
    refd = orc.GMMDiag(loc0, v0.sqrt(), w)

    def oracle(xstart):
        with torch.no_grad():
            if N == 0:
                return xstart, refd.logp(xstart).view(-1, 1) - tgt.logp(xstart).view(-1, 1)
            return orc.simulate_ei_ref(ts.cpu(), xstart, ctrl, sde, tgt.logp, refd.logp,
                                       lambda t, xx: orc.mog_score(xx, w, *sde.marginal_diag(t, means, var)), orc.InjectedNoise(z))[:2]

    ox, ornd = oracle(x0.cpu())
    # a grid of one or two steps over [0, 1] takes giant steps (x gain 12, score gain 23): like the fixture tests, allow 10x what a
    # one-ulp perturbation of x0 does to the oracle itself
    tol = max(TOL, 10 * gc.rel_err(oracle(x0.cpu() * (1 + 1.2e-7))[0], ox))
    scale = torch.stack([ornd.flatten().abs(), tgt.logp(ox).flatten().abs(), refd.logp(ox).flatten().abs()]).max(dim=0).values.clamp(min=1.0)
    ex, er = gc.rel_err(x.cpu(), ox), float(((rnd.cpu().flatten() - ornd.flatten()).abs() / scale).max())
    print(f"shape d={d} K={K} B={B} N={N}: x_N {ex:.2e}, rnd {er:.2e} (tolerance {tol:.1e})")
    assert x.shape == (B, d) and rnd.shape == (B, 1) and xs.shape == (N + 1, B, d)
    assert ex < tol and er < tol
    assert torch.equal(xs[0], x0) and torch.equal(xs[-1], x)


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,B,N,form", [(16, 3, 40, 12, "ei"), (40, 2, 33, 10, "em"), (128, 4, 48, 16, "ei"), (128, 3, 40000, 24, "ei")])
def test_full_covariance_reference_matches_oracle(gpu, d, K, B, N, form):
    """RDS with a FULL-covariance mixture reference (score_mog_full, distr/gauss.py:110-121; eq/sdes.py:329-345): precision images
    staged per workgroup, P (m - x) on the matrix pipe.  Checked against the oracle's covariance-form restatement (linalg.solve per
    step) under identical injected noise; the large case also checks reruns bit for bit and the Philox stream."""
    from sde_sampler_lrds_amd.distr.gauss import ManyModes
    from sde_sampler_lrds_amd.eq.sdes import VP
    from sde_sampler_lrds_amd.experiments.baseline_configs import _net
    from sde_sampler_lrds_amd.losses import oc
    from sde_sampler_lrds_amd.models.reparam import ClippedCtrl
    from sde_sampler_lrds_amd.reference import MarginalReference
    torch.manual_seed(500 + d)
    sde = VP(0.1, 10.0, 1.0, terminal_t=1.0)
    target = ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0, n_reference_samples=10)
    ctrl = ClippedCtrl(base_model=_net(d), clip_model=1e4)
    means = target.loc.clone() + 0.1 * torch.randn(K, d)
    A = torch.randn(K, d, d) / d ** 0.5
    cov = 0.3 * A @ A.transpose(-1, -2) + 0.4 * torch.eye(d)
    wts = torch.rand(K) + 0.5
    ref = MarginalReference(sde, "gmm", means_init=means, variances_init=cov, weights_init=wts)
    for m in (sde, target, ctrl, ref):
        m.to(gpu)
    cls = oc.EIReferenceSDELoss if form == "ei" else oc.EMReferenceSDELoss
    loss = cls(ctrl, ctrl, sde=sde, method="lv", reference_ctrl=ref)
    ts = torch.linspace(0.0, 1.0, N + 1, device=gpu)
    x0 = torch.randn(B, d, generator=torch.Generator().manual_seed(d)).to(gpu)
    pb = min(B, 48)
    z = torch.randn(N, B, d, generator=torch.Generator().manual_seed(d + 1))
    args = (target.unnorm_log_prob, ref.reference_distr.log_prob)
    x, rnd, _ = loss.simulate(ts, x0, *args, noise=z.to(gpu))
    osde = orc.VP(0.1, 10.0, 1.0, 1.0)
    tgt = orc.GMMDiag(target.loc.cpu(), target.scale.cpu(), target.mixture_weights.cpu())
    octrl = orc.Ctrl({k: v.detach().cpu() for k, v in ctrl.state_dict().items()}, "clipped", clip_model=1e4)
    loc0, cov0 = osde.marginal_full(torch.tensor(0.0), means, cov)
    refd = orc.GMMFullCov(loc0, cov0, wts.clone())
    fn = orc.simulate_ei_ref if form == "ei" else orc.simulate_em_ref

    def oracle(xstart):
        with torch.no_grad():
            return fn(ts.cpu(), xstart, octrl, osde, tgt.logp, refd.logp,
                      lambda t, xx: orc.mog_score_full(xx, wts.clone(), *osde.marginal_full(t, means, cov)), orc.InjectedNoise(z[:, :pb]))[:2]

    ox, ornd = oracle(x0[:pb].cpu())
    tol = max(TOL, 10 * gc.rel_err(oracle(x0[:pb].cpu() * (1 + 1.2e-7))[0], ox))  # coarse grids amplify one ulp beyond 1e-6
    scale = torch.stack([ornd.flatten().abs(), tgt.logp(ox).flatten().abs(), refd.logp(ox).flatten().abs()]).max(dim=0).values.clamp(min=1.0)
    ex = gc.rel_err(x[:pb].cpu(), ox)
    er = float(((rnd[:pb].cpu().flatten() - ornd.flatten()).abs() / scale).max())
    print(f"full-covariance reference d={d} K={K} B={B} N={N} {form}: x_N {ex:.2e}, rnd {er:.2e} (tolerance {tol:.1e})")
    assert ex < tol and er < tol
    if B > 1000:
        a = loss.simulate(ts, x0, *args)
        b = loss.simulate(ts, x0, *args)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and bool(torch.isfinite(a[1]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("wrapper", ["lerp", "cancel_drift"])
@pytest.mark.parametrize("loss_kind", ["em", "dds", "dis_ei"])
def test_control_wrappers_with_per_step_gains_in_every_loss(gpu, wrapper, loss_kind):
    """LerpCtrl (g(t), t/T) and CancelDriftCtrl (drift/g, g/2) carry per-step gains of their own SDE; the fixtures pin them under
    TimeReversalLoss / DiscreteTimeReversalLossEI, here they run under the EM loss (PIS form), DDS (whose loss has no SDE of its
    own) and DIS-EI against the oracle, injected noise."""
    from sde_sampler_lrds_amd.distr.gauss import GMM, IsotropicGauss
    from sde_sampler_lrds_amd.eq.sdes import VP
    from sde_sampler_lrds_amd.losses import oc
    from sde_sampler_lrds_amd.models.mlp import FourierMLP, TimeEmbed
    from sde_sampler_lrds_amd.models.reparam import CancelDriftCtrl, LerpCtrl
    torch.manual_seed(7)
    d, B, N = 8, 64, 32
    sde = VP(0.1, 10.0, 1.0, terminal_t=1.0)
    loc = 2.0 * torch.randn(3, d)
    target = GMM(dim=d, loc=loc, scale=0.7 * torch.ones(3, d), mixture_weights=torch.tensor([0.5, 0.3, 0.2]))
    prior = IsotropicGauss(dim=d, scale=1.0)
    net = FourierMLP(dim=d, activation=torch.nn.GELU(), num_layers=4, channels=64)
    sm = TimeEmbed(dim_out=1, activation=torch.nn.GELU(), num_layers=4, channels=64)
    with torch.no_grad():
        net.out_layer.weight.uniform_(-0.1, 0.1)
        sm.out_layer.bias.fill_(0.7)
    common = dict(base_model=net, score_model=sm, target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                  scale_score=1.0, sde=sde)
    ctrl = LerpCtrl(**common, prior_score=prior.score) if wrapper == "lerp" else CancelDriftCtrl(**common)
    for m in (sde, target, prior, ctrl):
        m.to(gpu)
    ts = torch.linspace(0.0, 1.0, N + 1, device=gpu)
    x0 = torch.randn(B, d, generator=torch.Generator().manual_seed(1)).to(gpu)
    z = torch.randn(N, B, d, generator=torch.Generator().manual_seed(2))
    osde = orc.VP(0.1, 10.0, 1.0, 1.0)
    tgt, opr = orc.GMMDiag(loc, 0.7 * torch.ones(3, d), torch.tensor([0.5, 0.3, 0.2])), orc.IsoGauss(d, 0.0, 1.0)
    octrl = orc.Ctrl({k: v.detach().cpu() for k, v in ctrl.state_dict().items()}, wrapper, clip_model=1e4, target_score=tgt.score, clip_score=1e4,
                     scale_score=1.0, sde=osde, prior_score=opr.score)
    noise = orc.InjectedNoise(z)
    with torch.no_grad():
        if loss_kind == "em":
            loss = oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv")
            x, rnd, _ = loss.simulate(ts, x0, target.unnorm_log_prob, prior.log_prob, noise=z.to(gpu))
            ox, ornd, _ = orc.simulate_em_ref(ts.cpu(), x0.cpu(), octrl, osde, tgt.logp, opr.logp, None, noise)
        elif loss_kind == "dds":
            loss = oc.ExponentialIntegratorSDELoss(ctrl, ctrl, sde=None, method="lv", alpha=1.0, sigma=1.0)
            x, rnd, _ = loss.simulate(ts, x0, target.unnorm_log_prob, prior.log_prob, compute_ito_int=True, noise=z.to(gpu))
            ox, ornd = orc.simulate_dds(ts.cpu(), x0.cpu(), octrl, 1.0, 1.0, tgt.logp, opr.logp, noise)[:2]
        else:
            loss = oc.DiscreteTimeReversalLossEI(ctrl, ctrl, sde=sde, method="lv")
            x, rnd, _ = loss.simulate(ts, x0, target.unnorm_log_prob, initial_log_prob=prior.log_prob, train=False, noise=z.to(gpu))
            ox, ornd = orc.simulate_dis_ei(ts.cpu(), x0.cpu(), octrl, osde, tgt.logp, opr.logp, noise)[:2]
    scale = torch.stack([ornd.flatten().abs(), tgt.logp(ox).flatten().abs(), opr.logp(ox).flatten().abs()]).max(dim=0).values.clamp(min=1.0)
    ex, er = gc.rel_err(x.cpu(), ox), float(((rnd.cpu().flatten() - ornd.flatten()).abs() / scale).max())
    print(f"{wrapper} under {loss_kind}: x_N {ex:.2e}, rnd {er:.2e}")
    assert ex < TOL and er < TOL and float(ox.abs().max()) < 1e3  # achieved <= 8.4e-7 on the six combinations
