"""SURVEY 8a-11 on the GPU: the initial particles drawn by the engine.

x0 = prior.sample((B,)) is a pure function of (seed, global particle index): x0 = loc + scale * z with z the Philox normals of
stream 1 at step 0 -- the definition the fixtures were generated with (tests/golden/gen_golden.py: x0 = philox_normal(seed, 0, 0, B, d,
stream=1)).  Checked here: the standalone sampler against the oracle's CPU definition for every prior kind, shard independence, and
that a simulate() which leaves x0 to the engine equals, bit for bit, the same simulate() fed the x0 tensor of sdeng_sample_x0 -- and the
reference's fixture outputs in 'identical seeds' mode."""
import pytest
import torch

from oracle import sde_oracle as orc
from sde_sampler_lrds_amd import engine as E
from sde_sampler_lrds_amd.distr.delta import Delta
from sde_sampler_lrds_amd.distr.gauss import Gauss, GaussFull, IsotropicGauss
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
from tests import build_cases as bc
from tests import golden_cases as gc

Z_TOL = 2.5e-6  # hardware sin/cos/log2 Box-Muller vs libm: measured 1.2e-6 max abs on z


def _priors(d):
    g = torch.Generator().manual_seed(d)
    A = torch.randn(d, d, generator=g)
    cov = 0.05 * A @ A.T + 0.5 * torch.eye(d)
    loc = torch.randn(d, generator=g)
    scale = 0.2 + torch.rand(d, generator=g)
    return {
        "iso": (IsotropicGauss(dim=d, loc=0.3, scale=1.7), lambda z: 0.3 + 1.7 * z, 1.7),
        "diag": (Gauss(dim=d, loc=loc, scale=scale), lambda z: loc + scale * z, float(scale.max())),
        "delta": (Delta(dim=d, loc=loc), lambda z: loc.expand_as(z), 0.0),
        "full": (GaussFull(dim=d, loc=loc, cov=cov), lambda z: loc + z @ torch.linalg.cholesky(cov).T, float(torch.linalg.cholesky(cov).abs().sum(1).max())),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("d", [2, 61, 100, 128])
@pytest.mark.parametrize("kind", ["iso", "diag", "delta", "full"])
def test_sample_x0_matches_oracle_definition(gpu, kind, d):
    prior, want_fn, amp = _priors(d)[kind]
    prior.to(gpu)
    seed, p0, B = 9, 12345, 333
    got = E.sample_prior(prior, B, seed, particle0=p0, device=gpu).cpu()
    z = orc.philox_normal(seed, 0, p0, B, d, stream=1)
    want = want_fn(z)
    err = float((got - want).abs().max())
    print(f"x0 {kind} d={d}: max abs err {err:.2e}")
    assert err <= Z_TOL * max(amp, 1e-30) + 1e-6 * float(want.abs().max()) * (kind == "full")
    if kind == "delta":
        assert torch.equal(got, want)
    # shard independence: two shards with their global offsets reproduce the single draw bit for bit
    a = E.sample_prior(prior, 100, seed, particle0=p0, device=gpu)
    b = E.sample_prior(prior, B - 100, seed, particle0=p0 + 100, device=gpu)
    assert torch.equal(torch.cat([a, b]).cpu(), got)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["rds_gmm", "pis_phi4", "cmcd_logreg"])
def test_in_kernel_draw_equals_materialised_x0(gpu, cfg):
    """cfg 2 (IsotropicGauss), cfg 3 (Delta: x0 = loc), cfg 4 (GaussFull + initial log-density): x0 drawn by the engine inside
    sdeng_simulate (sampler kernel ahead of the step loop) -- each against the same simulate() fed the x0 tensor of sdeng_sample_x0,
    sharded too."""
    B, N = 5000, 12
    if cfg == "rds_gmm":
        loss, ts, _, args, kw, info = cfgs.build_rds_gmm(gpu, B, N)
        prior = IsotropicGauss(dim=info["d"], scale=1.0).to(gpu)
    elif cfg == "pis_phi4":
        loss, ts, _, args, kw, info = cfgs.build_pis_phi4(gpu, B, N)
        prior = Delta(dim=info["d"]).to(gpu)
    else:
        loss, ts, _, args, kw, info = cfgs.build_cmcd_logreg(gpu, B, N)
        prior = info["prior"]
    loss.seed, loss.particle0 = 21, 0
    draw = E.InitialDraw(prior, B, gpu)
    x0 = draw.tensor(loss.seed, 0)
    if cfg == "pis_phi4":
        assert float(x0.abs().max()) == 0.0
    a = loss.simulate(ts, draw, *args, **kw)
    b = loss.simulate(ts, x0, *args, **kw)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "x0 drawn inside sdeng_simulate differs from sdeng_sample_x0"
    cut = 2003
    s1 = loss.simulate(ts, E.InitialDraw(prior, cut, gpu), *args, **kw)
    loss.particle0 = cut
    s2 = loss.simulate(ts, E.InitialDraw(prior, B - cut, gpu), *args, **kw)
    loss.particle0 = 0
    assert torch.equal(torch.cat([s1[0], s2[0]]), a[0]) and torch.equal(torch.cat([s1[1], s2[1]]), a[1]), "x0 depends on the sharding"
    res = loss.eval(ts, draw, *args, compute_weights=True, return_traj=True, use_ema=False, **{k: v for k, v in kw.items() if k == "initial_log_prob"})
    assert torch.equal(res.xs[0], x0) and torch.equal(res.samples, a[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rds_ei_gmm_d128_k4", "rds_ei_gmm_d128_k16", "rds_em_gmm_d16", "rds_em_gmm_fullcov_d40_k3", "dis_ei_d8"])
def test_identical_seeds_end_to_end_against_reference_fixture(gpu, name):
    """The fixtures' x0 IS the stream-1 draw (IsotropicGauss, scale 1): with x0 left to the engine nothing but the seed goes in, and the
    reference's outputs come out (tolerance of the Philox-mode parity test)."""
    from tests.test_gpu_parity import philox_tol, rnd_err
    c = gc.load(name)
    b = bc.build(c, gpu)
    prior = IsotropicGauss(dim=c.meta["d"], scale=1.0).to(gpu)
    draw = E.InitialDraw(prior, c.meta["B"], gpu)
    assert float((draw.tensor(b["loss"].seed).cpu() - c["x0"]).abs().max()) <= Z_TOL
    x, rnd, _ = b["loss"].simulate(b["ts"], draw, *b["args"], **b["kwargs"])
    ex, ernd = gc.rel_err(x.cpu(), c["out_x"]), rnd_err(rnd, c)
    tol = philox_tol(name)
    print(f"{name}: x0 drawn by the engine: max rel err x_N {ex:.2e}, rnd {ernd:.2e} (tolerance {tol:.1e})")
    assert ex < tol and ernd < tol
