"""Unit parity on the GPU box: the standalone ABI pieces (distribution log-density / score, control forward,
estimators) against the reference's known-answer vectors (tests/golden/unit_vectors.npz) and the oracle."""
import math

import pytest
import torch

from oracle import sde_oracle as orc
from sde_sampler_lrds_amd import engine as E
from sde_sampler_lrds_amd.distr.gauss import GMM, Gauss, GaussFull, IsotropicGauss
from sde_sampler_lrds_amd.distr.logistic_regression import LogisticRegression
from sde_sampler_lrds_amd.distr.phi_four import PhiFour
from sde_sampler_lrds_amd.distr.rings import Rings
from tests import build_cases as bc
from tests import golden_cases as gc


@pytest.mark.gpu
def test_distribution_kernels_match_reference_vectors(gpu):
    c = gc.load("unit_vectors")
    x = c["gmm_x"].to(gpu)
    g = GMM(dim=12, loc=c["gmm_loc"], scale=c["gmm_scale"], mixture_weights=c["gmm_w"].clone()).to(gpu)
    lp, sc = E.dist_eval(g, x)
    assert gc.rel_err(lp.cpu(), c["gmm_logp"]) < 2e-6 and gc.rel_err(sc.cpu(), c["gmm_score"]) < 1e-5
    m = c.meta["iso"]
    lp, sc = E.dist_eval(IsotropicGauss(dim=m["dim"], loc=m["loc"], scale=m["scale"]).to(gpu), x)
    assert gc.rel_err(lp.cpu(), c["iso_logp"]) < 2e-6 and gc.rel_err(sc.cpu(), c["iso_score"]) < 2e-6
    lp, sc = E.dist_eval(Gauss(dim=12, loc=c["gd_loc"], scale=c["gd_scale"]).to(gpu), x)
    assert gc.rel_err(lp.cpu(), c["gd_logp"]) < 2e-6 and gc.rel_err(sc.cpu(), c["gd_score"]) < 2e-6
    lp, sc = E.dist_eval(GaussFull(dim=12, loc=c["gf_loc"], cov=c["gf_cov"]).to(gpu), x)
    assert gc.rel_err(lp.cpu(), c["gf_logp"]) < 1e-5 and gc.rel_err(sc.cpu(), c["gf_score"]) < 1e-5
    m = c.meta["phi"]
    lp, sc = E.dist_eval(PhiFour(a=m["a"], b=m["b"], dim=m["dim"], beta=m["beta"]).to(gpu), c["phi_x"].to(gpu))
    assert gc.rel_err(lp.cpu(), c["phi_logp"]) < 2e-6 and gc.rel_err(sc.cpu(), c["phi_score"]) < 2e-6
    lp, sc = E.dist_eval(Rings(n_reference_samples=10), c["rings_x"].to(gpu))  # reference defaults, distr/rings.py:41-51
    assert gc.rel_err(lp.cpu(), c["rings_logp"]) < 1e-5 and gc.rel_err(sc.cpu(), c["rings_score"]) < 1e-5


@pytest.mark.gpu
def test_logreg_kernel_matches_autograd_fixture(gpu):
    """Closed-form logistic-regression score in HIP vs the reference's autograd score, incl. saturated rows."""
    c = gc.load("cmcd_logreg_d61")
    m = c.meta
    lr = LogisticRegression(c["X"], c["y"], intercept_mean=m["intercept_mean"], intercept_scale=m["intercept_scale"],
                            weight_scale=m["weight_scale"]).to(gpu)
    lp, sc = E.dist_eval(lr, c["score_x"].to(gpu))
    assert gc.rel_err(sc.cpu(), c["score_out"]) < 1e-5
    assert gc.rel_err(lp.cpu(), c["logp_out"]) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rds_ei_gmm_d128_k4", "pis_em_phi4_d100", "dds_two_modes_d2", "dds_rings_d2", "dis_orig_lerp_d8"])
def test_ctrl_forward_matches_oracle(gpu, name):
    """FP32-MFMA drift net + control wrapper at one time vs the oracle (mid-trajectory states)."""
    c = gc.load(name)
    b = bc.build(c, gpu)
    ctrl = b["loss"].generative_ctrl
    x = (c["xs_last2"][0] if "x_mid" not in c.a else c["x_mid"]).to(gpu)
    t = 0.37
    with torch.no_grad():  # host torch forward of a CPU build of the same mirror (== oracle, tested on CPU)
        want = bc.build(c, "cpu")["loss"].generative_ctrl(torch.tensor(t), x.cpu())
    gain, w = 1.0, 0.0
    if type(ctrl).__name__ == "LerpCtrl":
        sde = b["loss"]._sde_cpu()
        gain, w = float(sde.diff(torch.tensor(t), None)), float(torch.tensor(t) / sde.terminal_t)
    got = E.ctrl_forward(ctrl, t, x, score_gain=gain, lerp_w=w)
    err = gc.rel_err(got.cpu(), want)
    print(f"{name}: ctrl forward max rel err {err:.2e}")
    assert err < 5e-6


@pytest.mark.gpu
def test_logz_kernel_matches_oracle(gpu):
    g = torch.Generator().manual_seed(3)
    for B in (1, 7, 4096, 100003):
        rnd = (3.0 * torch.randn(B, 1, generator=g) + 40.0)
        stats, w = E.logz_stats(rnd.to(gpu))
        ref = orc.compute_results(rnd)
        s = stats.cpu()
        assert abs(s[0].item() - ref["elbo"]) < 1e-4 * abs(ref["elbo"])
        assert abs(s[1].item() - ref["log_norm_const_is"]) < 1e-4 * max(1.0, abs(ref["log_norm_const_is"]))
        if B > 1:
            assert abs(s[2].item() - ref["lv_loss"]) < 1e-3 * ref["lv_loss"]
        assert abs(s[3].item() - ref["ess"]) < 1e-3
        assert gc.rel_err(w.cpu(), ref["weights"]) < 1e-5
    # inf / nan propagate instead of trapping (filter() runs on rnd afterwards: losses/oc.py:67-81)
    bad = torch.tensor([[1.0], [float("inf")], [2.0]])
    stats, _ = E.logz_stats(bad.to(gpu))
    assert not math.isfinite(stats[0].item())


@pytest.mark.gpu
def test_logz_layout_contract(gpu):
    """sdeng_logz's 8 statistics, index by index, against parallel.stats_reference -- the function the 2-rank gloo test feeds through
    the product's gather / combine code (tests/test_parallel_cpu.py): the kernel and the multi-GPU combine agree on one layout."""
    from sde_sampler_lrds_amd import parallel
    g = torch.Generator().manual_seed(11)
    for B in (3, 1000, 65537):
        rnd = 2.0 * torch.randn(B, 1, generator=g) + 30.0
        got = E.logz_stats(rnd.to(gpu), want_weights=False)[0].cpu().double()
        want = parallel.stats_reference(rnd).double()
        for i, name in enumerate(("ELBO", "LOGZ", "VAR", "ESS", "MAX", "SUM_EXP", "SUM_EXP2", "SUM")):
            assert getattr(parallel, name) == i
            assert abs(got[i] - want[i]) <= 2e-4 * max(1.0, abs(float(want[i]))), (name, B, float(got[i]), float(want[i]))
        # sharded weights (one rank here): global_weights == softmax(-rnd)
        w, res = parallel.global_weights(rnd.to(gpu))
        assert gc.rel_err(w.cpu(), torch.softmax(-rnd.double(), 0).float()) < 1e-5 and abs(float(w.double().sum()) - 1.0) < 1e-5
