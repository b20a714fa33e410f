"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol of include/sdeng.h
(no compute without a GPU), the descriptor compiler and coefficient tables, the API surface of the loss
classes, and that there is no CPU execution path."""
import ctypes
import os
import re

import pytest
import torch

from sde_sampler_lrds_amd import _lib as L
from sde_sampler_lrds_amd import engine as E
from tests import build_cases as bc
from tests import golden_cases as gc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sdeng.h")).read()
    declared = set(re.findall(r"\b(sdeng_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(L.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/sdeng.h but not exported"
    assert set(L.EXPORTS) <= declared
    lib.sdeng_abi_version.restype = ctypes.c_int
    assert lib.sdeng_abi_version() == L.ABI_VERSION


def test_struct_sizes_match_header_layout():
    # sizes the C compiler produces for the same field lists (checked once with hipcc: see DESIGN.md)
    assert ctypes.sizeof(L.Dist) == 64
    assert ctypes.sizeof(L.TimeEmbed) == 104
    assert ctypes.sizeof(L.Ref) == 48
    assert ctypes.sizeof(L.Net) == 296 and ctypes.sizeof(L.Desc) == 736  # gcc on include/sdeng.h (ABI 3)


def test_workspace_bytes_and_bad_descriptors_without_gpu():
    lib = L.lib()
    d = L.Desc()
    d.abi_version, d.B, d.d, d.N = L.ABI_VERSION, 1024, 128, 256
    d.ref.kind, d.ref.k = L.REF_GMM_DIAG, 4
    need = lib.sdeng_workspace_bytes(ctypes.byref(d))
    assert need > 256 * 4 * 2 * 128 * 4  # at least the per-step reference tables
    d.d = 129
    assert lib.sdeng_workspace_bytes(ctypes.byref(d)) == 0
    rc = lib.sdeng_simulate(ctypes.byref(d), None)
    assert rc == L.E_INVALID and b"d" in lib.sdeng_last_error()
    d.d, d.abi_version = 128, 99
    assert lib.sdeng_simulate(ctypes.byref(d), None) == L.E_INVALID


@pytest.mark.parametrize("name", [n for n in gc.SIM_CASES if n != "cmcd_logreg_d61"])
def test_descriptor_compiles_on_host(name):
    c = gc.load(name)
    b = bc.build(c, "cpu")
    loss = b["loss"]
    keep = []
    net = E.net_desc(loss.generative_ctrl, "cpu", keep)
    assert net.w_in and net.t_embed.dim_out == 64
    kind, utils = E.resolve_reference(getattr(loss, "reference_ctrl", None))
    ref = E.ref_desc(kind, utils, "cpu", keep)
    assert ref.kind in (L.REF_NONE, L.REF_GAUSS_DIAG, L.REF_GMM_DIAG, L.REF_GMM_FULL)
    assert (ref.kind == L.REF_GMM_FULL) == ("fullcov" in name or "eigen" in name) and bool(ref.eigvecs) == (ref.kind == L.REF_GMM_FULL)
    tgt = E.resolve_logp(b["args"][0])
    if "tgt_cov" in c.a:  # full-covariance mixture target: a descriptor for the score inside the control only (terminal cost via torch)
        with pytest.raises(E.UnsupportedByEngine):
            E.dist_desc(tgt[0], "cpu", keep)
        ds = E.dist_desc(tgt[0], "cpu", keep, score_only=True)
        assert ds.kind == L.DIST_GMM_FULL and ds.k == c["tgt_loc"].shape[0] and ds.aux and ds.scale
        return
    assert tgt is not None and E.dist_desc(tgt[0], "cpu", keep).kind != L.DIST_NONE


def test_coef_table_matches_reference_scalars():
    """Per-step coefficients are the reference's own scalar formulas (checked against the oracle, which is
    pinned to the reference bit for bit in test_oracle_golden.test_sde_scalars)."""
    from oracle import sde_oracle as orc
    from sde_sampler_lrds_amd.eq.sdes import VP, PinnedBM
    ts = torch.linspace(0.0, 1.0, 17)
    tab = E.coef_table("ei", ts, VP(0.1, 10.0, 1.3, terminal_t=1.0), with_ref=True)
    o = orc.VP(0.1, 10.0, 1.3, 1.0)
    for k in range(16):
        s, t = ts[k], ts[k + 1]
        lam = o.lam(s, t)
        assert tab[k, 0] == ts[-1] - s
        assert tab[k, 1] == torch.sqrt(1.0 + lam)
        assert tab[k, 2] == 2.0 * o.sig ** 2 * (torch.sqrt(1.0 + lam) - 1.0)
        assert tab[k, 3] == o.sig * torch.sqrt(lam)
        assert tab[k, 4] == 0.5 * o.omega(s, t) and tab[k, 5] == torch.sqrt(o.omega(s, t))
        tau = ts[-1] - s
        assert tab[k, 9] == o.s(tau) and tab[k, 10] == o.s(tau) ** 2 * o.sigma_sq(tau) and tab[k, 11] == o.s(tau) ** 2
    tsp = torch.linspace(0.05, 4.9, 9)
    tabp = E.coef_table("ddpm", tsp, PinnedBM(diff_coeff=0.4472135954999579, terminal_t=5.0))
    op = orc.PinnedBM(0.4472135954999579, 5.0)
    for k in range(8):
        assert tabp[k, 4] == 0.5 * op.omega_ddpm(tsp[k], tsp[k + 1])
        assert tabp[k, 1] == tsp[k + 1] / tsp[k]


def test_eubo_coef_tables_follow_the_noising_loops():
    """compute_eubo coefficient rows (iteration order, times running backwards) against the oracle's scalars
    (losses/oc.py:325-358 EM, :539-564 EI)."""
    from oracle import sde_oracle as orc
    from sde_sampler_lrds_amd.eq.sdes import VP
    ts = torch.linspace(0.0, 1.0, 13)
    N, T = 12, ts[-1]
    o = orc.VP(0.1, 10.0, 1.0, 1.0)
    ei = E.coef_table("eubo_ei", ts, VP(0.1, 10.0, 1.0, terminal_t=1.0), with_ref=True)
    em = E.coef_table("eubo_em", ts, VP(0.1, 10.0, 1.0, terminal_t=1.0), with_ref=True, rescale=True)
    for k in range(N):
        s, t = ts[N - 1 - k], ts[N - k]
        mean_f, var_f = o.transition_params(T - t, T - s)
        assert ei[k, 0] == T - s and ei[k, 1] == mean_f and ei[k, 3] == var_f.sqrt() and ei[k, 2] == 1.0 and ei[k, 6] == 0.0
        assert ei[k, 4] == o.omega(s, t) and ei[k, 5] == torch.sqrt(o.omega(s, t))
        g, dt = o.diff(T - s), t - s
        assert em[k, 1] == mean_f and em[k, 3] == var_f.sqrt() and em[k, 2] == 1.0 / g
        assert em[k, 4] == dt * g ** 2 and em[k, 5] == var_f.sqrt() / mean_f
        assert em[k, 6] == 1.0 / mean_f - 1.0 + o.drift_coeff(T - s) * dt
        assert ei[k, 9] == o.s(T - s)


def test_training_calls_have_no_cpu_path_either():
    """KL and log-variance training both start with the HIP step loop: on CPU tensors they fail loudly, like simulate()."""
    c = gc.load("rds_ei_gmm_d8_k4")
    b = bc.build(c, "cpu")
    for method in ("kl", "lv"):
        b["loss"].method = method
        with pytest.raises(RuntimeError, match="MI355X"):
            b["loss"](b["ts"], b["x0"], *b["args"])


def test_no_cpu_execution_path():
    c = gc.load("rds_ei_gmm_d8_k4")
    b = bc.build(c, "cpu")
    with pytest.raises(RuntimeError, match="MI355X"):
        b["loss"].simulate(b["ts"], b["x0"], *b["args"])
    with pytest.raises(RuntimeError, match="MI355X"):
        b["loss"](b["ts"], b["x0"], *b["args"])  # training direction (KL: the fixture's method): the same step loop, the same refusal


def test_loss_surface_matches_reference_names():
    from sde_sampler_lrds_amd.losses import oc
    for cls in ["BaseOCLoss", "EMReferenceSDELoss", "EIReferenceSDELoss", "DDPMLikeReferenceSDELoss",
                "ControlledLangevinSDELoss", "DiscreteTimeReversalLossEI", "TimeReversalLoss", "ExponentialIntegratorSDELoss"]:
        assert hasattr(oc, cls)
    with pytest.raises(ValueError, match="Unknown loss method"):
        oc.BaseOCLoss(None, None, method="nope")
    with pytest.raises(ValueError, match="single trajectory"):
        oc.BaseOCLoss(None, None, method="lv_traj", traj_per_sample=1)
    l = oc.BaseOCLoss(None, None)
    l.load_state_dict({"n_filtered": 3})
    assert l.state_dict() == {"n_filtered": 3}


def test_mirror_modules_match_oracle_on_cpu():
    """The host-side torch forward of the mirrors (API surface, not the simulate path) equals the oracle."""
    from oracle import sde_oracle as orc
    c = gc.load("pis_em_phi4_d100")
    b = bc.build(c, "cpu")
    ctrl = b["loss"].generative_ctrl
    tgt = orc.PhiFour(c.meta["a"], c.meta["b"], c.meta["d"], c.meta["beta"])
    o = orc.Ctrl(c.params("ctrl."), "score", clip_model=1e4, target_score=tgt.score, clip_score=1e4, scale_score=1.0)
    t = c["ts"][-1] - c["ts"][c.meta["N"] // 2]
    with torch.no_grad():
        assert gc.rel_err(ctrl(t, c["x_mid"]), c["u_mid"]) < 1e-6
    assert gc.rel_err(o(t, c["x_mid"]), c["u_mid"]) < 1e-6


def test_fit_gmm_gives_a_diagonal_reference():
    """experiments/benchmark_utils.py:336-361 mirror: the learned-reference preparation step of LRDS."""
    from sde_sampler_lrds_amd.experiments.benchmark_utils import fit_gmm
    g = torch.Generator().manual_seed(0)
    data = torch.cat([torch.randn(400, 3, generator=g) + 3.0, torch.randn(400, 3, generator=g) - 3.0])
    w, m, v = fit_gmm(2, data)
    assert w.shape == (2,) and m.shape == (2, 3) and v.shape == (2, 3)
    assert abs(float(w.sum()) - 1.0) < 1e-5 and float(m.abs().mean()) > 2.0


def test_fit_gmm_full_covariances():
    from sde_sampler_lrds_amd.experiments.benchmark_utils import fit_gmm
    g = torch.Generator().manual_seed(0)
    L_ = torch.tensor([[1.0, 0.0, 0.0], [0.6, 0.8, 0.0], [0.0, 0.3, 0.5]])
    data = torch.cat([torch.randn(3000, 3, generator=g) @ L_.T + 3.0, torch.randn(3000, 3, generator=g) * 0.5 - 3.0])
    w, m, cov = fit_gmm(2, data, em_type="full")
    assert cov.shape == (2, 3, 3) and torch.allclose(w.sum(), torch.tensor(1.0), atol=1e-5)
    k = int(m[:, 0].argmax())
    assert torch.allclose(cov[k], L_ @ L_.T, atol=0.12)
    with pytest.raises(NotImplementedError):
        fit_gmm(2, data, em_type="tied")


def test_interpolate_states_equals_the_reference_bookkeeping():
    """eq.integrator.interpolate_states (all steps at once) == the step-by-step emission of eq/integrator.py:110-128,
    on grids where ts hits grid points, falls between them, and several ts fall inside one step."""
    from oracle import sde_oracle as orc
    from sde_sampler_lrds_amd.eq.integrator import interpolate, interpolate_states
    g = torch.Generator().manual_seed(3)
    for n_grid, n_out in ((12, 5), (9, 30), (17, 17)):
        grid = torch.cat([torch.zeros(1), torch.sort(torch.rand(n_grid - 1, generator=g)).values, torch.ones(1)])
        ts = torch.linspace(0.0, 1.0, n_out)
        x0 = torch.randn(6, 3, generator=g)
        incs = torch.randn(n_grid, 6, 3, generator=g)
        drift, diff = (lambda s, x: -0.7 * x), (lambda s: torch.tensor(0.9))
        want = orc.euler_integrate(drift, diff, ts, x0, grid, lambda k, s, t, x: incs[k] * torch.sqrt(t - s))
        states = [x0]
        for k in range(n_grid):
            s_, t_ = grid[k], grid[k + 1]
            states.append(states[-1] + drift(s_, states[-1]) * (t_ - s_) + diff(s_) * (incs[k] * torch.sqrt(t_ - s_)))
        got = interpolate_states(ts, grid, torch.stack(states))
        assert torch.equal(got, want)
        # on the grid itself the states come back exactly
        assert torch.equal(interpolate_states(grid, grid, torch.stack(states)), torch.stack(states))
    # the single-step helper keeps the reference's signature and result
    xs, xt = torch.zeros(2, 2), torch.ones(2, 2)
    out = interpolate(torch.tensor([0.25, 0.5, 0.9]), torch.tensor(0.0), torch.tensor(0.5), xs, xt)
    assert out.shape == (2, 2, 2) and torch.allclose(out[0], torch.full((2, 2), 0.5))


def test_euler_states_refuses_cpu_and_unknown_sdes():
    from sde_sampler_lrds_amd import engine as E
    from sde_sampler_lrds_amd.eq.integrator import EulerIntegrator
    from sde_sampler_lrds_amd.eq.sdes import VP
    with pytest.raises(RuntimeError, match="MI355X"):
        EulerIntegrator().integrate(VP(0.1, 10.0, 1.0), torch.linspace(0, 1, 5), torch.zeros(4, 3), timesteps=torch.linspace(0, 1, 5))
    assert E.L.CTRL_NONE == 3


@pytest.mark.parametrize("name", __import__("tests.golden_cases", fromlist=["x"]).SAMPLER_CASES)
def test_annealed_samplers_match_reference_fixture(name):
    """additions/ebm_mle.py smc_sampler / re_sampler (and the MALA / ULA moves under them) against the reference's own
    output: same inputs, same seed of torch's global generator, same random-number consumption order -> same chains."""
    from sde_sampler_lrds_amd.additions import ebm_mle
    from tests import golden_cases as gc
    c = gc.load(name)
    m = c.meta
    x_init, times, steps = gc.sampler_inputs(m)
    kw = dict(m["kw"], **gc.sampler_precond(m))
    if m.get("pdds"):
        from sde_sampler_lrds_amd.eq.sdes import VP
        kw.update(use_pdds_weights=True, sde=VP(0.1, 10.0, 1.0, terminal_t=1.0))
    fn = gc.sampler_fn(m)
    torch.manual_seed(m["seed"])
    if m["sampler"] == "smc":
        samples, steps_out, diags = ebm_mle.smc_sampler(x_init, times, fn, m["n_warm"], m["n_steps"], steps.clone(), **kw)
    else:
        samples, steps_out, diags = ebm_mle.re_sampler(x_init, times, fn, kw.pop("swap_frequency"), m["n_warm"], m["n_steps"], steps.clone(), **kw)
    assert samples.shape == c["samples"].shape
    assert float((samples - c["samples"]).abs().max()) < 1e-5
    assert float((steps_out.reshape(m["n_levels"], m["B"], 1) - c["steps_out"]).abs().max()) < 1e-7
    for k, v in diags.items():
        assert float((torch.as_tensor(v).float() - c["diag_" + k]).abs().max()) < 1e-5, k
    if name == "smc_tempered_d3":
        assert float(c["diag_ess"].min()) < 0.7  # the fixture did go through the resampling branch


def test_sampler_refusals_and_pairings():
    from sde_sampler_lrds_amd.additions import ebm_mle
    a, b = ebm_mle.make_re_pairings(6)
    assert a.tolist() == [[0, 1], [2, 3], [4, 5]] and b.tolist() == [[1, 2], [3, 4]]
    x, t, st = torch.zeros(4, 2), torch.zeros(3, 4, 1), torch.ones(3, 4, 1)
    with pytest.raises(ValueError):
        ebm_mle.smc_sampler(x, t, None, 1, 1, st, use_pdds_weights=True, sde=None)
    with pytest.raises(ValueError):
        ebm_mle.smc_sampler(torch.zeros(3, 4, 2), t, None, 1, 1, st, per_noise_init=True, reweight_threshold=1.0)


@pytest.mark.skipif(not os.path.isdir("/root/reference/sde_sampler"), reason="the reference checkout only exists in the build container")
def test_reference_objects_compile_to_descriptors():
    """INTEGRATION.md, route 1: the drop-in loss classes are handed the REFERENCE's own objects (its VP, ClippedCtrl / ScoreCtrl around
    its FourierMLP, its ManyModes / IsotropicGauss, a bound ``reference_ctrl`` with ``reference_distr_utils``).  Descriptor
    compilation is host-side, so it is checked here: same coefficient table as with this package's classes, every object
    recognised."""
    import sys
    import types
    for n in ("wandb", "torchquad", "torchsde"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.path.insert(0, "/root/reference")
    try:
        from sde_sampler.distr import gauss as r_gauss
        from sde_sampler.eq import sdes as r_sdes
        from sde_sampler.models import mlp as r_mlp
        from sde_sampler.models import reparam as r_rep
    finally:
        sys.path.remove("/root/reference")
    from sde_sampler_lrds_amd.eq.sdes import VP
    from sde_sampler_lrds_amd.losses import oc
    d, K = 16, 3
    r_sde = r_sdes.VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, scale_diff_coeff=1.0, terminal_t=1.0)
    target = r_gauss.ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, n_reference_samples=10)
    prior = r_gauss.IsotropicGauss(dim=d, scale=1.0)
    net = r_mlp.FourierMLP(dim=d, activation=torch.nn.GELU(), num_layers=4, channels=64)
    sm = r_mlp.TimeEmbed(dim_out=1, activation=torch.nn.GELU(), num_layers=4, channels=64)

    class FakeRDS:  # what solver/oc.py:563-592 leaves on the solver object
        reference_distr_utils = dict(means_init=target.loc.clone(), variances_init=0.5 * torch.ones(K, d), weights_init=torch.ones(K))

        def reference_ctrl(self, t, x):
            return r_sde.marginal_gmm_score(t=t, x=x, **self.reference_distr_utils)

    rds = FakeRDS()
    ts = torch.linspace(0.0, 1.0, 9)
    keep = []
    for ctrl in (r_rep.ClippedCtrl(base_model=net, clip_model=1e4),
                 r_rep.ScoreCtrl(base_model=net, score_model=sm, target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4),
                 r_rep.LerpCtrl(base_model=net, score_model=sm, target_score=target.score, detach_score=False, clip_score=1e4, clip_model=1e4,
                                sde=r_sde, prior_score=prior.score)):
        nd = E.net_desc(ctrl, "cpu", keep)
        assert nd.ctrl_kind == {"ClippedCtrl": L.CTRL_CLIPPED, "ScoreCtrl": L.CTRL_SCORE, "LerpCtrl": L.CTRL_LERP}[type(ctrl).__name__]
        tgt, lerp_prior = E.ctrl_target(ctrl)
        assert (tgt is None) == (nd.ctrl_kind == L.CTRL_CLIPPED) and (lerp_prior is not None) == (nd.ctrl_kind == L.CTRL_LERP)
    ctrl = r_rep.ClippedCtrl(base_model=net, clip_model=1e4)
    loss = oc.EIReferenceSDELoss(ctrl, ctrl, sde=r_sde, method="lv", reference_ctrl=rds.reference_ctrl)
    kind, utils = E.resolve_reference(loss.reference_ctrl)
    assert kind == "gmm" and E.ref_desc(kind, utils, "cpu", keep).kind == L.REF_GMM_DIAG
    mine = oc.EIReferenceSDELoss(ctrl, ctrl, sde=VP(0.1, 10.0, 1.0, terminal_t=1.0), method="lv", reference_ctrl=rds.reference_ctrl)
    assert torch.equal(loss._coef(ts, "cpu", with_ref=True), mine._coef(ts, "cpu", with_ref=True))
    res = E.resolve_logp(target.unnorm_log_prob)
    assert res is not None and E.dist_desc(res[0], "cpu", keep).kind == L.DIST_GMM_DIAG
    assert E.dist_desc(prior, "cpu", keep).kind == L.DIST_ISO_GAUSS


def test_coef_cache_is_keyed_by_content_not_address():
    """a fresh ``ts`` per call with other values must never hit a stale table, whatever address it lands on; the same values
    in a new tensor reuse the table; a changed SDE buffer rebuilds it."""
    c = gc.load("rds_ei_gmm_d8_k4")
    loss = bc.build(c, "cpu")["loss"]
    ts1 = torch.linspace(0.0, 1.0, 9)
    t1 = loss._coef(ts1, "cpu", with_ref=True)
    assert loss._coef(ts1, "cpu", with_ref=True) is t1                      # same object: identity hit
    assert loss._coef(ts1.clone(), "cpu", with_ref=True) is t1              # same values, new tensor: content hit
    ts2 = torch.linspace(0.0, 1.0, 9) ** 2                                  # same length, other grid
    t2 = loss._coef(ts2, "cpu", with_ref=True)
    assert not torch.equal(t1, t2)
    del ts2
    ts3 = torch.linspace(0.0, 0.5, 9)                                       # may land on ts2's address
    assert not torch.equal(loss._coef(ts3, "cpu", with_ref=True), t2)
    before = loss._coef(ts1, "cpu", with_ref=True).clone()
    with torch.no_grad():
        loss.sde.diff_coeff_sq_max.mul_(2.0)                                # e.g. load_state_dict of another VP
    after = loss._coef(ts1, "cpu", with_ref=True)
    assert not torch.equal(before, after)
