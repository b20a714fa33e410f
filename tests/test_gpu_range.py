"""Dynamic range of the split-f16 drift net.  The GEMMs run as three f16 products per fp32 product
(sim_device.hpp): f16 has fp32's mantissa budget only inside its NORMAL range, and overflows at 65 504.  Two mechanisms give the
path fp32's range, both held to the oracle here:

* weights: each matrix is stored times a power of two that brings its largest entry to [1, 2) when that entry is below 2^-10 or at /
  above 2^14 (k_weight_scales), the layer's output is multiplied back -- the reference's OWN default initialisation puts the last
  layer at |w| <= 1.25e-7 (models/utils.py:7-22 kaiming_uniform_zeros_), f16-subnormal: before this the control at init carried a
  1.3e-4 relative error, now fp32 round-off;
* states / activations: an operand beyond 65 504 is detected after the fact (one compare per tile-step) and the step's net is
  re-evaluated with per-particle power-of-two scaling (mlp_hidden_safe): |x0| = 1e5 gives the oracle's finite result, not NaN.
"""
import pytest
import torch

from oracle import baseline_oracles as bo
from oracle import sde_oracle as orc
from sde_sampler_lrds_amd import engine as E
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
from sde_sampler_lrds_amd.models import utils as mutils
from sde_sampler_lrds_amd.models.mlp import FourierMLP
from sde_sampler_lrds_amd.models.reparam import ClippedCtrl


def _fp64_forward(ctrl, t, x):
    """The control in fp64 (the yardstick both fp32 implementations are measured against)."""
    import copy
    c64 = copy.deepcopy(ctrl).double().cpu()
    with torch.no_grad():
        return c64(torch.tensor(t, dtype=torch.float64), x.double().cpu())


def _oracle_forward(ctrl, t, x):
    sd = {k: v.detach().cpu() for k, v in ctrl.state_dict().items()}
    with torch.no_grad():
        return orc.Ctrl(sd, "clipped", clip_model=ctrl.clip_model)(torch.tensor(t), x.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("d", [128, 40, 2])
def test_control_at_the_reference_default_initialisation(gpu, d):
    """make_model(...) starts every sampler from kaiming_uniform_zeros_ / init_bias_uniform_zeros on the last layer: |w| ~ 1e-7."""
    torch.manual_seed(d)
    net = FourierMLP(dim=d, activation=torch.nn.GELU(), num_layers=4, channels=64, last_bias_init=mutils.init_bias_uniform_zeros,
                     last_weight_init=mutils.kaiming_uniform_zeros_)
    assert float(net.out_layer.weight.abs().max()) < 2e-7  # f16-subnormal
    ctrl = ClippedCtrl(base_model=net, clip_model=1e4).to(gpu)
    x = 2.0 * torch.randn(333, d, device=gpu)
    got = E.ctrl_forward(ctrl, 0.37, x).cpu().double()
    want64, want32 = _fp64_forward(ctrl, 0.37, x), _oracle_forward(ctrl, 0.37, x).double()
    scale = float(want64.abs().max())
    e_hip, e_orc = float((got - want64).abs().max()) / scale, float((want32 - want64).abs().max()) / scale
    print(f"default init d={d}: |u| max {scale:.2e}; HIP vs fp64 {e_hip:.2e}, torch fp32 vs fp64 {e_orc:.2e}, HIP vs oracle {float((got - want32).abs().max()) / scale:.2e}")
    assert scale < 1e-4 and e_hip < 1e-6 and float((got - want32).abs().max()) / scale < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("which,factor", [("input_embed", 1e-6), ("hidden_layer.0", 3e-5), ("hidden_layer.1", 1e-7), ("out_layer", 1e-9),
                                          ("out_layer", 3e5), ("hidden_layer.0", 1e5), ("input_embed", 4e4)])
def test_every_layer_keeps_fp32_accuracy_out_of_the_f16_range(gpu, which, factor):
    """One layer at a time far below / above the range in which the un-scaled split is accurate.  Error of the HIP control against fp64,
    relative to the largest output, must stay at the fp32 evaluation's own (torch CPU) -- within 4x."""
    torch.manual_seed(7)
    d = 72
    net = cfgs._net(d)
    mod = net
    for part in which.split("."):
        mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
    with torch.no_grad():
        mod.weight.mul_(factor)
        mod.bias.mul_(factor)
    ctrl = ClippedCtrl(base_model=net, clip_model=None).to(gpu)
    x = torch.randn(200, d, device=gpu)
    got = E.ctrl_forward(ctrl, 0.6, x).cpu().double()
    want64, want32 = _fp64_forward(ctrl, 0.6, x), _oracle_forward(ctrl, 0.6, x).double()
    scale = float(want64.abs().max())
    e_hip, e_orc = float((got - want64).abs().max()) / scale, float((want32 - want64).abs().max()) / scale
    print(f"{which} x {factor:g}: |u| max {scale:.2e}; HIP vs fp64 {e_hip:.2e}, torch fp32 vs fp64 {e_orc:.2e}")
    assert bool(torch.isfinite(got).all()) and e_hip < max(4 * e_orc, 5e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,mag", [(128, 4, 1e5), (100, 3, 3e6), (40, 4, 7e4)])
def test_states_beyond_the_f16_range_match_the_oracle(gpu, d, K, mag):
    """|x0| ~ 1e5 .. 3e6: finite in the reference (fp32 GEMMs), NaN on the un-guarded f16 split.  cfg-2 shaped sampler, injected noise,
    against the oracle; healthy particles in the same tiles are unaffected (same bits as a run without the large ones)."""
    B, N = 96, 12
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N, d=d, K=K, seed=11 + d)
    g = torch.Generator().manual_seed(5)
    big = x0.clone()
    rows = torch.tensor([3, 17, 40, 41, 95])
    big[rows] = (mag * torch.randn(len(rows), d, generator=g)).to(gpu)
    z = torch.randn(N, B, d, generator=g)
    x, rnd, _ = loss.simulate(ts, big, *args, noise=z.to(gpu), **kw)
    run = bo.runner("rds_gmm", info, ts)
    ox, ornd, scale = run(big.cpu(), orc.InjectedNoise(z))
    assert bool(torch.isfinite(ox).all()), "the reference arithmetic itself is finite here"
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(rnd[torch.isfinite(ornd)]).all())
    # A state of 1e5 is pulled back to O(1) within a few steps by the reference drift: the fp32 round-off it carried while it was large
    # (1e5 x 6e-8 per operation) is then a RELATIVE error of 1e-3 .. 1e-2 of the end point -- in the reference's own arithmetic.  Per
    # particle: what a one-ulp relative change of x0 does to the oracle's x_N; the tolerance is max(1e-5, 10 x that), as everywhere.
    rel = lambda a, b: ((a - b).abs() / b.abs().clamp(min=1.0)).amax(dim=1)  # noqa: E731
    fin = torch.isfinite(ornd).flatten()
    rscale = torch.maximum(ornd.abs().flatten(), torch.tensor(scale)).clamp(min=1.0)
    sens, rsens = torch.zeros(B), torch.zeros(B)
    probes = [(big.cpu() * (1.0 + 1.2e-7), orc.InjectedNoise(z))] + [(big.cpu(), bo.PerturbedNoise(orc.InjectedNoise(z), salt=i)) for i in range(3)]
    for xp, nz in probes:  # one ulp of x0, and three +-1.2e-6 sign patterns on the normals (the probes of the other parity tests)
        px, prnd, _ = run(xp, nz)
        sens = torch.maximum(sens, rel(px, ox))
        rsens = torch.maximum(rsens, ((prnd.flatten() - ornd.flatten()).abs() / rscale).nan_to_num(0.0))
    healthy = torch.ones(B, dtype=torch.bool)
    healthy[rows] = False
    # large particles against the large set's sensitivity, the healthy ones against the healthy set's (a 12-step grid over [0, 1] takes
    # giant steps -- x gain 12 per step -- so a particle that did not amplify under these four probes can under another rounding pattern)
    tol = torch.empty(B)
    tol[~healthy] = max(1e-5, 10 * float(sens[~healthy].max()))
    tol[healthy] = max(1e-5, 10 * float(sens[healthy].max()))
    ex = rel(x.cpu(), ox)
    er = ((rnd.cpu().flatten() - ornd.flatten()).abs() / rscale)
    print(f"|x0| ~ {mag:g}, d={d}: large particles x_N rel err {float(ex[rows].max()):.2e} (own one-ulp sensitivity {float(sens[rows].max()):.2e}), "
          f"healthy particles {float(ex[healthy].max()):.2e}; rnd {float(er[fin].max()):.2e} (sensitivity {float(rsens[fin].max()):.2e}); "
          f"{int(fin.sum())} / {B} finite log-weights in the oracle")
    assert bool((ex <= tol).all()), (ex / tol).max()
    assert bool((er[fin] <= torch.clamp(10 * rsens[fin], min=1e-5)).all())
    # healthy particles sharing a tile with a large one went through the range-safe twin too: the bulk at round-off like everywhere
    assert float(ex[healthy].median()) < 2e-6


@pytest.mark.gpu
def test_large_hidden_activations_take_the_safe_path(gpu):
    """Moderate states, but first-layer weights that push the hidden activations past 65 504."""
    torch.manual_seed(3)
    d = 128
    net = cfgs._net(d)
    with torch.no_grad():
        net.input_embed.weight.mul_(3e3)  # |W_in| ~ 3e2 (inside the un-scaled weight range): activations ~ 3e3 * |x| * sqrt(d)
    ctrl = ClippedCtrl(base_model=net, clip_model=None).to(gpu)
    x = 30.0 * torch.randn(64, d, device=gpu)
    got = E.ctrl_forward(ctrl, 0.2, x).cpu().double()
    want64 = _fp64_forward(ctrl, 0.2, x)
    want32 = _oracle_forward(ctrl, 0.2, x).double()
    scale = float(want64.abs().max())
    e_hip, e_orc = float((got - want64).abs().max()) / scale, float((want32 - want64).abs().max()) / scale
    print(f"hidden activations up to {float(torch.nn.functional.linear(x, net.input_embed.weight.to(gpu)).abs().max()):.1e}: HIP vs fp64 {e_hip:.2e}, torch fp32 {e_orc:.2e}")
    assert bool(torch.isfinite(got).all()) and e_hip < max(4 * e_orc, 5e-7)


@pytest.mark.gpu
def test_non_finite_inputs_stay_non_finite(gpu):
    """inf / NaN states are the reference's business (BaseOCLoss.filter counts them, losses/oc.py:67-81): the guard must not launder them."""
    B, N, d = 48, 6, 128
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N, d=d, K=4, seed=2)
    bad = x0.clone()
    bad[5, 7] = float("inf")
    bad[20, 100] = float("nan")
    x, rnd, _ = loss.simulate(ts, bad, *args, **kw)
    assert not bool(torch.isfinite(x[5]).all()) and not bool(torch.isfinite(x[20]).all())
    assert not bool(torch.isfinite(rnd[5]).all()) and not bool(torch.isfinite(rnd[20]).all())
    ok = torch.ones(B, dtype=torch.bool)
    ok[[5, 20]] = False
    assert bool(torch.isfinite(x[ok.to(gpu)]).all()) and bool(torch.isfinite(rnd[ok.to(gpu)]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rds_ei_gmm_d128_k4", "pis_em_phi4_d100", "cmcd_logreg_d61", "rds_em_gmm_d16", "dds_two_modes_d2", "rds_ei_gmm_d128_k16"])
def test_step_loop_at_the_default_initialisation_magnitude(gpu, name):
    """Every freshly built model starts with its last layer at |w| <= 1.25e-7 (models/utils.py:7-22): stored scaled up, un-scaled where the
    output tiles are produced (the plain path of the step loop, not the twin).  The fixture's net with its output layer x 1e-6, injected
    noise: x_N and the log-weights against the ORACLE on the same net."""
    from tests import build_cases as bc
    from tests import golden_cases as gc
    from tests.test_gpu_parity import TOL, replay_noise, rnd_scale, sensitivity
    c = gc.load(name)
    for key in ("out_layer.weight", "out_layer.bias"):
        c.a["ctrl.base_model." + key] = c.a["ctrl.base_model." + key] * 1e-6
    noise = replay_noise(c)
    x_o, r_o = gc.run_oracle(c)
    b = bc.build(c, gpu)
    x, rnd, _ = b["loss"].simulate(b["ts"], b["x0"], *b["args"], noise=noise.to(gpu), **b["kwargs"])
    torch.cuda.synchronize()
    c.a["out_x"], c.a["rnd"] = x_o, r_o
    ex = gc.rel_err(x.cpu(), x_o)
    er = float(((rnd.cpu().double() - r_o.double()).abs() / rnd_scale(c).double()).max())
    tol = max(TOL, 10 * sensitivity(name))
    print(f"{name} with the output layer x 1e-6: x_N {ex:.2e}, rnd {er:.2e} against the oracle (tolerance {tol:.1e})")
    assert bool(torch.isfinite(x).all()) and ex < tol and er < tol


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rds_ei_gmm_d128_k16", "rds_ei_gmm_fullcov_d128_k4", "pis_gmm_full_d128_k3", "pis_logreg_d61", "rds_ei_gmm_d128_k4",
                                  "pis_em_phi4_d100"])
def test_scaled_weights_in_the_step_loops_without_a_twin(gpu, name):
    """(the last two names: standard kernels, where scaled input / hidden layers go through the twin instead)
    The matrix-pipe / full-covariance mixture kernels and the in-loop logistic-regression control carry no range-safe twin: scaled weight
    matrices are un-scaled layer by layer there (mlp_hidden_scaled).  The fixture's net with its input layer x 1e-5 and its second hidden
    layer x 2e-4 (entries below 2^-10: both stored scaled up), the output weights x 5e3 so that the control keeps its size (activations
    stay inside f16's range: these kernels have no overflow twin): injected noise, x_N and the log-weights against the ORACLE run on
    the same modified net."""
    from tests import build_cases as bc
    from tests import golden_cases as gc
    from tests.test_gpu_parity import TOL, replay_noise, rnd_scale, sensitivity
    c = gc.load(name)
    for key, f in (("input_embed.weight", 1e-5), ("input_embed.bias", 1e-5), ("hidden_layer.1.weight", 2e-4), ("hidden_layer.1.bias", 2e-4),
                   ("out_layer.weight", 5e3)):
        c.a["ctrl.base_model." + key] = c.a["ctrl.base_model." + key] * f
    noise = replay_noise(c)
    x_o, r_o = gc.run_oracle(c)
    b = bc.build(c, gpu)
    x, rnd, _ = b["loss"].simulate(b["ts"], b["x0"], *b["args"], noise=noise.to(gpu), **b["kwargs"])
    torch.cuda.synchronize()
    c.a["out_x"], c.a["rnd"] = x_o, r_o  # (rnd_scale reads the case's own outputs)
    ex = gc.rel_err(x.cpu(), x_o)
    er = float(((rnd.cpu().double() - r_o.double()).abs() / rnd_scale(c).double()).max())
    tol = max(TOL, 10 * sensitivity(name))
    print(f"{name} with scaled layers: x_N {ex:.2e}, rnd {er:.2e} against the oracle (tolerance {tol:.1e})")
    assert bool(torch.isfinite(x).all()) and ex < tol and er < tol
