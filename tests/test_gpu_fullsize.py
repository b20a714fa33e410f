"""BASELINE.json's full sizes on the GPU, checked through size-independent properties:

* sharding independence: the batch simulated as two shards (Philox keyed by global particle index) equals the
  single-shard run BIT FOR BIT, and a rerun is bit-identical (no atomics, no order dependence);
* a block of 32 particles from the middle of the batch equals the CPU oracle run on just those particles with the
  same counter-based noise (Philox mode; tolerance as tests/test_gpu_parity.py: hardware sin/cos/log2 vs libm);
* estimator identities: log Z / ELBO / ESS from the HIP reduction equal fp64 torch on the same log-weights, the
  softmax weights sum to one.
"""
import math

import pytest
import torch

from oracle import baseline_oracles as bo
from oracle import sde_oracle as orc
from sde_sampler_lrds_amd import engine as E
from sde_sampler_lrds_amd import parallel
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
from tests import golden_cases as gc

P0, PB = 40000, 32  # the block of particles cross-checked against the oracle


def _sd(mod):
    return {k: v.detach().cpu() for k, v in mod.state_dict().items()}


def _run_full_and_shards(loss, ts, x0, args, kw):
    B = x0.shape[0]
    cut = (B // 2) + 16 * 3 + 5  # a cut that is not a multiple of the 16-particle tile
    loss.particle0 = 0
    full = loss.simulate(ts, x0, *args, **kw)
    for _ in range(3):  # reruns at full occupancy: every wave, every step must reproduce (this caught a hardware-level
        again = loss.simulate(ts, x0, *args, **kw)  # corruption that no small-batch test could see)
        assert torch.equal(full[0], again[0]) and torch.equal(full[1], again[1]), "rerun differs"
    a = loss.simulate(ts, x0[:cut], *args, **kw)
    loss.particle0 = cut
    z = loss.simulate(ts, x0[cut:].contiguous(), *args, **kw)
    loss.particle0 = 0
    assert torch.equal(torch.cat([a[0], z[0]]), full[0]), "x_N depends on the sharding"
    assert torch.equal(torch.cat([a[1], z[1]]), full[1]), "log-weights depend on the sharding"
    assert bool(torch.isfinite(full[0]).all()) and bool(torch.isfinite(full[1]).all())
    return full


def _plain_equals_trajectory_twin(loss, ts, x0, args, kw, full, name, nb=8192):
    """The benchmarked instantiation (PAR = 0: no injected-noise / trajectory paths in the step loop) against its parity-mode twin
    (PAR = 1, selected by return_traj or noise_in -- the instantiation every injected-noise fixture test runs): same Philox normals,
    same arithmetic, so x_N and the log-weights must agree BIT FOR BIT.  This carries the strict injected-noise evidence of
    tests/test_gpu_parity.py over to the kernel bench.py times.  Checked on the first ``nb`` particles of the full-size batch (the
    trajectory of all of them would be 8.6 GB at cfg 2), which by the sharding test equal rows [0, nb) of the full run."""
    loss.particle0 = 0
    x, rnd, xs = loss.simulate(ts, x0[:nb], *args, return_traj=True, **kw)
    assert xs.shape == (ts.numel(), nb, x0.shape[1]) and torch.equal(xs[-1], x)
    assert torch.equal(x, full[0][:nb]), f"{name}: PAR=1 twin x_N differs from the plain kernel's"
    assert torch.equal(rnd, full[1][:nb]), f"{name}: PAR=1 twin log-weights differ from the plain kernel's"
    # ... and the twin with INJECTED noise equal to the Philox stream reproduces the same bits (noise_in path = in-register path)
    nz = E.philox_noise(int(loss.seed), ts.numel() - 1, 2048, x0.shape[1], 0, x0.device)
    xi, rndi, _ = loss.simulate(ts, x0[:2048], *args, noise=nz, **kw)
    assert torch.equal(xi, full[0][:2048]) and torch.equal(rndi, full[1][:2048]), f"{name}: injected Philox normals give other bits"
    print(f"{name}: plain kernel == trajectory twin == injected-noise twin, bit for bit ({nb} / 2048 particles)")


def _check_estimators(rnd):
    res = parallel.global_results(rnd)
    r = (-rnd.double().flatten()).cpu()
    logz = float(torch.logsumexp(r, 0) - math.log(r.numel()))
    w = torch.softmax(r, 0)
    assert abs(res["log_norm_const_is"] - logz) < 1e-4 * max(1.0, abs(logz))
    assert abs(res["elbo"] - float(r.mean())) < 1e-4 * max(1.0, abs(float(r.mean())))
    assert abs(res["ess"] - float(1.0 / (w * w).sum() / r.numel())) < 1e-4
    _, weights = E.logz_stats(rnd, want_weights=True)
    assert abs(float(weights.double().sum()) - 1.0) < 1e-4


def _block_vs_oracle(cfg, info, ts, x0, x, rnd, seed, name, p0=P0, pb=PB):
    """A block of the full-size run against the oracle on just those particles with the same counter-based noise ('identical seeds'
    mode).  Tolerance: 1e-5 (the north_star's bound), or 10 x what the block itself moves when every normal is perturbed by the
    kernel's Box-Muller error (1.2e-6), whichever is larger -- measured here, printed with the achieved error."""
    run = bo.runner(cfg, info, ts)
    xb = x0[p0:p0 + pb].cpu()
    base = orc.PhiloxNoise(seed, particle0=p0)
    ox, ornd, scale = run(xb, base)
    sens = 0.0
    for salt in range(2):
        px, prnd, _ = run(xb, bo.PerturbedNoise(base, salt=salt))
        sens = max(sens, gc.rel_err(px, ox), float((prnd - ornd).abs().max()) / scale)
    x_err = gc.rel_err(x[p0:p0 + pb].cpu(), ox)
    r_err = float((rnd[p0:p0 + pb].cpu().flatten() - ornd.flatten()).abs().max()) / scale
    tol = max(1e-5, 10 * sens)
    print(f"{name}: block [{p0},{p0 + pb}) vs oracle: x_N {x_err:.2e}, rnd {r_err:.2e}  (tolerance {tol:.1e}; noise sensitivity {sens:.1e})")
    assert x_err < tol and r_err < tol


@pytest.mark.gpu
def test_cfg2_rds_gmm_65536x256(gpu):
    B, N = 65536, 256
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N)
    loss.seed = 5
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw)
    _check_estimators(rnd)
    _block_vs_oracle("rds_gmm", info, ts, x0, x, rnd, 5, "cfg2")
    _plain_equals_trajectory_twin(loss, ts, x0, args, kw, (x, rnd), "cfg2")


@pytest.mark.gpu
def test_cfg3_pis_phi4_131072x512(gpu):
    B, N = 131072, 512
    loss, ts, x0, args, kw, info = cfgs.build_pis_phi4(gpu, B, N)
    loss.seed = 6
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw)
    _check_estimators(rnd)
    _block_vs_oracle("pis_phi4", info, ts, x0, x, rnd, 6, "cfg3")
    _plain_equals_trajectory_twin(loss, ts, x0, args, kw, (x, rnd), "cfg3", nb=4096)


@pytest.mark.gpu
def test_cfg4_cmcd_logreg_shard_65536x256(gpu):
    B, N = 65536, 256
    loss, ts, x0, args, kw, info = cfgs.build_cmcd_logreg(gpu, B, N)
    loss.seed = 7
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw)
    _check_estimators(rnd)
    _block_vs_oracle("cmcd_logreg", info, ts, x0, x, rnd, 7, "cfg4")
    _plain_equals_trajectory_twin(loss, ts, x0, args, kw, (x, rnd), "cfg4")


@pytest.mark.gpu
@pytest.mark.parametrize("d", [17, 33, 65, 96, 127])
def test_pis_phi4_pad_boundaries(gpu, d):
    """phi^4 lattice sizes just past a tile-count boundary (d = 17, 33, 65: the last tile holds one live feature), at a tile boundary
    (96) and one short of full (127): the pad masks are only applied on tiles that can hold pads (sim_device.hpp feat_live), and the
    lattice's neighbour coupling reads across the live / pad edge."""
    import math
    B, N = 300, 24
    loss, ts, x0, args, kw, info = cfgs.build_pis_phi4(gpu, B, N, d=d)
    loss.seed = 3
    x, rnd, _ = loss.simulate(ts, x0, *args, **kw)
    _block_vs_oracle("pis_phi4", info, ts, x0, x, rnd, 3, f"pis phi4 d={d}", p0=0, pb=64)
    assert bool(torch.isfinite(rnd).all())


# ---- larger mixtures (K > 4): the workgroup-shared, double-buffered table copy -------------------------------------------
# 1-4 LDS-DMA chunks per wave, idle DMA waves (K = 5), the exact 160 KiB LDS fit (d = 128, K = 32), tables staged in
# 2 and 4 pieces per step (K = 40, 100), dpad < 128, ragged batches that leave waves and whole rounds without a tile.
SHARED = [(128, 5, 1000), (128, 16, 4096 + 7), (128, 24, 300), (128, 32, 33000), (128, 40, 200), (128, 100, 40000), (40, 6, 2500),
          (16, 48, 70000), (64, 64, 9), (100, 70, 600)]


@pytest.mark.gpu
@pytest.mark.parametrize("d,K,B", SHARED)
def test_shared_mixture_table(gpu, d, K, B):
    N, p0, pb = 12, max(0, B // 2 - 8), min(B, 24)
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N, d=d, K=K, seed=d + K)
    loss.seed = 11
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw) if B >= 64 else loss.simulate(ts, x0, *args, **kw)
    _block_vs_oracle("rds_gmm", info, ts, x0, x, rnd, 11, f"shared table d={d} K={K} B={B}", p0=p0, pb=pb)


# ---- four-component references: the centred shared-variance form (gmm_resp_centred) and the general form next to it -------------
@pytest.mark.gpu
@pytest.mark.parametrize("d,B", [(128, 2048 + 5), (100, 777), (40, 300), (8, 64)])
@pytest.mark.parametrize("kind", ["shared", "distinct", "three_shared"])
def test_four_mode_reference_forms(gpu, d, B, kind):
    """K = 4 with ONE variance vector (varying over the features) and unequal weights runs the centred table / one-fma logits;
    distinct variances run the general K = 4 form; K = 3 with a shared vector the general shared-variance form.  Each against the
    oracle on a block, identical seeds; pad features at d = 100, 40, 8."""
    K = 3 if kind == "three_shared" else 4
    g = torch.Generator().manual_seed(100 * d + K)
    v = 0.3 + torch.rand(d, generator=g)
    var = v.repeat(K, 1) if kind != "distinct" else 0.3 + torch.rand(K, d, generator=g)
    w = torch.arange(1.0, K + 1.0)
    N = 16
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N, d=d, K=K, seed=d + K, ref_var=var, ref_weights=w)
    loss.seed = 17
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw) if B >= 64 * 2 else loss.simulate(ts, x0, *args, **kw)
    _block_vs_oracle("rds_gmm", info, ts, x0, x, rnd, 17, f"K={K} {kind} d={d}", p0=max(0, B // 2 - 12), pb=min(B, 24))
    assert bool(torch.isfinite(rnd).all())


# ---- the 12-wave (three per SIMD) instantiations without a fixture of their own at full occupancy ------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind,d", [("none", 96), ("none", 128), ("gaussian", 64), ("gmm", 48)])
def test_three_wave_kernels_at_full_occupancy(gpu, kind, d):
    """No reference + ClippedCtrl: built for 12 waves per workgroup at every tile count (sim_kernel.hpp sd_waves_of); small Gaussian /
    mixture references next to them (8 waves).  40 000 particles = every wave slot busy, ragged last round: reruns and shards
    bit-equal, and blocks of the big run -- one per wave slot -- equal a small launch of just that block."""
    from sde_sampler_lrds_amd.distr.gauss import ManyModes
    from sde_sampler_lrds_amd.eq.sdes import VP, ScaledBM
    from sde_sampler_lrds_amd.losses import oc
    from sde_sampler_lrds_amd.models.reparam import ClippedCtrl
    from sde_sampler_lrds_amd.reference import MarginalReference
    torch.manual_seed(d)
    B, N = 40000 + 7, 10
    target = ManyModes(n_modes=3, dim=d, var=0.5, seed_loc=1, n_reference_samples=10)
    ctrl = ClippedCtrl(base_model=cfgs._net(d), clip_model=1e4)
    if kind == "none":
        sde, ref = ScaledBM(diff_coeff=0.4, terminal_t=2.0), None
    else:
        sde = VP(0.1, 10.0, 1.0, terminal_t=1.0)
        ref = (MarginalReference(sde, "gaussian", x_init=0.3 * torch.randn(d), var_init=0.5 + torch.rand(d)) if kind == "gaussian" else
               MarginalReference(sde, "gmm", means_init=target.loc.clone(), variances_init=0.3 + torch.rand(3, d), weights_init=torch.tensor([1.0, 2.0, 3.0])))
    for m in (sde, target, ctrl, ref):
        if m is not None:
            m.to(gpu)
    loss = (oc.EMReferenceSDELoss if kind == "none" else oc.EIReferenceSDELoss)(ctrl, ctrl, sde=sde, method="kl", reference_ctrl=ref)
    loss.seed = 31
    ts = torch.linspace(0.0, float(sde.terminal_t), N + 1, device=gpu)
    x0 = torch.randn(B, d, device=gpu)
    refd = ref.reference_distr.to(gpu).log_prob if ref is not None else (lambda x: torch.zeros(x.shape[0], device=x.device))
    args = (target.unnorm_log_prob, refd)
    full = _run_full_and_shards(loss, ts, x0, args, {})
    ntiles, grid = (B + 15) // 16, 256
    for wave in range(12):
        tile = 11 + grid * wave
        if tile >= ntiles:
            break
        lo = 16 * tile - 8
        loss.particle0 = lo
        part = loss.simulate(ts, x0[lo:lo + 48].contiguous(), *args)
        assert torch.equal(part[0], full[0][lo:lo + 48]) and torch.equal(part[1], full[1][lo:lo + 48]), f"wave slot {wave}"
    loss.particle0 = 0


# ---- every kernel family at full occupancy --------------------------------------------------------------------------
# The golden cases replicated to 32 768+ particles (each replica draws its own Philox noise): reruns and shards must be
# bit-identical, and blocks of the big run -- chosen so that every wave slot of a workgroup is covered -- must equal a
# small launch of just that block (one wave per SIMD: the regime the fixtures pin against the reference).  Kernels built for three
# waves per SIMD (sim_kernel.hpp sd_waves_of: the phi^4 and no-reference families, small-d references) have 12 wave slots.
BIG = [("rds_ei_gmm_d128_k16", 32768), ("rds_ei_gmm_d8_k4", 65536), ("rds_ddpm_gmm_d16_snr", 65536), ("rds_em_gmm_d16", 65536),
       ("rds_ei_vp_default_d16", 65536), ("rds_ei_pbm_default_d16", 65536), ("dds_two_modes_d2", 65536), ("dds_rings_d2", 65536),
       ("dis_ei_d8", 65536), ("dis_orig_lerp_d8", 65536), ("pis_em_phi4_d100", 65536), ("cmcd_logreg_d61", 32768), ("cmcd_gmm_iso_d16", 65536), ("cmcd_gmm_diag_d40", 32768), ("cmcd_phi4_d100", 32768), ("pis_logreg_d61", 32768), ("dds_logreg_d61", 32768),
       # full-covariance mixtures: as the reference (staged precision images, workgroup barriers per piece) and as the target of a score control
       ("rds_ei_gmm_fullcov_d128_k4", 32768), ("rds_em_gmm_fullcov_d40_k3", 32768), ("pis_gmm_full_d128_k3", 32768), ("pis_two_modes_full_d20", 65536),
       ("dds_two_modes_full_d5", 65536)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,B", BIG)
def test_every_kernel_family_at_full_occupancy(gpu, name, B):
    from tests import build_cases as bc
    c = gc.load(name)
    b = bc.build(c, gpu)
    loss, ts, args, kw = b["loss"], b["ts"], b["args"], b["kwargs"]
    x0 = b["x0"].repeat((B + b["x0"].shape[0] - 1) // b["x0"].shape[0], 1)[:B].contiguous()
    loss.particle0 = 0
    full = loss.simulate(ts, x0, *args, **kw)
    for _ in range(2):
        again = loss.simulate(ts, x0, *args, **kw)
        assert torch.equal(full[0], again[0]) and torch.equal(full[1], again[1]), "rerun differs"
    ntiles, grid = B // 16, min(256, B // 16)
    for wave in range(12):  # tile = block + grid * (wave + W * round), W = 8 or 12 waves per workgroup: one block per wave slot
        tile = 37 + grid * wave
        if tile >= ntiles:
            break
        lo = 16 * tile - 8  # straddles a tile boundary on purpose
        loss.particle0 = lo
        part = loss.simulate(ts, x0[lo:lo + 48].contiguous(), *args, **kw)
        assert torch.equal(part[0], full[0][lo:lo + 48]), f"wave slot {wave}: x_N of the block differs from the full launch"
        assert torch.equal(part[1], full[1][lo:lo + 48]), f"wave slot {wave}: log-weights of the block differ"
    loss.particle0 = 0
    assert bool(torch.isfinite(full[1]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("name,B", [("eubo_ei_gmm_d128_k4", 32768), ("eubo_em_gmm_d16_k4", 65536), ("eubo_dis_ei_d8", 65536), ("eubo_cmcd_gmm_d16", 65536)])
def test_compute_eubo_at_full_occupancy(gpu, name, B):
    from tests import build_cases as bc
    c = gc.load(name)
    b = bc.build(c, gpu)
    loss, ts, args = b["loss"], b["ts"], b["args"]
    kw = {k: v for k, v in b["kwargs"].items() if k == "initial_log_prob"}
    x0 = b["x0"].repeat((B + b["x0"].shape[0] - 1) // b["x0"].shape[0], 1)[:B].contiguous()
    loss.particle0 = 0
    xa = x0.clone()
    full = loss.compute_eubo(ts, xa, *args, **kw)
    xb = x0.clone()
    again = loss.compute_eubo(ts, xb, *args, **kw)
    assert torch.equal(full, again) and torch.equal(xa, xb), "rerun differs"  # (CMCD: x stays untouched, as upstream)
    for wave in range(8):
        tile = 37 + min(256, B // 16) * wave
        if tile >= B // 16:
            break
        lo = 16 * tile - 8
        loss.particle0 = lo
        xp = x0[lo:lo + 48].clone()
        part = loss.compute_eubo(ts, xp, *args, **kw)
        assert torch.equal(part, full[lo:lo + 48]) and torch.equal(xp, xa[lo:lo + 48]), f"wave slot {wave}: block differs from the full launch"
    loss.particle0 = 0
    assert bool(torch.isfinite(full).all())


@pytest.mark.gpu
def test_false_shared_variance_promise_is_loud(gpu):
    """sdeng_ref.shared_var is the CALLER's promise that all components share one variance vector; the matrix-pipe mixture path
    (RF_GMM_MM, 4 < K <= 64) reads only row 0 of vars_init.  The promise is checked on the device (k_same_var): a false one poisons the
    per-step table, so the call returns NaN everywhere instead of the scores of another mixture.  Through the C ABI
    directly -- the Python host derives the flag from the tensor itself and cannot lie."""
    import ctypes as C

    from sde_sampler_lrds_amd import _lib as L
    d, K, B, N = 64, 8, 256, 8
    var = 0.5 * torch.ones(K, d)
    var[3, 5] = 0.7  # one entry differs
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N, d=d, K=K, seed=3, ref_var=var)
    honest = loss.simulate(ts, x0, *args, **kw)  # the host sees the rows differ: vector path
    assert bool(torch.isfinite(honest[0]).all()) and bool(torch.isfinite(honest[1]).all())
    orig = E.ref_desc

    def lying(kind, utils, device, keep):
        r = orig(kind, utils, device, keep)
        r.shared_var = 1
        return r
    E.ref_desc = lying
    try:
        x, rnd, _ = loss.simulate(ts, x0, *args, **kw)
    finally:
        E.ref_desc = orig
    assert bool(torch.isnan(x).all()) and bool(torch.isnan(rnd).all())
    # ... and an honest shared-variance mixture of the same shape runs the matrix-pipe path and is finite
    loss2, ts2, x02, args2, kw2, _ = cfgs.build_rds_gmm(gpu, B, N, d=d, K=K, seed=3)
    ok = loss2.simulate(ts2, x02, *args2, **kw2)
    assert bool(torch.isfinite(ok[0]).all()) and bool(torch.isfinite(ok[1]).all())
