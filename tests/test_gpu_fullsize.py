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


def _tol(x_err, rnd_err, name):
    print(f"{name}: block [{P0},{P0 + PB}) vs oracle: x_N {x_err:.2e}, rnd {rnd_err:.2e}")
    assert x_err < 2e-4 and rnd_err < 2e-4


def _rnd_err(rnd, ref, scale):
    return float((rnd.cpu().flatten() - ref.flatten()).abs().max()) / scale


@pytest.mark.gpu
def test_cfg2_rds_gmm_65536x256(gpu):
    B, N = 65536, 256
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(gpu, B, N)
    loss.seed = 5
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw)
    _check_estimators(rnd)
    sde = orc.VP(0.1, 10.0, 1.0, 1.0)
    tgt = orc.GMMDiag(info["target"].loc.cpu(), info["target"].scale.cpu(), info["target"].mixture_weights.cpu())
    ctrl = orc.Ctrl(_sd(info["ctrl"]), "clipped", clip_model=1e4)
    means, var, w = info["means"].cpu(), 0.5 * torch.ones(info["K"], info["d"]), torch.ones(info["K"])
    loc0, v0 = sde.marginal_diag(torch.tensor(0.0), means, var)
    refd = orc.GMMDiag(loc0, v0.sqrt(), w)
    with torch.no_grad():
        ox, ornd, _ = orc.simulate_ei_ref(ts.cpu(), x0[P0:P0 + PB].cpu(), ctrl, sde, tgt.logp, refd.logp,
                                          lambda t, xx: orc.mog_score(xx, w, *sde.marginal_diag(t, means, var)),
                                          orc.PhiloxNoise(5, particle0=P0))
    _tol(gc.rel_err(x[P0:P0 + PB].cpu(), ox), _rnd_err(rnd[P0:P0 + PB], ornd, max(1.0, float(tgt.logp(ox).abs().max()))), "cfg2")


@pytest.mark.gpu
def test_cfg3_pis_phi4_131072x512(gpu):
    B, N = 131072, 512
    loss, ts, x0, args, kw, info = cfgs.build_pis_phi4(gpu, B, N)
    loss.seed = 6
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw)
    _check_estimators(rnd)
    g, T = math.sqrt(0.2), 5.0
    sde = orc.ScaledBM(g, T)
    tgt = orc.PhiFour(0.1, 0.0, info["d"], 20.0)
    ctrl = orc.Ctrl(_sd(info["ctrl"]), "score", clip_model=1e4, target_score=tgt.score, clip_score=1e4, scale_score=1.0)
    refd = orc.GaussDiag(torch.zeros(info["d"]), torch.full((info["d"],), g * math.sqrt(T)))
    with torch.no_grad():
        ox, ornd, _ = orc.simulate_em_ref(ts.cpu(), x0[P0:P0 + PB].cpu(), ctrl, sde, tgt.logp, refd.logp, None,
                                          orc.PhiloxNoise(6, particle0=P0))
    _tol(gc.rel_err(x[P0:P0 + PB].cpu(), ox), _rnd_err(rnd[P0:P0 + PB], ornd, max(1.0, float(ornd.abs().max()))), "cfg3")


@pytest.mark.gpu
def test_cfg4_cmcd_logreg_shard_65536x256(gpu):
    B, N = 65536, 256
    loss, ts, x0, args, kw, info = cfgs.build_cmcd_logreg(gpu, B, N)
    loss.seed = 7
    x, rnd, _ = _run_full_and_shards(loss, ts, x0, args, kw)
    _check_estimators(rnd)
    tgt = orc.LogReg(info["X"], info["y"], 4.5, -2.5, 0.5)
    prior = orc.GaussFull(info["mean"], info["cov"])
    ctrl = orc.Ctrl(_sd(info["ctrl"]), "score", clip_model=1e4, target_score=tgt.score, clip_score=1e4, scale_score=1.0)
    with torch.no_grad():
        ox, ornd, _ = orc.simulate_cmcd(ts.cpu(), x0[P0:P0 + PB].cpu(), ctrl, tgt.score, prior.score, 1.0, 1.0, 1e5, tgt.logp,
                                        prior.logp, orc.PhiloxNoise(7, particle0=P0))
    _tol(gc.rel_err(x[P0:P0 + PB].cpu(), ox), _rnd_err(rnd[P0:P0 + PB], ornd, max(1.0, float(ornd.abs().max()))), "cfg4")


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
    g, T = math.sqrt(0.2), 5.0
    sde = orc.ScaledBM(g, T)
    tgt = orc.PhiFour(0.1, 0.0, d, 20.0)
    ctrl = orc.Ctrl(_sd(info["ctrl"]), "score", clip_model=1e4, target_score=tgt.score, clip_score=1e4, scale_score=1.0)
    refd = orc.GaussDiag(torch.zeros(d), torch.full((d,), g * math.sqrt(T)))
    with torch.no_grad():
        ox, ornd, _ = orc.simulate_em_ref(ts.cpu(), x0[:64].cpu(), ctrl, sde, tgt.logp, refd.logp, None, orc.PhiloxNoise(3))
    x_err = gc.rel_err(x[:64].cpu(), ox)
    r_err = _rnd_err(rnd[:64], ornd, max(1.0, float(ornd.abs().max())))
    print(f"pis phi4 d={d}: x_N {x_err:.2e}, rnd {r_err:.2e}")
    assert x_err < 2e-4 and r_err < 2e-4 and bool(torch.isfinite(rnd).all())


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
    sde = orc.VP(0.1, 10.0, 1.0, 1.0)
    tgt = orc.GMMDiag(info["target"].loc.cpu(), info["target"].scale.cpu(), info["target"].mixture_weights.cpu())
    ctrl = orc.Ctrl(_sd(info["ctrl"]), "clipped", clip_model=1e4)
    means, var, w = info["means"].cpu(), 0.5 * torch.ones(K, d), torch.ones(K)
    loc0, v0 = sde.marginal_diag(torch.tensor(0.0), means, var)
    refd = orc.GMMDiag(loc0, v0.sqrt(), w)
    with torch.no_grad():
        ox, ornd, _ = orc.simulate_ei_ref(ts.cpu(), x0[p0:p0 + pb].cpu(), ctrl, sde, tgt.logp, refd.logp,
                                          lambda t, xx: orc.mog_score(xx, w, *sde.marginal_diag(t, means, var)),
                                          orc.PhiloxNoise(11, particle0=p0))
    x_err = gc.rel_err(x[p0:p0 + pb].cpu(), ox)
    r_err = _rnd_err(rnd[p0:p0 + pb], ornd, max(1.0, float(tgt.logp(ox).abs().max())))
    print(f"shared table d={d} K={K} B={B}: x_N {x_err:.2e}, rnd {r_err:.2e}")
    assert x_err < 2e-4 and r_err < 2e-4


# ---- every kernel family at full occupancy --------------------------------------------------------------------------
# The golden cases replicated to 32 768+ particles (each replica draws its own Philox noise): reruns and shards must be
# bit-identical, and blocks of the big run -- chosen so that every wave slot of a workgroup is covered -- must equal a
# small launch of just that block (one wave per SIMD: the regime the fixtures pin against the reference).
BIG = [("rds_ei_gmm_d128_k16", 32768), ("rds_ei_gmm_d8_k4", 65536), ("rds_ddpm_gmm_d16_snr", 65536), ("rds_em_gmm_d16", 65536),
       ("rds_ei_vp_default_d16", 65536), ("rds_ei_pbm_default_d16", 65536), ("dds_two_modes_d2", 65536), ("dds_rings_d2", 65536),
       ("dis_ei_d8", 65536), ("dis_orig_lerp_d8", 65536), ("pis_em_phi4_d100", 32768), ("cmcd_logreg_d61", 32768), ("cmcd_gmm_iso_d16", 65536), ("cmcd_gmm_diag_d40", 32768), ("cmcd_phi4_d100", 32768), ("pis_logreg_d61", 32768), ("dds_logreg_d61", 32768)]


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
    for wave in range(8):  # tile = block + grid * (wave + 8 * round): one block per wave slot
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
