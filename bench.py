#!/usr/bin/env python
"""Benchmark of the hot path: particle-steps/sec of one whole ``simulate`` (+ log-Z estimators), and log-Z abs-err vs the reference.

Headline workload (BASELINE.json configs[1]): ManyModes d=128 (K=4), RDS with a diagonal-GMM reference, VP(0.1,10),
exponential integrator, 65 536 particles x 256 steps per GPU, FourierMLP drift net; x0 and the step noise are drawn by the engine
(Philox, keyed by the global particle index; x0 by a sampler kernel inside the timed pass, the noise in the step loop's registers).  A "step" of this bench = one full pass: all 256 SDE steps of the batch, terminal cost,
and the log-Z / ESS reduction (the window the reference times as eval/sample_time, solver/oc.py:148-158); the inputs are the
seed and the model.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rds_gmm|pis_phi4|cmcd_logreg]

For N>1 launch with torch.distributed.run (one rank per GPU, RCCL): the particle batch is sharded (weak scaling: 65 536 particles
per rank), no collective inside the step loop, one 36-byte all-gather for the final log-Z / ESS.

Prints ONE JSON line (rank 0): the contract keys for the headline workload, plus
  roofline       what binds the step-loop kernel: the SIMD vector-issue port (`bound: valu_issue`, frac = measured port-busy share, from the
                 counter record of THIS library build) -- or, without such a record, the matrix pipe's live utilisation; plus the
                 matrix-pipe (`mfma_pipe`) and FP32-equivalent (`fp32_equiv`) figures, the issue model (`issue`) and HBM traffic (PMC)
  cpu_baseline   the CPU oracle (a port of the reference's torch loop) on a bounded sample, host cores stated
  log_z_abs_err  |log Z_HIP - log Z_oracle| on a block of 2 048 particles with identical seeds (x0 and noise), outside every timed window
  other_configs  the same five items for configs[2] (PhiFour PIS, 131 072 x 512) and configs[3] (CMCD logistic regression, one GPU's
                 shard 65 536 x 256) -- N = 1 only
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense FP32 MFMA = packed-fp32 vector peak (256 FLOP/clk/CU)
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/F16 MFMA ~2.5 PF dense
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
# Issue cost per instruction on one SIMD with two resident waves, ns (tools/ubench/valu_cost.hip, profiles/r01_ubench_valu_cost.log:
# the "2w" column / 2).  MFMA and vector instructions of the waves of a SIMD do not overlap on gfx950
# (profiles/r01_ubench_mfma_valu_serialize.log), so a SIMD's time per tile-step is at least the sum over classes of count x cost.
ISSUE_NS = {"fp32 fma/mul/add": 1.335, "transcendental": 3.55, "int64 mad (Philox)": 2.00, "int32 mul": 2.0, "convert": 1.89,
            "other vector (logic, select, move, cross-lane)": 1.14, "mfma 16x16x32 f16": 7.11}
Z_BLOCK = 2048                 # particles of the log-Z parity block


def host_cores() -> int:
    """CPU share of this process: cgroup quota if one is set, else the affinity mask, capped at 16 (the share a
    one-GPU box grants; the machine itself reports hundreds of cores and oversubscribing them stalls torch)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def oracle_leg(cfg, info, ts, hip_block, seed, N, cpu_budget_s, chunk):
    """Everything that touches oracle/ (the checker): (a) log-Z / end-point parity of the HIP block against the oracle run with the
    same seeds, (b) the CPU baseline -- the oracle timed on a bounded sample of the same workload, all host cores.  Neither is
    inside a GPU-timed window."""
    from oracle import baseline_oracles as bo
    from oracle import sde_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    run = bo.runner(cfg, info, ts)
    out = {}
    # (a) identical seeds: x0 = the engine's stream-1 draw, noise = its stream-0 draws, both restated on the CPU
    hx, hrnd = hip_block
    x0 = bo.initial_particles(cfg, info, seed, 0, hx.shape[0])
    ox, ornd, scale = run(x0, orc.PhiloxNoise(seed, particle0=0))
    lz_hip, lz_orc = bo.log_z(hrnd.cpu()), bo.log_z(ornd)
    out["log_z_abs_err"] = abs(lz_hip - lz_orc)
    out["parity"] = {"block": f"particles [0, {hx.shape[0]}) x {N} steps, identical seeds (x0: Philox stream 1, noise: stream 0)",
                     "log_z_hip": lz_hip, "log_z_oracle": lz_orc,
                     "x_N_max_rel_err": float(((hx.cpu() - ox).abs() / ox.abs().clamp(min=1.0)).max()),
                     "rnd_max_err_rel_to_largest_summand": float((hrnd.cpu().flatten() - ornd.flatten()).abs().max()) / scale,
                     # the plain figure next to it: log-weights are sums of terminal log-densities of magnitude `scale` that largely
                     # cancel, so one fp32 ulp of a summand (scale x 6e-8) is already 1e-5 of a small |rnd|
                     "rnd_max_rel_err_plain": float(((hrnd.cpu().flatten() - ornd.flatten()).abs() / ornd.flatten().abs().clamp(min=1e-30)).max()),
                     "rnd_median_rel_err_plain": float(((hrnd.cpu().flatten() - ornd.flatten()).abs() / ornd.flatten().abs().clamp(min=1e-30)).median()),
                     "rnd_largest_summand": scale}
    # (b) CPU baseline: torch's own generator for the noise, like the reference
    if cpu_budget_s > 0:
        d = info["d"]
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            run(bo.initial_particles(cfg, info, seed, 0, 64), orc.TorchNoise())  # warm-up (thread pool, allocator)
            done, wall = 0, 0.0
            while wall < cpu_budget_s:
                xc = torch.zeros(chunk, d) if cfg == "pis_phi4" else torch.randn(chunk, d, generator=gen)
                if cfg == "cmcd_logreg":
                    xc = info["mean"] + xc @ torch.linalg.cholesky(info["cov"]).T
                t0 = time.perf_counter()
                _, rnd, _ = run(xc, orc.TorchNoise())
                orc.compute_results(rnd)
                wall += time.perf_counter() - t0
                done += chunk
                print(f"[cpu_baseline {cfg}] {done} particles x {N} steps in {wall:.1f} s on {cores} threads", file=sys.stderr, flush=True)
        out["cpu_baseline"] = dict(value=done * N / wall, unit="particle-steps/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"{done} particles x {N} steps of the same workload (chunks of {chunk}), torch CPU fp32, {wall:.1f} s")
    return out


def library_digest() -> str:
    """Digest of the kernel sources + build flags the loaded libsdeng.so was built from (sde_sampler_lrds_amd/build.py)."""
    from sde_sampler_lrds_amd import build as B
    return B._digest()


def pmc_record(cfg):
    """The committed counter record of this workload's step-loop kernel -- only if it was collected with THIS build of the library
    (``library_digest`` stamped by tools/pmc_passes.sh equals the digest of the sources here).  A kernel change without a PMC refresh
    therefore drops the counter-derived fields from the line instead of quoting stale counters."""
    import glob
    dig = library_digest()
    for pj in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{cfg}.json")), reverse=True):
        j = json.load(open(pj))
        if j.get("library_digest") == dig:
            return j, os.path.relpath(pj, ROOT)
    return None, None


def mfma_per_tile_step(cfg, info):
    """v_mfma_f32_16x16x32_f16 instructions one 16-particle tile issues per SDE step: 3 split products x (output tiles x K-blocks) of
    every GEMM in the step (sim_device.hpp dense / dense_pre; cmcd_kernel.hpp).  Equals SQ_INSTS_MFMA / tile-steps of the PMC passes."""
    nt = (info["d"] + 15) // 16
    kb = (nt + 1) // 2
    net = 3 * (4 * kb + 4 * 2 + 4 * 2 + nt * 2)
    if cfg != "cmcd_logreg":
        return net
    n = info["X"].shape[0]
    return net + 3 * (((n + 15) // 16) * kb + nt * ((n + 31) // 32) + nt * kb)  # + logits, gradient and prior-precision products


def roofline(cfg, info, B, N, k_ms, extra_flops=0):
    """What bounds the step-loop kernel, and how close it runs to that bound.

    The kernel keeps the state on chip (HBM: 0.5 % of the per-step-launch traffic) and its GEMMs are small, so neither HBM nor the
    matrix pipe binds: the SIMD's vector-instruction issue does.  Headline: ``bound = "valu_issue"``, ``frac`` = the measured share
    of SIMD cycles in which the vector port is issuing (PMC, <= 1).  That needs the counter record of THIS library build
    (``pmc_record``); without one the headline falls back to the matrix pipe's own utilisation, measured live: f16 FLOP issued
    (split products included) over the dense f16 MFMA peak.  The FP32-equivalent figure of earlier rounds stays as a named
    sub-field: it is an accounting equivalence, not a utilisation."""
    flops_ps = info["flops"] + extra_flops
    tile_steps = ((B + 15) // 16) * N
    fp32_equiv = flops_ps * B * N / (k_ms * 1e-3) / 1e12
    issued = mfma_per_tile_step(cfg, info) * tile_steps * 16384.0 / (k_ms * 1e-3) / 1e12  # one 16x16x32 MFMA = 2*16*16*32 FLOP
    r = {"bound": "mfma", "achieved": issued, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": issued / PEAK_F16_MFMA_TFLOPS,
         "traffic": None, "kernel_ms": k_ms, "algorithmic_flops_per_particle_step": flops_ps,
         "algorithmic_hbm_bytes_per_launch": (2 * info["d"] + 1) * 4 * B,  # x0 in (written by the sampler kernel), x_N and rnd out
         "mfma_pipe": {"issued_tflops_f16": issued, "peak": PEAK_F16_MFMA_TFLOPS, "frac": issued / PEAK_F16_MFMA_TFLOPS,
                       "mfma_per_tile_step": mfma_per_tile_step(cfg, info),
                       "reading": "what the matrix pipe executes (three f16 split products per fp32 product) over the dense f16 MFMA peak, "
                                  "from the kernel time measured in this run: the pipe idles while the vector instructions of the same SIMD issue"},
         "fp32_equiv": {"tflops": fp32_equiv, "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS, "ratio": fp32_equiv / PEAK_FP32_MFMA_TFLOPS,
                        "reading": "algorithmic fp32 drift-net FLOP per second over the dense FP32 MFMA rate: an accounting equivalence (results "
                                   "carry fp32 accuracy, the products run on the f16 pipe at 3/16 of an fp32 MFMA's time), NOT a utilisation -- "
                                   "it is not bounded by 1"},
         "note": "no counter record for this library build: headline = matrix-pipe utilisation measured live; run tools/pmc_passes.sh to "
                 "add the vector-issue figures"}
    from sde_sampler_lrds_amd.experiments.baseline_configs import FULL_SIZE
    j, src = pmc_record(cfg) if ((B, N) == FULL_SIZE[cfg] and info.get("K", 4) == 4) else (None, None)
    if j is None:
        return r
    c = j["counters_per_launch"]
    if j.get("traffic"):
        r["traffic"] = j["traffic"]["bytes"]
        r["traffic_unit"] = "bytes/launch"
        r["traffic_source"] = f"{src} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
    fp = c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]
    counts = {"fp32 fma/mul/add": fp, "transcendental": c["SQ_INSTS_VALU_TRANS_F32"], "int64 mad (Philox)": c["SQ_INSTS_VALU_INT64"],
              "int32 mul": c["SQ_INSTS_VALU_INT32"], "convert": c["SQ_INSTS_VALU_CVT"], "mfma 16x16x32 f16": c["SQ_INSTS_MFMA"]}
    counts["other vector (logic, select, move, cross-lane)"] = max(0.0, c["SQ_INSTS_VALU"] - sum(counts.values()))
    model_ms = sum(counts[k] * ISSUE_NS[k] for k in counts) / N_SIMD * 1e-6
    wps = int(j.get("waves_per_simd") or (3 if cfg == "pis_phi4" else 2))  # waves per SIMD of the instantiation (sim_kernel.hpp sd_waves_of)
    r["issue"] = {"instr_per_tile_step": {k: v / tile_steps for k, v in counts.items()},
                  "vector_instr_per_tile_step": c["SQ_INSTS_VALU"] / tile_steps, "issue_cost_ns": ISSUE_NS,
                  "model_ms": model_ms, "model_frac_of_kernel": model_ms / k_ms, "source": f"{src} (SQ_INSTS_* per launch), "
                  "profiles/r01_ubench_valu_cost.log (cost per class)",
                  "reading": "model_ms = sum over classes of count x issue cost / 1024 SIMDs: the time the kernel's own instruction stream "
                             "needs at full issue rate"}
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"):
        if k in c:
            r["issue"][k] = c[k]
    if "SQ_ACTIVE_INST_VALU" in c and c.get("SQ_WAVE_CYCLES"):
        busy = min(1.0, wps * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"])
        r.update(bound="valu_issue", achieved=busy, peak=1.0, unit="fraction of SIMD cycles with the vector-issue port busy", frac=busy,
                 note=f"binding resource: vector-instruction issue per SIMD.  frac = waves per SIMD ({wps}) x SQ_ACTIVE_INST_VALU / "
                      f"SQ_WAVE_CYCLES from {src}, counters of this library build (digest {j['library_digest'][:12]}); the kernel gets "
                      "faster by issuing fewer vector instructions or by hiding them under its own MFMAs.  `mfma_pipe` and `fp32_equiv` are the "
                      "matrix-side figures, measured live")
    return r


def measure(cfg, device, B, N, steps, warmup, spinup, dist, rank, world, modes=4):
    """Timed passes of one workload: W warm-up, K timed, barrier + synchronize on both sides, max over ranks."""
    from sde_sampler_lrds_amd import _lib as L
    from sde_sampler_lrds_amd import engine as E
    from sde_sampler_lrds_amd import parallel
    from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
    kwb = dict(K=modes) if cfg == "rds_gmm" else {}
    loss, ts, _, args, kw, info = cfgs.BUILDERS[cfg](device, 16, N, **kwb)  # the model; x0 is drawn by the engine
    prior = cfgs.prior_of(cfg, info, device)
    loss.seed = 1
    loss.particle0 = rank * B  # global particle index -> sharding-independent x0 and noise
    x0 = E.InitialDraw(prior, B, device)
    ev = L.HipEvents()
    loss.timing_events = ev

    def one_pass():
        """simulate + terminal cost + log-Z / ESS reduction (+ all-gather), all enqueued on the stream; the 36-byte
        result is read back by .result() -- after the timed region for all but the last pass, so that consecutive
        passes run back to back (a sampler in production does not idle the GPU between batches either)."""
        _, rnd, _ = loss.simulate(ts, x0, *args, **kw)
        return parallel.global_results_async(rnd, dist)

    if dist is not None:  # create the RCCL communicator now (~20 ms the first time): an idle gap right before the timed
        dist.barrier()    # region would let the clocks drop again
        one_pass().result()
        dist.barrier()
    # bring the GPU to its sustained clocks first: the same pass, untimed (the first ~40 ms after idle run ~15 % slower than
    # steady state); rank-local passes WITHOUT the all-gather: their number is time-based and may differ between ranks
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < spinup:
        _, rnd_spin, _ = loss.simulate(ts, x0, *args, **kw)
        parallel.global_results_async(rnd_spin, None).result()
    for _ in range(warmup):
        one_pass().result()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = [one_pass() for _ in range(steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    results = [p.result() for p in pending]  # every pass produced its estimators
    assert all(math.isfinite(r["log_norm_const_is"]) for r in results)
    if dist is not None:
        wt = torch.tensor([wall], device=device)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        wall = wt.item()
    # step-loop kernel duration: HIP events recorded by sdeng_simulate around that launch, on the launch stream
    samples = [ev.elapsed_ms()]
    for _ in range(min(5, steps)):
        loss.simulate(ts, x0, *args, **kw)
        samples.append(ev.elapsed_ms())
    k_ms = sum(samples) / len(samples)
    # the log-Z parity block (identical seeds), outside the timed region
    loss.particle0 = 0
    loss.timing_events = None
    hx, hrnd, _ = loss.simulate(ts, E.InitialDraw(prior, Z_BLOCK, device), *args, **kw)
    torch.cuda.synchronize()
    return dict(cfg=cfg, B=B, N=N, info=info, ts=ts, wall=wall, steps=steps, value=world * B * N * steps / wall, ms_per_step=1e3 * wall / steps,
                kernel_ms=k_ms, res=results[-1], hip_block=(hx, hrnd), seed=1)


def path_of(cfg, info):
    if cfg == "rds_gmm":
        K = info["K"]
        small = K <= 4 and K * 2 * 16 * ((info["d"] + 15) // 16) <= 1024
        return (f"k_simulate<NT={(info['d'] + 15) // 16},REF={'GMM' if small else 'GMM_BIG'},SC=NONE,FORM=LIN>: "
                + ("K <= 4: responsibilities in registers, per-wave LDS table by LDS-DMA; all components share one variance vector "
                   "(variances_init = 0.5, the reference's default initialisation) -> centred shared-variance form, one fma per element and "
                   "component (a fitted reference with distinct variances runs the general form: 3 + 2 per element and component); "
                   "8 waves per workgroup = 2 per SIMD" if small
                   else "K > 4: workgroup-shared double-buffered LDS table, online softmax"))
    if cfg == "pis_phi4":
        return f"k_simulate<NT={(info['d'] + 15) // 16},REF=NONE,SC=PHI4,FORM=EM>: ScoreCtrl with the phi^4 lattice score in registers (neighbour exchange by cross-lane moves); 12 waves per workgroup = 3 per SIMD (the step loop fits 168 registers)"
    return ("k_simulate_cmcd<NT=4,LOGREG>: one drift-net + one annealed-score evaluation per step (the reference: 2 + 4), design matrix in "
            "LDS as split-f16 MFMA images, full-covariance prior precision through L2; 8 waves per workgroup = 2 per SIMD")


def config_entry(m, cpu_budget_s, chunk, with_cpu):
    cfg, info = m["cfg"], m["info"]
    extra = 0
    if cfg == "cmcd_logreg":  # logits + gradient products of the logistic-regression score and the prior precision product, per step
        extra = 2 * 2 * info["X"].shape[0] * info["d"] + 2 * info["d"] ** 2
    e = {"workload": info["workload"] + f", {m['B']} particles x {m['N']} steps", "particles": m["B"], "sde_steps": m["N"],
         "value": m["value"], "unit": "particle-steps/s", "ms_per_step": m["ms_per_step"], "steps": m["steps"], "kernel_ms": m["kernel_ms"],
         "path": path_of(cfg, info), "log_norm_const_is": m["res"]["log_norm_const_is"], "ess": m["res"]["ess"],
         "roofline": roofline(cfg, info, m["B"], m["N"], m["kernel_ms"], extra)}
    e.update(oracle_leg(cfg, info, m["ts"], m["hip_block"], m["seed"], m["N"], cpu_budget_s if with_cpu else 0.0, chunk))
    if "cpu_baseline" in e:
        e["speedup_vs_cpu"] = e["value"] / e["cpu_baseline"]["value"]
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["rds_gmm", "pis_phi4", "cmcd_logreg"], default="rds_gmm",
                    help="BASELINE.json configs[1] (headline, default) / configs[2] / configs[3]; honoured with --gpus N "
                         "(cmcd_logreg at 4 ranks = BASELINE's 262 144 particles over 4 GPUs, 65 536 per rank)")
    ap.add_argument("--particles", type=int, default=None, help="per GPU (default: the workload's BASELINE size)")
    ap.add_argument("--sde-steps", type=int, default=None)
    ap.add_argument("--modes", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--spinup", type=float, default=0.3, help="seconds of untimed passes before the warm-up (clock ramp)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("SDENG_BENCH_FORCE_DIST"):  # the env knob rehearses the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    from sde_sampler_lrds_amd.experiments.baseline_configs import FULL_SIZE
    wl = a.workload
    B, N = FULL_SIZE[wl]
    if a.particles is not None:
        B = a.particles
    if a.sde_steps is not None:
        N = a.sde_steps
    head = measure(wl, device, B, N, a.steps, a.warmup, a.spinup, dist, rank, world, modes=a.modes)
    others = []
    if world == 1 and not a.no_other_configs and wl == "rds_gmm":
        for cfg in ("pis_phi4", "cmcd_logreg"):
            others.append(measure(cfg, device, *FULL_SIZE[cfg], a.steps, a.warmup, a.spinup, None, 0, 1))
    if rank == 0:
        with_cpu = world == 1 and not a.no_cpu_baseline
        chunks = {"rds_gmm": 8192, "pis_phi4": 4096, "cmcd_logreg": 2048}
        e = config_entry(head, 15.0, chunks[wl], with_cpu)
        desc = {"rds_gmm": f"ManyModes d=128 K={a.modes}, RDS gmm-ref, VP(0.1,10), EI integrator",
                "pis_phi4": "PhiFour d=100, PIS target-informed drift, EM integrator",
                "cmcd_logreg": "LogisticRegression d=61 (sonar-shaped synthetic design matrix), CMCD lv-loss sampler"}[wl]
        out = {
            "metric": "particle-steps/sec (batch*n_steps/wall) + log-Z abs-err vs ref", "value": head["value"], "unit": "particle-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{desc}, {B} particles x {N} steps per GPU, FourierMLP(4x64) drift, x0 and noise drawn by the engine (Philox)",
                       "name": wl, "particles_per_gpu": B, "sde_steps": N, "parallelism": f"particle-sharded x{world}"},
            "log_z_abs_err": e["log_z_abs_err"], "parity": e["parity"], "path": e["path"],
            "log_norm_const_is": e["log_norm_const_is"], "ess": e["ess"], "spinup_s": a.spinup, "roofline": e["roofline"],
            "library_digest": library_digest(),
        }
        out["roofline"]["kernel"] = e["path"].split(":")[0] + " (one launch = all sde_steps of the batch)"
        if "cpu_baseline" in e:
            out["cpu_baseline"] = e["cpu_baseline"]
            out["speedup_vs_cpu"] = e["speedup_vs_cpu"]
        if others:
            out["other_configs"] = [config_entry(m, 8.0, chunks[m["cfg"]], with_cpu) for m in others]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
