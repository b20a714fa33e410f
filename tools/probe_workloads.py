"""Step-loop kernel time on the other BASELINE.json configurations (random-init weights, synthetic data):
   cfg 3: PhiFour d=100, PIS (ScoreCtrl with the target score), Euler-Maruyama, 131 072 particles x 512 steps
   cfg 4: logistic regression d=61 (sonar-shaped synthetic design matrix), CMCD, 65 536 particles x 256 steps (one GPU's shard)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402

if os.environ.get("SDENG_LIB"):  # A/B against another build of the library
    L.LIB_PATH = os.environ["SDENG_LIB"]
from sde_sampler_lrds_amd.experiments.baseline_configs import build_cmcd_logreg, build_pis_phi4  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["pis", "cmcd"]
    for name in which:
        if name == "pis":
            B, N, d = 131072, 512, 100
            loss, ts, x0, args, kw, _ = build_pis_phi4(dev, B, N)
        else:
            B, N, d = 65536, 256, 61
            loss, ts, x0, args, kw, _ = build_cmcd_logreg(dev, B, N)
        loss.seed = 1
        ev = L.HipEvents()
        loss.timing_events = ev
        times = []
        for rep in range(8):  # the clocks need ~40 ms of work to ramp up: report the steady state (min of the later reps)
            x, rnd, _ = loss.simulate(ts, x0, *args, **kw)
            torch.cuda.synchronize()
            times.append(ev.elapsed_ms())
        ms = min(times[2:])
        fl = 2 * (2 * 64 * d + 2 * 64 * 64)
        print(f"{name}: B={B} N={N} d={d}: kernel {ms:.2f} ms -> {B*N/(ms*1e-3):.3e} p-steps/s ({fl*B*N/(ms*1e-3)/1e12:.1f} TFLOP/s alg.)  "
              f"rnd mean {rnd.mean().item():.4f} finite {bool(torch.isfinite(rnd).all())}", flush=True)
