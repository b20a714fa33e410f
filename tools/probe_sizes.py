"""Kernel timing probe: step-loop kernel time at the cfg-2 shape (optionally with an alternative library build)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
alt = os.environ.get("SDENG_LIB")
from sde_sampler_lrds_amd import _lib as L  # noqa: E402

if alt:
    L.LIB_PATH = alt
from sde_sampler_lrds_amd.experiments.baseline_configs import build_rds_gmm  # noqa: E402

dev = torch.device("cuda:0")
for (B, N, K) in [(2048, 100, 4), (65536, 64, 4), (65536, 256, 4), (131072, 256, 4), (65536, 256, 16)]:
    loss, ts, x0, args, _, info = build_rds_gmm(dev, B, N, K=K)
    fl = info["flops"]
    ev = L.HipEvents()
    loss.timing_events = ev
    for rep in range(3):
        t0 = time.perf_counter()
        x, rnd, _ = loss.simulate(ts, x0, *args)
        torch.cuda.synchronize()
        ms = ev.elapsed_ms()
    print(f"lib={alt or 'default'} B={B} N={N} K={K}: kernel {ms:.2f} ms -> {B*N/(ms*1e-3):.3e} p-steps/s ({fl*B*N/(ms*1e-3)/1e12:.1f} TFLOP/s)  rnd mean {rnd.mean().item():.4f}", flush=True)
