"""Kernel time vs batch size / step count at the cfg-2 shape: separates per-launch overhead from per-tile-step cost."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd.experiments.baseline_configs import build_rds_gmm  # noqa: E402

dev = torch.device("cuda:0")
for (B, N) in [(32768, 256), (65536, 256), (98304, 256), (131072, 256), (262144, 256), (65536, 64), (65536, 128), (65536, 512), (65536, 1024)]:
    loss, ts, x0, args, _, info = build_rds_gmm(dev, B, N)
    ev = L.HipEvents()
    loss.timing_events = ev
    t = []
    for rep in range(6):
        loss.simulate(ts, x0, *args)
        torch.cuda.synchronize()
        t.append(ev.elapsed_ms())
    best = min(t[1:])
    print(f"B={B:7d} N={N:5d}: kernel ms {' '.join(f'{v:.2f}' for v in t)}  best {best:.2f} -> {B*N/(best*1e-3):.3e} p-steps/s, {best*1e6/(B/16/2048*N):.0f} ns per tile-step-round", flush=True)
