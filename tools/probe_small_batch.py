"""Small-batch latency of the step loop (few tiles: one wave per SIMD or fewer), optionally with an alternative library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
alt = os.environ.get("SDENG_LIB")
from sde_sampler_lrds_amd import _lib as L  # noqa: E402

if alt:
    L.LIB_PATH = alt
from sde_sampler_lrds_amd.experiments.baseline_configs import build_rds_gmm  # noqa: E402

dev = torch.device("cuda:0")
for (B, N) in [(512, 100), (2048, 256), (4096, 256), (6000, 256), (8192, 256), (16384, 256), (32768, 256)]:
    loss, ts, x0, args, _, info = build_rds_gmm(dev, B, N, K=4)
    ev = L.HipEvents()
    loss.timing_events = ev
    for split in (False, True):  # True: SDENG_FLAG_SPLIT_TILES (four waves per tile; honoured up to 16 384 particles)
        loss.split_tiles = split
        best = 1e9
        for rep in range(6):
            x, rnd, _ = loss.simulate(ts, x0, *args)
            torch.cuda.synchronize()
            best = min(best, ev.elapsed_ms())
        print(f"lib={os.path.basename(alt) if alt else 'default'} B={B} N={N} split_tiles={split}: kernel {best:.3f} ms = {1e3 * best / N:.2f} us/step -> "
              f"{B*N/(best*1e-3):.3e} p-steps/s  rnd mean {rnd.mean().item():.4f}", flush=True)
