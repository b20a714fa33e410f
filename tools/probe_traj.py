"""return_traj=True (the PAR = 1 kernels, what compute_results' first pass runs, solver/oc.py:139-145): kernel time and trajectory-store
bandwidth ((N+1) B d 4 bytes / kernel time) for the three workloads, next to the no-trajectory kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

dev = torch.device("cuda:0")
for cfg, B, N in (("rds_gmm", 65536, 256), ("pis_phi4", 65536, 256), ("cmcd_logreg", 65536, 256)):
    loss, ts, x0, args, kw, info = cfgs.BUILDERS[cfg](dev, B, N)
    ev = L.HipEvents()
    loss.timing_events = ev
    res = {}
    for traj in (False, True):
        best = 1e9
        for rep in range(6):
            out = loss.simulate(ts, x0, *args, return_traj=traj, **kw)
            torch.cuda.synchronize()
            best = min(best, ev.elapsed_ms())
            del out
        res[traj] = best
    gb = (N + 1) * B * info["d"] * 4 / 1e9
    print(f"{cfg}: B={B} N={N} d={info['d']}: kernel {res[False]:.2f} ms, with trajectory {res[True]:.2f} ms (+{res[True] - res[False]:.2f} ms for {gb:.2f} GB "
          f"-> {gb / (res[True] * 1e-3) / 1e3:.2f} TB/s over the whole kernel, {gb / (max(res[True] - res[False], 1e-9) * 1e-3) / 1e3:.2f} TB/s marginal)", flush=True)
