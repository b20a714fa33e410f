"""Step-loop time of the BASELINE workloads with the drift net at the reference's default initialisation magnitude (last layer x 1e-6:
stored scaled) against the livened net the bench uses.  SDENG_LIB: another library build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
if os.environ.get("SDENG_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["SDENG_LIB"])
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

dev = torch.device("cuda:0")
for cfg in ("rds_gmm", "pis_phi4", "cmcd_logreg"):
    B, N = cfgs.FULL_SIZE[cfg]
    for scaled in (False, True):
        loss, ts, x0, args, kw, info = cfgs.BUILDERS[cfg](dev, B, N)
        if scaled:
            with torch.no_grad():
                loss.generative_ctrl.base_model.out_layer.weight.mul_(1e-6)
                loss.generative_ctrl.base_model.out_layer.bias.mul_(1e-6)
        ev = L.HipEvents()
        loss.timing_events = ev
        best = 1e9
        for _ in range(6):
            x, rnd, _ = loss.simulate(ts, x0, *args, **kw)
            torch.cuda.synchronize()
            best = min(best, ev.elapsed_ms())
        print(f"{cfg}: last layer {'x 1e-6 (scaled image)' if scaled else 'as in the bench'}: kernel {best:.3f} ms, finite {bool(torch.isfinite(rnd).all())}", flush=True)
