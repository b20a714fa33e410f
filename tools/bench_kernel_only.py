"""Run the step loop of one BASELINE.json workload a few times (for rocprofv3 kernel-trace / counter passes).
PROBE_CFG = rds_gmm (cfg 2, default) | pis_phi4 (cfg 3) | cmcd_logreg (cfg 4); PROBE_B / PROBE_N / PROBE_K / PROBE_REPS override sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import engine as E  # noqa: E402
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

dev = torch.device("cuda:0")
cfg = os.environ.get("PROBE_CFG", "rds_gmm")
B0, N0 = cfgs.FULL_SIZE[cfg]
B = int(os.environ.get("PROBE_B", B0))
N = int(os.environ.get("PROBE_N", N0))
kw_build = {"K": int(os.environ["PROBE_K"])} if os.environ.get("PROBE_K") else {}
loss, ts, x0, args, kw, info = cfgs.BUILDERS[cfg](dev, B, N, **kw_build)
x_in = E.InitialDraw(cfgs.prior_of(cfg, info, dev), B, dev) if os.environ.get("PROBE_DRAW", "1") == "1" else x0
for _ in range(int(os.environ.get("PROBE_REPS", 3))):
    loss.simulate(ts, x_in, *args, **kw)
torch.cuda.synchronize()
print("done", cfg, B, N)
