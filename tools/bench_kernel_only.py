"""Run the cfg-2 simulate a few times (for rocprofv3 counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd.experiments.baseline_configs import build_rds_gmm  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", 65536))
N = int(os.environ.get("PROBE_N", 256))
loss, ts, x0, args, _, info = build_rds_gmm(dev, B, N, K=4)
fl = info["flops"]
for _ in range(int(os.environ.get("PROBE_REPS", 3))):
    loss.simulate(ts, x0, *args)
torch.cuda.synchronize()
print("done")
