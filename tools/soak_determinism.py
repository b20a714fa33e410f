"""Soak test: many reruns of the full-size BASELINE workloads must stay bit-identical to the first run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

dev = torch.device("cuda:0")
R = int(os.environ.get("SOAK_RERUNS", "60"))
from functools import partial  # noqa: E402

for name, build, B, N in (("cfg2 rds_ei_gmm", cfgs.build_rds_gmm, 65536, 256), ("cfg3 pis_phi4", cfgs.build_pis_phi4, 131072, 512),
                          ("cfg4 cmcd_logreg", cfgs.build_cmcd_logreg, 65536, 256),
                          ("cfg2 with K=16 modes (workgroup-shared table)", partial(cfgs.build_rds_gmm, K=16), 65536, 256),
                          ("cfg2 with K=64 modes (table staged in 2 pieces)", partial(cfgs.build_rds_gmm, K=64), 65536, 128)):
    loss, ts, x0, args, kw, _ = build(dev, B, N)
    loss.seed = 11
    ref = loss.simulate(ts, x0, *args, **kw)
    bad = 0
    for r in range(R):
        out = loss.simulate(ts, x0, *args, **kw)
        if not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {B} x {N}, {R} reruns, {bad} differed from the first run", flush=True)
