"""Rerun determinism probe: the same simulate twice must be bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

if os.environ.get("SDENG_LIB"):
    L.LIB_PATH = os.environ["SDENG_LIB"]

dev = torch.device("cuda:0")


def probe(name, build, B, N, **kw):
    loss, ts, x0, args, kwargs, info = build(dev, B, N, **kw)
    loss.seed = 5
    a = loss.simulate(ts, x0, *args, **kwargs)
    torch.cuda.synchronize()
    for rep in range(3):
        b = loss.simulate(ts, x0, *args, **kwargs)
        torch.cuda.synchronize()
        dx = (a[0] - b[0]).abs()
        rows = (dx.amax(1) > 0).nonzero().flatten()
        dr = (a[1] - b[1]).abs()
        print(f"{name} B={B} N={N} {kw}: rerun {rep}: x rows differing {rows.numel()} max|dx| {float(dx.max()):.3e}  rnd differing {int((dr > 0).sum())} "
              f"first rows {rows[:8].tolist()} tiles {sorted(set((rows // 16).tolist()))[:8]}", flush=True)


if os.environ.get("PROBE_ONLY"):
    probe("cfg2 lib=" + os.environ.get("SDENG_LIB", "default"), cfgs.build_rds_gmm, 32768, 64)
    sys.exit(0)
for B in (2048, 32768, 32768 + 16, 65536):
    probe("cfg2", cfgs.build_rds_gmm, B, 64)
probe("cfg2", cfgs.build_rds_gmm, 65536, 256)
probe("cfg2", cfgs.build_rds_gmm, 131072, 256)
probe("cfg2-K16", cfgs.build_rds_gmm, 65536, 32, K=16)
probe("cfg3", cfgs.build_pis_phi4, 131072, 64)
probe("cfg4", cfgs.build_cmcd_logreg, 65536, 64)
