"""Where do two reruns of the same simulate first differ?  (trajectory dump, per step / feature statistics)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

if os.environ.get("SDENG_LIB"):
    L.LIB_PATH = os.environ["SDENG_LIB"]
dev = torch.device("cuda:0")
B, N = 32768, int(os.environ.get("PROBE_N", "16"))
loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(dev, B, N)
loss.seed = 5
runs = []
for rep in range(4):
    x, rnd, xs = loss.simulate(ts, x0, *args, return_traj=True, **kw)
    torch.cuda.synchronize()
    runs.append((x.clone(), rnd.clone(), xs.clone()))
ref = runs[0]
for rep in range(1, 4):
    d = (runs[rep][2] - ref[2]).abs()  # [N+1,B,d]
    for k in range(N + 1):
        rows = (d[k].amax(1) > 0).nonzero().flatten()
        if rows.numel():
            cols = (d[k].amax(0) > 0).nonzero().flatten()
            r0 = int(rows[0])
            print(f"run{rep} vs run0: first differing step {k}: rows {rows.numel()} (first {rows[:6].tolist()}, tiles {sorted(set((rows // 16).tolist()))[:6]}), "
                  f"cols {cols.numel()} (first {cols[:12].tolist()}), max|d| {float(d[k].max()):.3e}, row {r0}: cols {(d[k][r0] > 0).nonzero().flatten()[:16].tolist()} "
                  f"vals {runs[rep][2][k][r0][(d[k][r0] > 0)][:4].tolist()} vs {ref[2][k][r0][(d[k][r0] > 0)][:4].tolist()}", flush=True)
            break
    else:
        print(f"run{rep} vs run0: identical trajectories; rnd differing {int(((runs[rep][1] - ref[1]).abs() > 0).sum())}")
