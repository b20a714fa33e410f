"""Five reruns of the cfg-2 step loop with one library: pairwise count of differing rows, and where the differing tiles of
(run 0, run 1) sit in the launch (workgroup = tile % grid, wave = tile // grid % 8, round = tile // (8 grid))."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

if os.environ.get("SDENG_LIB"):
    L.LIB_PATH = os.environ["SDENG_LIB"]
dev = torch.device("cuda:0")
B, N = int(os.environ.get("PROBE_B", 32768)), int(os.environ.get("PROBE_N", 64))
loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(dev, B, N)
loss.seed = 5
runs = []
for rep in range(5):
    x, rnd, _ = loss.simulate(ts, x0, *args, **kw)
    torch.cuda.synchronize()
    runs.append(x.clone())
name = os.path.basename(os.environ.get("SDENG_LIB", "default"))
print(name, "pairwise rows differing:")
for i in range(5):
    print("   ", [int(((runs[i] - runs[j]).abs().amax(1) > 0).sum()) for j in range(5)])
rows = ((runs[0] - runs[1]).abs().amax(1) > 0).nonzero().flatten()
tiles = sorted(set((rows // 16).tolist()))
grid = 256
print("    run0 vs run1: tiles differing", len(tiles), "of", B // 16,
      "by wave", sorted(collections.Counter((t // grid) % 8 for t in tiles).items()),
      "by round", sorted(collections.Counter(t // (8 * grid) for t in tiles).items()))
