"""Host-side cost of one pass of a BASELINE workload (the Python / ctypes work between two launches): enqueue time per simulate()
call without synchronising, and a cProfile of 30 calls.  PROBE_CFG as in bench_kernel_only.py."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import engine as E  # noqa: E402
from sde_sampler_lrds_amd import parallel  # noqa: E402
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs  # noqa: E402

dev = torch.device("cuda:0")
cfg = os.environ.get("PROBE_CFG", "cmcd_logreg")
B, N = cfgs.FULL_SIZE[cfg]
loss, ts, x0, args, kw, info = cfgs.BUILDERS[cfg](dev, 16, N)
x_in = E.InitialDraw(cfgs.prior_of(cfg, info, dev), B, dev)
for _ in range(3):
    loss.simulate(ts, x_in, *args, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
pend = []
for _ in range(10):
    _, rnd, _ = loss.simulate(ts, x_in, *args, **kw)
    pend.append(parallel.global_results_async(rnd, None))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{cfg}: enqueue {1e3 * (t1 - t0) / 10:.3f} ms per pass, total {1e3 * (t2 - t0) / 10:.3f} ms per pass")
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    loss.simulate(ts, x_in, *args, **kw)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
