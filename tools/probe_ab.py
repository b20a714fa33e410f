"""A/B of library builds on the step loop of one workload: SDENG_LIBS = space-separated library paths ('default' = the in-tree build),
PROBE_CFG = rds_gmm (default) | pis_phi4 | cmcd_logreg, PROBE_D = another dimension for the same workload."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = '''
import os, sys, torch
sys.path.insert(0, %r)
from sde_sampler_lrds_amd import _lib as L
alt = os.environ.get("SDENG_LIB")
if alt: L.LIB_PATH = alt
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
dev = torch.device("cuda:0")
cfg = os.environ.get("PROBE_CFG", "rds_gmm")
B, N = cfgs.FULL_SIZE[cfg]
extra = {'d': int(os.environ['PROBE_D'])} if os.environ.get('PROBE_D') else {}
loss, ts, x0, args, kw, info = cfgs.BUILDERS[cfg](dev, B, N, **extra)
ev = L.HipEvents(); loss.timing_events = ev
best = 1e9
for rep in range(12):
    x, rnd, _ = loss.simulate(ts, x0, *args, **kw); torch.cuda.synchronize(); best = min(best, ev.elapsed_ms())
print(f"{os.path.basename(alt) if alt else 'default':24s} kernel {best:.3f} ms  rnd mean {rnd.mean().item():.5f}")
''' % ROOT
for lib in os.environ.get("SDENG_LIBS", "default").split():
    env = dict(os.environ)
    if lib != "default":
        env["SDENG_LIB"] = lib
    subprocess.run([sys.executable, "-c", CODE], env=env, check=False)
