"""Time the cfg-2 step-loop kernel under ablation builds: which vector component costs what.

Build the variants first (one translation unit recompiled with a debug define, the rest of the library reused), e.g.

    tools/variant_lib.sh NOPHILOX sim_8_2_0 -DSD_DBG_NOPHILOX
    tools/variant_lib.sh NOREF    sim_8_2_0 -DSD_DBG_NOREF
    tools/variant_lib.sh NOGELU   sim_8_2_0 -DSD_DBG_NOGELU

then run this script on the GPU box: it times the default library and every tools/var_*.so.bin it finds.  The variants
compute garbage by construction -- they exist to be timed, never to be checked.
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
from sde_sampler_lrds_amd import _lib as L
if os.environ.get("SDENG_LIB"): L.LIB_PATH = os.environ["SDENG_LIB"]
from sde_sampler_lrds_amd.experiments.baseline_configs import build_rds_gmm
dev = torch.device("cuda:0")
loss, ts, x0, args, _, info = build_rds_gmm(dev, 65536, 256, K=4)
ev = L.HipEvents(); loss.timing_events = ev
best = 1e9
for rep in range(8):
    loss.simulate(ts, x0, *args); torch.cuda.synchronize(); best = min(best, ev.elapsed_ms())
print(f"kernel {best:.2f} ms")
''' % ROOT
libs = [("default", None)] + [(os.path.basename(p)[4:-7], p) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "var_*.so.bin")))]
for tag, path in libs:
    env = dict(os.environ)
    if path:
        env["SDENG_LIB"] = path
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"{tag:40s} {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
