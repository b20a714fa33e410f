"""Time the cfg-2 step-loop kernel under ablation builds (tools/abl_*.so.bin): which vector component costs what."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
from sde_sampler_lrds_amd import _lib as L
if os.environ.get("SDENG_LIB"): L.LIB_PATH = os.environ["SDENG_LIB"]
import bench
dev = torch.device("cuda:0")
loss, ts, x0, args, _, info = build_rds_gmm(dev, 65536, 256, K=4)
fl = info["flops"]
ev = L.HipEvents(); loss.timing_events = ev
best = 1e9
for rep in range(4):
    x, rnd, _ = loss.simulate(ts, x0, *args); torch.cuda.synchronize(); best = min(best, ev.elapsed_ms())
print(f"kernel {best:.2f} ms")
''' % ROOT
libs = [("full", None)] + [(os.path.basename(p)[4:-7], p) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so.bin")))]
for tag, path in libs:
    env = dict(os.environ)
    if path:
        dst = f"/tmp/{tag}.so"
        shutil.copy(path, dst)
        env["SDENG_LIB"] = dst
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"{tag:40s} {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
