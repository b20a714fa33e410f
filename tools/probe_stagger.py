"""Sweep the wave-stagger delay (SDENG_STAGGER) at the cfg-2 shape; one subprocess per value (read once per process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
import bench
from sde_sampler_lrds_amd import _lib as L
dev = torch.device("cuda:0")
loss, ts, x0, args, parts, fl = bench.build_rds_gmm(dev, 65536, 256, K=4)
ev = L.HipEvents(); loss.timing_events = ev
best = 1e9
for rep in range(4):
    x, rnd, _ = loss.simulate(ts, x0, *args); torch.cuda.synchronize(); best = min(best, ev.elapsed_ms())
print(f"kernel {best:.2f} ms -> {65536*256/(best*1e-3):.3e} p-steps/s ({fl*65536*256/(best*1e-3)/1e12:.1f} TFLOP/s) rnd mean {rnd.mean().item():.5f}")
''' % ROOT
for v in sys.argv[1:] or ["0", "2", "4", "6", "8"]:
    env = dict(os.environ, SDENG_STAGGER=v)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"stagger={v}: {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)
