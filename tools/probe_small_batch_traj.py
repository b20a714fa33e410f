"""Small batches with and without the trajectory output, standard against split-tile kernel (cfg 2 model, d=128, K=4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L
if os.environ.get("SDENG_LIB"):  # A/B against another build of the library
    L.LIB_PATH = os.path.abspath(os.environ["SDENG_LIB"])
from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
dev = torch.device("cuda:0")
for B, N in ((6000, 256), (2048, 256), (512, 100)):
    loss, ts, x0, args, kw, info = cfgs.build_rds_gmm(dev, B, N)
    ev = L.HipEvents(); loss.timing_events = ev
    for split in (False, True):
        for traj in (False, True):
            loss.split_tiles = split
            best = 1e9
            for _ in range(8):
                loss.simulate(ts, x0, *args, return_traj=traj); torch.cuda.synchronize(); best = min(best, ev.elapsed_ms())
            print(f"B={B} N={N} split_tiles={split} return_traj={traj}: step-loop kernel {best:.3f} ms", flush=True)
