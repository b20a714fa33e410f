"""cfg 2 (ManyModes d=128, 65 536 x 256) with K = 8..64 components: step-loop kernel time.  SDENG_REF_MM=0: the vector path; SDENG_LIB: another library build."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sde_sampler_lrds_amd import _lib as L
if os.environ.get("SDENG_LIB"):  # A/B against another build of the library
    L.LIB_PATH = os.path.abspath(os.environ["SDENG_LIB"])
from sde_sampler_lrds_amd.experiments.baseline_configs import build_rds_gmm
dev = torch.device("cuda:0")
for K in (8, 16, 24, 32, 64):
    loss, ts, x0, args, _, info = build_rds_gmm(dev, 65536, 256, K=K)
    ev = L.HipEvents(); loss.timing_events = ev
    for rep in range(3):
        x, rnd, _ = loss.simulate(ts, x0, *args); torch.cuda.synchronize(); ms = ev.elapsed_ms()
    print(f"K={K}: kernel {ms:.2f} ms -> {65536*256/(ms*1e-3):.3e} p-steps/s finite {bool(torch.isfinite(rnd).all())} rnd mean {rnd.mean().item():.4f}")
