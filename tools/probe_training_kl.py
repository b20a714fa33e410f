"""KL-method training step (back-propagation through the trajectory as a discrete adjoint, losses/oc.py _kl_loss): wall time per step and
the split between the HIP trajectory and the adjoint recursion.  Workloads: BASELINE cfg 1 (DDS, TwoModes d=2, 4096 x 64) and an RDS-EI
model at d=128."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details  # noqa: E402


def bench(label, model, n=10):
    model.setup_optim()
    for i in range(3):
        model.step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        model.step(i)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / n
    print(f"{label}: KL training step {per * 1e3:.2f} ms", flush=True)
    if os.environ.get("PROBE_PROFILE"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(5):
            model.step(i)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(30)


tgt = make_target_details("two_modes", dim=2)
bench("cfg 1: DDS, TwoModes d=2, 4096 x 64", make_model("dds_orig", "default", "kl", "em", "target_informed_zero_init", "uniform", dict(sigma=1.0), tgt,
                                                        dict(train_steps=10, train_batch_size=4096, eval_batch_size=4096), optim_details=dict(lr=1e-3), n_steps=64))
for d, B, N in ((16, 512, 100), (128, 512, 100)):
    tgt = make_target_details("many_modes", dim=d, n_modes=4)
    g = torch.Generator().manual_seed(0)
    bench(f"RDS-EI gmm-ref d={d}, {B} x {N}", make_model("vp-ref", "gmm", "kl", "ei", "base_zero_init", "uniform",
          dict(means_ref=4 * torch.rand(4, d, generator=g) - 2, variances_ref=0.5 * torch.ones(4, d), weights_ref=torch.ones(4)),
          tgt, dict(train_steps=10, train_batch_size=B, eval_batch_size=B), optim_details=dict(lr=1e-3), n_steps=N))
