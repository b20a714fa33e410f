"""CMCD step-loop time vs dimension (mixture target, IsotropicGauss prior): d <= 64 kernels fit the register file,
the d = 128 instantiation spills."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details  # noqa: E402

B, N = 65536, 256
for d in (16, 64, 100, 128):
    tgt = make_target_details("many_modes", dim=d, n_modes=4)
    model = make_model("cmcd", "default", "lv", "em", "target_informed_zero_init", "uniform", dict(), tgt,
                       dict(train_steps=0, train_batch_size=512, eval_batch_size=B), n_steps=N)
    x = model.prior.sample((B,)).to(model.device)
    ts = torch.linspace(0.0, 1.0, N + 1, device=model.device)
    ev = L.HipEvents()
    model.loss.timing_events = ev
    for _ in range(3):
        out = model.loss.simulate(ts, x, model.clipped_target_unnorm_log_prob, initial_log_prob=model.prior.log_prob, train=False)
        torch.cuda.synchronize()
        ms = ev.elapsed_ms()
    print(f"cmcd d={d}: kernel {ms:.2f} ms -> {B * N / (ms * 1e-3):.3e} particle-steps/s, finite {bool(torch.isfinite(out[1]).all())}", flush=True)
