"""EulerIntegrator kernel (k_euler): time and achieved HBM store bandwidth with the trajectory written (N+1 states)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
from sde_sampler_lrds_amd import engine as E  # noqa: E402
from sde_sampler_lrds_amd.distr.gauss import ManyModes  # noqa: E402
from sde_sampler_lrds_amd.eq.sdes import VP, LangevinSDE  # noqa: E402

dev = torch.device("cuda:0")
for d, B, N in ((128, 65536, 64), (16, 262144, 64), (2, 1048576, 32)):
    x = torch.randn(B, d, device=dev)
    ts = torch.linspace(0.0, 1.0, N + 1, device=dev)
    tgt = ManyModes(n_modes=4, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0, n_reference_samples=10).to(dev)
    for name, sde in (("OU (VP)", VP(0.1, 10.0, 1.0, terminal_t=1.0).to(dev)), ("Langevin, 4-mode mixture", LangevinSDE(tgt.score, 1.0, 100.0).to(dev))):
        ev = L.HipEvents()
        for _ in range(3):
            xs = E.euler_states(sde, ts, x, seed=3, events=ev)
            torch.cuda.synchronize()
            ms = ev.elapsed_ms()
        gb = (N + 2) * B * d * 4 / 1e9
        print(f"{name:26s} d={d:3d} B={B:7d} N={N}: {ms:6.2f} ms, {gb:.2f} GB stored+loaded -> {gb / (ms * 1e-3) / 1e3:.2f} TB/s, "
              f"{B * N / (ms * 1e-3):.3e} particle-steps/s", flush=True)
        del xs
