"""Step-loop time with a full-covariance mixture reference (cfg-2 shape)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import _lib as L  # noqa: E402
if os.environ.get("SDENG_LIB"):  # A/B against another build of the library
    L.LIB_PATH = os.path.abspath(os.environ["SDENG_LIB"])
from sde_sampler_lrds_amd.distr.gauss import ManyModes  # noqa: E402
from sde_sampler_lrds_amd.eq.sdes import VP  # noqa: E402
from sde_sampler_lrds_amd.experiments.baseline_configs import _net  # noqa: E402
from sde_sampler_lrds_amd.losses import oc  # noqa: E402
from sde_sampler_lrds_amd.models.reparam import ClippedCtrl  # noqa: E402
from sde_sampler_lrds_amd.reference import MarginalReference  # noqa: E402

dev = torch.device("cuda:0")
B, N = 65536, 256
for d, K in ((128, 4), (64, 4), (16, 4)):
    torch.manual_seed(1)
    sde = VP(0.1, 10.0, 1.0, terminal_t=1.0)
    target = ManyModes(n_modes=K, dim=d, var=0.5, seed_loc=42, mixture_weight_factor=3.0, n_reference_samples=10)
    ctrl = ClippedCtrl(base_model=_net(d), clip_model=1e4)
    A = torch.randn(K, d, d) / d ** 0.5
    ref = MarginalReference(sde, "gmm", means_init=target.loc.clone(), variances_init=0.3 * A @ A.transpose(-1, -2) + 0.4 * torch.eye(d),
                            weights_init=torch.ones(K))
    for m in (sde, target, ctrl, ref):
        m.to(dev)
    loss = oc.EIReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", reference_ctrl=ref)
    ts = torch.linspace(0.0, 1.0, N + 1, device=dev)
    x0 = torch.randn(B, d, device=dev)
    ev = L.HipEvents()
    loss.timing_events = ev
    for _ in range(3):
        x, rnd, _ = loss.simulate(ts, x0, target.unnorm_log_prob, ref.reference_distr.log_prob)
        torch.cuda.synchronize()
        ms = ev.elapsed_ms()
    print(f"full-covariance reference d={d} K={K}: kernel {ms:.2f} ms -> {B * N / (ms * 1e-3):.3e} particle-steps/s, finite {bool(torch.isfinite(rnd).all())}", flush=True)
