"""20 log-variance training steps at the reference's default size (512 x 100, d = 128) for a rocprofv3 --kernel-trace --stats run:
which GPU kernels make up a step (HIP trajectory, fused forward + backward of the drift net, the gradient GEMMs, Adam)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details  # noqa: E402

d, B, N = 128, int(os.environ.get("PROBE_B", 512)), int(os.environ.get("PROBE_N", 100))
tgt = make_target_details("many_modes", dim=d, n_modes=4)
g = torch.Generator().manual_seed(0)
model = make_model("vp-ref", "gmm", "lv", "ei", "base_zero_init", "uniform",
                   dict(means_ref=4 * torch.rand(4, d, generator=g) - 2, variances_ref=0.5 * torch.ones(4, d), weights_ref=torch.ones(4)),
                   tgt, dict(train_steps=10, train_batch_size=B, eval_batch_size=B), optim_details=dict(lr=1e-3), n_steps=N)
model.setup_optim()
model.loss.fused_training = os.environ.get("PROBE_FUSED", "1") == "1"
for i in range(25):
    model.step(i)
torch.cuda.synchronize()
print("done")
