"""Host-side profile of a log-variance training step (where do the ~5 ms go when the GPU work is ~1.5 ms?)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details  # noqa: E402

d, B, N = 128, 512, 100
FUSED = os.environ.get("PROBE_FUSED", "1") == "1"
tgt = make_target_details("many_modes", dim=d, n_modes=4)
g = torch.Generator().manual_seed(0)
model = make_model("vp-ref", "gmm", "lv", "ei", "base_zero_init", "uniform",
                   dict(means_ref=4 * torch.rand(4, d, generator=g) - 2, variances_ref=0.5 * torch.ones(4, d), weights_ref=torch.ones(4)),
                   tgt, dict(train_steps=10, train_batch_size=B, eval_batch_size=B), optim_details=dict(lr=1e-3), n_steps=N)
model.setup_optim()
model.loss.fused_training = FUSED
for i in range(5):
    model.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(40):
    model.step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
