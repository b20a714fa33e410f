"""Where does a log-variance training step spend its time? (make_model vp-ref / gmm reference, ManyModes)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_lrds_amd import engine as E  # noqa: E402
from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details  # noqa: E402

for d, B, N in ((16, 512, 100), (128, 512, 100), (128, 2048, 256)):
    tgt = make_target_details("many_modes", dim=d, n_modes=4)
    g = torch.Generator().manual_seed(0)
    model = make_model("vp-ref", "gmm", "lv", "ei", "base_zero_init", "uniform",
                       dict(means_ref=4 * torch.rand(4, d, generator=g) - 2, variances_ref=0.5 * torch.ones(4, d), weights_ref=torch.ones(4)),
                       tgt, dict(train_steps=10, train_batch_size=B, eval_batch_size=B), optim_details=dict(lr=1e-3), n_steps=N)
    model.setup_optim()
    per_mode = {}
    for fused in (False, True):  # the batched control pass as eager torch autograd, then as the fused HIP forward + backward
        model.loss.fused_training = fused
        for i in range(5):
            model.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            model.step(i)
        torch.cuda.synchronize()
        per_mode[fused] = (time.perf_counter() - t0) / 20
    per = per_mode[True]
    print(f"d={d} B={B} N={N}: training step {per_mode[False]*1e3:.2f} ms with the eager batched pass, {per_mode[True]*1e3:.2f} ms with sdeng_ctrl_vjp", flush=True)
    x = model.prior.sample((B,)).to(model.device)
    ts = model.train_ts
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        z = E.philox_noise(1, N, B, d, 0, x.device)
    torch.cuda.synchronize(); t_noise = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        with torch.no_grad():
            model.loss.simulate(ts, x, model.clipped_target_unnorm_log_prob, model.reference_distr.log_prob, return_traj=True, noise=z)
    torch.cuda.synchronize(); t_sim = (time.perf_counter() - t0) / 20
    print(f"d={d} B={B} N={N}: step {per*1e3:.2f} ms  (noise generation {t_noise*1e3:.2f} ms, HIP simulate with trajectory {t_sim*1e3:.2f} ms, "
          f"rest = batched autograd pass + optimiser {max(per - t_noise - t_sim, 0)*1e3:.2f} ms) -> {B*N/per:.3e} particle-steps/s in training", flush=True)
