"""Log-variance training step with a ScoreCtrl (DDS on TwoModes, PIS on PhiFour): wall time per step."""
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
    print(f"{label}: training step {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)


for method in ("lv", "kl"):
    tgt = make_target_details("two_modes", dim=2)
    bench(f"DDS TwoModes d=2 4096 x 64 [{method}]", make_model("dds_orig", "default", method, "em", "target_informed_zero_init", "uniform", dict(sigma=1.0), tgt,
          dict(train_steps=10, train_batch_size=4096, eval_batch_size=4096), optim_details=dict(lr=1e-3), n_steps=64))
    tgt = make_target_details("many_modes", dim=16, n_modes=4)
    bench(f"PIS ManyModes d=16 2048 x 100 [{method}]", make_model("pis_orig", "default", method, "em", "target_informed_zero_init", "uniform", dict(sigma=0.4472135954999579), tgt,
          dict(train_steps=10, train_batch_size=2048, eval_batch_size=2048), optim_details=dict(lr=1e-3), n_steps=100))
    tgt = make_target_details("phi_four", dim=100)
    bench(f"PIS PhiFour d=100 512 x 128 [{method}]", make_model("pis_orig", "default", method, "em", "target_informed_zero_init", "uniform", dict(sigma=0.4472135954999579), tgt,
          dict(train_steps=10, train_batch_size=512, eval_batch_size=512), optim_details=dict(lr=1e-3), n_steps=128))
for method in ("lv", "kl"):
    tgt = make_target_details("many_modes", dim=16, n_modes=4)
    bench(f"DIS (LerpCtrl) ManyModes d=16 2048 x 100 [{method}]", make_model("dis_orig", "default", method, "em", "target_informed_lerp_tempering", "uniform", dict(sigma=1.0), tgt,
          dict(train_steps=10, train_batch_size=2048, eval_batch_size=2048), optim_details=dict(lr=1e-3), n_steps=100))
    bench(f"CMCD ManyModes d=16 2048 x 100 [{method}]", make_model("cmcd", "default", method, "em", "target_informed_zero_init", "uniform", dict(), tgt,
          dict(train_steps=10, train_batch_size=2048, eval_batch_size=2048), optim_details=dict(lr=1e-3), n_steps=100))
