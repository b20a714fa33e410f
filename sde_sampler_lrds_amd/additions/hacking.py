"""Evaluation wrapper of the experiment notebooks (mirror of ``sde_sampler/additions/hacking.py:14-102``): the sampling
metrics of ``Trainable.evaluate`` plus the EUBO-side metrics obtained from noising trajectories started at target
samples (``loss.compute_eubo``).  Both passes are HIP launches; ``run`` trains first (log-variance losses: HIP step loop
+ one batched autograd pass of the control per step)."""
from __future__ import annotations

import math

import torch



def evaluate_eubo(trainable, results, compute_eubo_last_arg, use_ema):
    """additions/hacking.py:14-33."""
    samples = trainable.target.sample((trainable.eval_batch_size,)).to(trainable.device)
    with torch.no_grad():
        rnd_target = trainable.loss.compute_eubo(trainable.eval_ts.to(trainable.device), samples,
                                                 trainable.clipped_target_unnorm_log_prob, compute_eubo_last_arg, use_ema=use_ema)
    neg = -rnd_target
    weights = torch.nn.functional.softmax(neg, dim=0)
    results.metrics["eval/log_norm_const_is_f"] = -rnd_target.logsumexp(dim=0).item() + math.log(len(weights))
    results.metrics["eval/eubo"] = neg.mean().item()
    results.metrics["eval/effective_sample_size_f"] = (1.0 / (weights ** 2).sum()).item()
    results.metrics["eval/norm_effective_sample_size_f"] = results.metrics["eval/effective_sample_size_f"] / len(weights)
    return results


class TrainableWrapper:
    """additions/hacking.py:36-102 (evaluation side)."""

    def __init__(self, trainable, verbose=True):
        self.trainable = trainable
        self.verbose = verbose

    def run(self, keep_training_metrics=False):
        """additions/hacking.py:43-66: train for the remaining steps (log-variance training runs on the HIP step loop; KL
        training raises), then evaluate with the EUBO-side metrics."""
        import time
        t = self.trainable
        if getattr(t, "optim", None) is None:
            t.setup_optim()
        training_metrics, training_time = [], 0.0
        for i in range(t.n_steps, t.train_steps):
            t0 = time.time()
            metrics = t.step(i)
            training_time += time.time() - t0
            if keep_training_metrics:
                training_metrics.append(metrics)
        results = self.evaluate(use_ema=getattr(t, "use_ema", False))
        results.metrics["eval/training_time"] = training_time
        if keep_training_metrics:
            return results, ({k: [m[k] for m in training_metrics if k in m] for k in training_metrics[0]} if training_metrics else {})
        return results

    def compute_results_eubo(self, results, use_ema=True):
        t = self.trainable
        if hasattr(t.loss, "compute_eubo") and getattr(t, "eubo_available", False) and hasattr(t.target, "sample"):
            last = t.reference_distr.log_prob if hasattr(t, "reference_distr") else t.prior.log_prob
            results = evaluate_eubo(t, results, last, use_ema=use_ema)
        return results

    @torch.no_grad()
    def evaluate(self, use_ema=True, log=True):
        t = self.trainable
        use_ema_ = getattr(t, "use_ema", False) and use_ema
        results = t.compute_results(use_ema=use_ema_)
        return self.compute_results_eubo(results, use_ema=use_ema_)
