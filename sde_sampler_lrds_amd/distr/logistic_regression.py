"""Bayesian logistic regression posterior (mirror of ``sde_sampler/distr/logistic_regression.py:11-92``).

The reference reads ``data/<name>.pkl`` with ``pickle.load``; this mirror takes the design matrix directly
(``X_train [n, d-1]``, ``y_train [n]``) -- pickles are not loaded here -- and keeps the attribute names."""
from __future__ import annotations

import torch
from torch.distributions.utils import probs_to_logits
from torch.nn.functional import binary_cross_entropy_with_logits

from .base import Distribution


class LogisticRegression(Distribution):
    def __init__(self, X_train, y_train, use_intercept=True, intercept_mean=0.0, intercept_scale=2.5, weight_scale=1.0,
                 threshold=1e-8, dim=None, **kwargs):
        if not use_intercept:
            raise NotImplementedError("the engine covers use_intercept=True (all conf/target logreg configs)")
        super().__init__(dim=X_train.shape[-1] + 1, **kwargs)
        self.register_buffer("X_train", X_train.float(), persistent=False)
        self.register_buffer("y_train", y_train.float().flatten(), persistent=False)
        self.threshold = 1e-8  # the reference hard-codes 1e-8 regardless of the argument (:26)
        self.use_intercept = True
        self.register_buffer("weight_scale", torch.tensor(float(weight_scale)), persistent=False)
        self.register_buffer("intercept_mean", torch.tensor(float(intercept_mean)), persistent=False)
        self.register_buffer("intercept_scale", torch.tensor(float(intercept_scale)), persistent=False)

    def unnorm_log_prob(self, x, *args, **kwargs):
        params = x.reshape((-1, x.shape[-1]))
        w, c = params[..., :-1], params[..., -1]
        prior = torch.distributions.Normal(0.0, self.weight_scale).log_prob(w).sum(-1)
        prior = prior + torch.distributions.Normal(self.intercept_mean, self.intercept_scale).log_prob(c)
        probs = torch.special.expit(torch.matmul(self.X_train, w.T).T + c.unsqueeze(-1))
        probs = torch.clip(probs, self.threshold, 1.0 - self.threshold)
        logits = probs_to_logits(probs, is_binary=True)
        y = self.y_train.unsqueeze(0).expand((logits.shape[0], -1))
        return (-binary_cross_entropy_with_logits(logits, y, reduction="none").sum(dim=-1) + prior).unsqueeze(-1)
