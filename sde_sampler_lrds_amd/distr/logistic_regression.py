"""Bayesian logistic regression posterior (mirror of ``sde_sampler/distr/logistic_regression.py:11-92``).

Constructor as upstream -- ``LogisticRegression(dim, data_type, use_intercept=True, intercept_mean, intercept_scale, weight_scale,
threshold)`` (:14) -- so ``conf/target/{sonar,ionosphere,cancer,credit}.yaml`` and ``make_model`` build it unchanged.  Upstream reads
``data/<data_type>.pkl`` with ``pickle.load`` (:16-17); a pickle executes code when loaded, so this mirror never opens one.  The
design matrix comes from, in this order:

* the ``X_train`` / ``y_train`` keywords (or the legacy positional form ``LogisticRegression(X_train, y_train, ...)``),
* a data set registered in-process with ``register_dataset(name, X_train, y_train)``,
* ``<DATA_DIR>/<data_type>.pt`` -- a plain tensor file ``{"X_train": [n, d-1], "y_train": [n]}`` written by the USER from their copy of
  the data (``torch.save`` of tensors; it is read with ``torch.load(..., weights_only=True)``), ``DATA_DIR`` = ``$SDENG_DATA_DIR`` or
  ``./data``.

Nothing else differs: same buffers (``X_train``, ``y_train``, scales), same log-density (:41-61)."""
from __future__ import annotations

import os

import torch
from torch.distributions.utils import probs_to_logits
from torch.nn.functional import binary_cross_entropy_with_logits

from .base import Distribution

_REGISTRY: dict[str, tuple[torch.Tensor, torch.Tensor]] = {}


def register_dataset(name: str, X_train: torch.Tensor, y_train: torch.Tensor) -> None:
    """Make ``LogisticRegression(dim, name)`` find this design matrix (``X_train [n, d-1]``, ``y_train [n]`` in {0, 1})."""
    _REGISTRY[name] = (X_train.detach().float().cpu().clone(), y_train.detach().float().flatten().cpu().clone())


def _load_dataset(name: str):
    if name in _REGISTRY:
        return _REGISTRY[name]
    path = os.path.join(os.environ.get("SDENG_DATA_DIR", "data"), f"{name}.pt")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"no design matrix for data_type '{name}': pass X_train= / y_train=, call register_dataset('{name}', X, y), or write "
            f"{path} = torch.save({{'X_train': X, 'y_train': y}}) from your copy of the data (the reference's data/{name}.pkl is a "
            "pickle and is deliberately not loaded)")
    blob = torch.load(path, map_location="cpu", weights_only=True)
    return blob["X_train"].float(), blob["y_train"].float().flatten()


class LogisticRegression(Distribution):
    def __init__(self, dim=None, data_type=None, use_intercept=True, intercept_mean=0.0, intercept_scale=2.5, weight_scale=1.0,
                 threshold=1e-8, X_train=None, y_train=None, **kwargs):
        if torch.is_tensor(dim):  # legacy positional form: LogisticRegression(X_train, y_train, ...)
            X_train, y_train, dim, data_type = dim, data_type, None, None
        if X_train is None:
            if data_type is None:
                raise ValueError("LogisticRegression needs data_type (a registered / saved data set) or X_train and y_train")
            X_train, y_train = _load_dataset(data_type)
        if not use_intercept:
            raise NotImplementedError("the engine covers use_intercept=True (all conf/target logreg configs)")
        super().__init__(dim=X_train.shape[-1] + 1, **kwargs)  # upstream ignores its `dim` argument too (:23)
        if dim is not None and int(dim) != self.dim:
            raise ValueError(f"dim={dim} but the design matrix has {X_train.shape[-1]} features (+1 intercept)")
        self.data_type = data_type
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
