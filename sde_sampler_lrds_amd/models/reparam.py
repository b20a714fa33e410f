"""Control wrappers (mirror of ``sde_sampler/models/reparam.py``: ClippedCtrl :18-43, ScoreCtrl :63-117,
LerpCtrl :148-199).  See models/mlp.py for how these relate to the HIP path."""
from __future__ import annotations

from typing import Callable

import torch
from torch.nn import Module


def _clip(v, m):
    return v if m is None else v.clip(min=-1.0 * m, max=m)


class ClippedCtrl(Module):
    def __init__(self, base_model: Module, clip_model: float | None = None, name: str = "ctrl", **kwargs):
        super().__init__()
        self.base_model, self.clip_model, self.name = base_model, clip_model, name

    def clipped_base_model(self, t, x):
        return _clip(self.base_model(t, x), self.clip_model)

    def forward(self, t, x):
        return self.clipped_base_model(t, x)


class ScoreCtrl(ClippedCtrl):
    """base_model(t,x) + scale * clip(target_score(x)) * clip(score_model(t))."""

    def __init__(self, *args, target_score: Callable, score_model: Module | None = None, detach_score: bool = True,
                 scale_score: float = 1.0, clip_score: float | None = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.score_model, self.target_score = score_model, target_score
        self.detach_score, self.scale_score, self.clip_score = detach_score, scale_score, clip_score

    def clipped_target_score(self, t, x):
        x = x.detach() if self.detach_score else x
        return _clip(self.target_score(x, create_graph=self.detach_score), self.clip_score)

    def clipped_score_model(self, t, x):
        return _clip(self.score_model(t, x), self.clip_model)

    def forward(self, t, x):
        score = self.scale_score * self.clipped_target_score(t, x)
        if self.score_model is not None:
            score = score * self.clipped_score_model(t, x)
        return self.clipped_base_model(t, x) + score


class LerpCtrl(ScoreCtrl):
    """base_model + g(t) * scale * clip(lerp(prior_score, target_score, t/T)) * clip(score_model(t))."""

    def __init__(self, *args, sde, prior_score: Callable, hard_constrain: bool = False, scale_lerp: float = 1.0, **kwargs):
        super().__init__(*args, **kwargs)
        if hard_constrain:
            raise NotImplementedError("hard_constrain is not part of the engine (conf/model/lerp.yaml leaves it off)")
        self.sde, self.prior_score, self.hard_constrain, self.scale_lerp = sde, prior_score, hard_constrain, scale_lerp

    def clipped_interpolated_score(self, t, x):
        x = x.detach() if self.detach_score else x
        mix = torch.lerp(self.prior_score(x), self.target_score(x, create_graph=self.detach_score), t / self.sde.terminal_t)
        return _clip(mix, self.clip_score)

    def forward(self, t, x):
        score = self.scale_score * self.clipped_interpolated_score(t, x)
        if self.score_model is not None:
            score = score * self.clipped_score_model(t, x)
        return self.clipped_base_model(t, x) + self.sde.diff(t, x) * score


class CancelDriftCtrl(ScoreCtrl):
    """Langevin initialisation (models/reparam.py:120-145): a ScoreCtrl that also subtracts the denoising SDE's drift,
    base_model + drift(t,x)/g(t) + g(t)/2 * scale * clip(target_score) * clip(score_model(t))  (``use_rescaling``; otherwise
    drift/g^2 + score/2)."""

    def __init__(self, *args, sde, langevin_init: bool = True, use_rescaling: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        if sde is None or getattr(sde, "noise_type", None) not in ("diagonal", "scalar"):  # reparam.py:125-126 (DDS has no SDE)
            raise ValueError(f"Invalid sde for CancelDriftCtrl: {sde!r}")
        self.sde, self.langevin_init, self.use_rescaling = sde, langevin_init, use_rescaling

    def forward(self, t, x):
        ctrl = self.clipped_base_model(t, x)
        g, f = self.sde.diff(t, x), self.sde.drift(t, x)
        score = self.scale_score * self.clipped_target_score(t, x)
        if self.score_model is not None:
            score = score * self.clipped_score_model(t, x)
        if self.use_rescaling:
            return ctrl + (f / g) + 0.5 * g * score
        return ctrl + (f / torch.square(g)) + 0.5 * score

