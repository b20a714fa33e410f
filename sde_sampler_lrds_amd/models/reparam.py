"""Control wrappers (mirror of ``sde_sampler/models/reparam.py``: ClippedCtrl :18-43, RemoveReferenceCtrl :46-64, ScoreCtrl :67-117, CancelDriftCtrl :120-145,
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


class RemoveReferenceCtrl(Module):
    """models/reparam.py:46-64: ``score(t, x) - ref_score(t, x)`` (``use_rescaling=False``) or ``- sde.diff(t, x) * ref_score(t, x)``
    (``use_rescaling=True``; upstream's constructor then forbids passing the ``sde`` its forward needs, so that branch raises
    AttributeError when called -- kept).  "Only used for Langevin init", i.e. around a CancelDriftCtrl.  The step-loop kernel runs the
    ``use_rescaling=False`` form as a control modifier (SDENG_FLAG_REMOVE_REF: u -= reference score, which the tail already holds)."""

    def __init__(self, score, ref_score, use_rescaling=True, sde=None):
        super().__init__()
        assert not (use_rescaling and (sde is not None))
        self.score, self.ref_score, self.use_rescaling, self.sde = score, ref_score, use_rescaling, sde

    def forward(self, t, x):
        ret = self.score(t, x)
        if self.use_rescaling:
            ret -= self.sde.diff(t, x) * self.ref_score(t, x)
        else:
            ret -= self.ref_score(t, x)
        return ret


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

