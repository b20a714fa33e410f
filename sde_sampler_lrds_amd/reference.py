"""Reference drifts of RDS (solver/oc.py:513-592 ``change_reference_type`` + ``reference_ctrl``): the
score of a Gaussian / Gaussian-mixture reference noised to time t (eq/sdes.py:265-279, 329-345).

``MarginalReference`` is a callable ``(t, x) -> [B,d]`` (torch, host-side use) that also exposes
``reference_distr_utils`` / ``ref_type`` exactly like the reference's RDS solver does, which is what the HIP
engine's descriptor compiler reads (engine.resolve_reference)."""
from __future__ import annotations

import torch

from .distr.gauss import score_gauss, score_gauss_full, score_mog, score_mog_full


class MarginalReference(torch.nn.Module):
    def __init__(self, sde, ref_type: str, **utils):
        super().__init__()
        assert ref_type in ("gaussian", "gmm", "default")
        self.sde, self.ref_type = sde, ref_type
        self.reference_distr_utils = {k: (tuple(a.float() for a in v) if isinstance(v, tuple) else v.float()) for k, v in utils.items()}

    def to(self, *a, **k):
        self.reference_distr_utils = {key: (tuple(e.to(*a, **k) for e in v) if isinstance(v, tuple) else v.to(*a, **k))
                                      for key, v in self.reference_distr_utils.items()}
        return super().to(*a, **k)

    @property
    def reference_distr(self):
        u, t0 = self.reference_distr_utils, torch.tensor(0.0, device=self.sde.terminal_t.device)
        if "means_init" in u:
            return self.sde.marginal_gmm_distr(t0, u["means_init"], u["variances_init"], u["weights_init"])
        return self.sde.marginal_distr(t0, u["x_init"], u["var_init"])

    def forward(self, t, x):
        u = self.reference_distr_utils
        if "means_init" in u:
            loc, var = self.sde.marginal_params(t, u["means_init"], var_init=u["variances_init"], is_mixture=True)
            if isinstance(var, tuple):  # eq/sdes.py:341-342
                return score_mog_full(x, u["weights_init"], loc, None, precisions=var[0], covariances_log_det=var[1])
            if var.dim() == 3:
                return score_mog_full(x, u["weights_init"], loc, var)
            return score_mog(x, u["weights_init"], loc, var)
        loc, var = self.sde.marginal_params(t, u["x_init"], var_init=u["var_init"])
        if isinstance(var, tuple):  # eq/sdes.py:274-277
            return score_gauss_full(x, loc, None, precisions=var[0])
        if var.dim() == 2:
            return score_gauss_full(x, loc, var)
        return score_gauss(x, loc, var)
