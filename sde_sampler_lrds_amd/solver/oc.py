"""Solver objects with the reference's names and evaluation surface (``sde_sampler/solver/oc.py``:
TrainableDiff :22, Bridge :185, CMCD :264, PIS :349, DDS :426, RDS :495), built from a plain dict instead of
Hydra (Hydra / OmegaConf are not installable here).  They wire prior / SDE / drift net / loss exactly like the
reference's ``setup_models`` and expose ``compute_results`` / ``evaluate`` (solver/oc.py:129-182,
solver/base.py:339-357), ``change_reference_type`` (:513-588), ``update_prior`` (:291-303), ``state_dict``.
Every ``simulate`` underneath is one HIP launch.  The training loop (``Trainable.step/run``,
solver/base.py:401-502) is the training direction and is not provided.
"""
from __future__ import annotations

import math
import time
from functools import partial

import torch

from ..distr.delta import Delta
from ..distr.gauss import Gauss, GaussFull, IsotropicGauss
from ..eq import sdes
from ..losses import oc as losses
from ..models.mlp import FourierMLP, TimeEmbed
from ..models.reparam import CancelDriftCtrl, ClippedCtrl, LerpCtrl, ScoreCtrl
from ..models import utils as mutils
from ..reference import MarginalReference
from ..utils.common import Results, clip_and_log, get_timesteps

LOSSES = {name: getattr(losses, name) for name in (
    "EMReferenceSDELoss", "EIReferenceSDELoss", "DDPMLikeReferenceSDELoss", "ControlledLangevinSDELoss",
    "DiscreteTimeReversalLossEI", "TimeReversalLoss", "ExponentialIntegratorSDELoss")}
SDES = {"VP": sdes.VP, "CosineVP": sdes.CosineVP, "ScaledBM": sdes.ScaledBM, "PinnedBM": sdes.PinnedBM,
        "ControlledLangevinSDE": sdes.ControlledLangevinSDE}


def build_ctrl(model: str, dim: int, sde, prior, target):
    """conf/model/{basic,score,lerp}.yaml + conf/model/base/{fouriermlp,time_embed}.yaml."""
    act = torch.nn.GELU()
    net = FourierMLP(dim=dim, activation=act, num_layers=4, channels=64, last_bias_init=mutils.init_bias_uniform_zeros,
                     last_weight_init=mutils.kaiming_uniform_zeros_)
    if model == "basic":
        return ClippedCtrl(base_model=net, clip_model=1e4)
    if model not in ("score", "lerp", "langevin_init"):
        raise NotImplementedError(f"model '{model}' (unet) has no HIP kernel")
    bias_init = mutils.init_bias_uniform_zeros if model == "score" else partial(mutils.init_bias_uniform_constant, val=1.0)
    sm = TimeEmbed(dim_out=1, activation=act, num_layers=4, channels=64, last_bias_init=bias_init,
                   last_weight_init=mutils.kaiming_uniform_zeros_)
    common = dict(base_model=net, score_model=sm, target_score=target.score, detach_score=False, clip_score=1e4,
                  clip_model=1e4, scale_score=1.0)
    if model == "score":
        return ScoreCtrl(**common)
    if model == "lerp":
        return LerpCtrl(**common, sde=sde, prior_score=prior.score)
    if model == "langevin_init":  # conf/model/langevin_init.yaml
        return CancelDriftCtrl(**common, sde=sde, langevin_init=True)
    raise NotImplementedError(f"model '{model}' (langevin_init / unet) has no HIP kernel")


class TrainableDiff:
    """Evaluation half of solver/oc.py:22-182 (+ the parts of solver/base.py it needs)."""

    def __init__(self, cfg: dict, target, device="cuda"):
        self.cfg, self.target = cfg, target
        self.device = torch.device(device)
        self.eval_batch_size = cfg.get("eval_batch_size", 6000)
        self.train_batch_size = cfg.get("train_batch_size", 512)
        self.clip_target = cfg.get("clip_target")
        ts_cfg = dict(cfg["timesteps"])
        self.train_timesteps = partial(get_timesteps, **ts_cfg)
        self.eval_timesteps = self.train_timesteps
        self.eval_ts = None
        self.use_ema = bool(cfg.get("use_ema", False))
        self.ema_steps, self.ema_decay = cfg.get("ema_steps", 10), cfg.get("ema_decay", 0.995)  # conf/solver/basic_oc_base.yaml:23-26
        # solver/oc.py:37 sets True for every diffusion solver and :356/:436 switch it off for PIS/DDS; here it is on
        # only where compute_eubo is a HIP launch (the reference-SDE losses of RDS)
        self.eubo_available = False
        self.seed = cfg.get("seed", 1)
        torch.manual_seed(self.seed)
        self.setup_models()
        self.to(self.device)

    # -- wiring ----------------------------------------------------------------------------
    def make_prior(self):
        p = self.cfg["prior"]
        if p["kind"] == "delta":
            return Delta(dim=self.target.dim)
        return IsotropicGauss(dim=self.target.dim, scale=p.get("scale", 1.0))

    def make_sde(self, **extra):
        c = dict(self.cfg["sde"])
        return SDES[c.pop("kind")](**c, **extra)

    def setup_models(self, langevin_based=False, skip_prior=False):
        if not skip_prior:
            self.prior = self.make_prior()
        if self.cfg.get("sde") is None:
            self.sde = None
        elif langevin_based:
            self.sde = self.make_sde(prior_score=self.prior.score, target_score=self.target.score)
        else:
            self.sde = self.make_sde()
        self.generative_ctrl = build_ctrl(self.cfg["model"], self.target.dim, self.sde, self.prior, self.target)
        if self.use_ema:  # solver/oc.py:69-78
            total = self.cfg.get("train_steps", 1) / (self.train_batch_size * self.ema_steps)
            alpha = min(1.0, (1.0 - self.ema_decay) / total)
            self.generative_ctrl_ema = torch.optim.swa_utils.AveragedModel(
                self.generative_ctrl, multi_avg_fn=torch.optim.swa_utils.get_ema_multi_avg_fn(1.0 - alpha))
        else:
            self.generative_ctrl_ema = self.generative_ctrl

    def make_loss(self, **extra):
        c = dict(self.cfg["loss"])
        cls = LOSSES[c.pop("kind")]
        loss = cls(self.generative_ctrl, self.generative_ctrl_ema, sde=self.sde,
                   filter_samples=getattr(self.target, "filter", None), **c, **extra)
        loss.seed = self.seed
        loss.graph_training = bool(self.cfg.get("graph_training", False))  # log-variance training: batched control pass as a hipGraph (works; measured no gain: the step is bound by its read-backs, tools/probe_train_host.py)
        loss.split_tiles = bool(self.cfg.get("split_tiles", True))  # evaluation batches of a few thousand particles: low-latency kernels
        return loss

    def modules(self):
        return [m for m in (self.target, self.prior, self.sde, self.generative_ctrl, getattr(self, "generative_ctrl_ema", None),
                            getattr(self, "_reference", None), getattr(self, "reference_distr", None)) if isinstance(m, torch.nn.Module)]

    def to(self, device):
        self.device = torch.device(device)
        for m in self.modules():
            m.to(self.device)
        return self

    # -- reference surface -------------------------------------------------------------------
    def clipped_target_unnorm_log_prob(self, x):
        return clip_and_log(self.target.unnorm_log_prob(x), max_norm=self.clip_target, name="target")

    def sample_prior(self, batch_size):
        """``self.prior.sample((batch_size,))`` of solver/oc.py:120,132.  ``native_prior`` (default on): x0 is left to the engine
         -- a pure function of (loss.seed, global particle index), drawn inside the step-loop kernel, so "identical seeds"
        covers x0 too and a sharded run draws the same particles as a single-GPU one.  Off: torch's global generator, like upstream."""
        from .. import engine as E
        if self.cfg.get("native_prior", True) and self.device.type == "cuda":
            try:
                draw = E.InitialDraw(self.prior, batch_size, self.device)
                draw.desc([])
                return draw
            except E.UnsupportedByEngine:
                pass
        return self.prior.sample((batch_size,)).to(self.device)

    def compute_results(self, use_ema=True) -> Results:
        """solver/oc.py:129-160: one pass with trajectories + weights, one timed pass without."""
        x = self.sample_prior(self.eval_batch_size)
        if self.eval_ts is None:
            self.eval_ts = self.eval_timesteps(device=self.device) if self._plain_grid() else self.eval_timesteps().to(self.device)
        ts = self.eval_ts
        results = self._compute_results(ts, x, use_ema=use_ema, compute_weights=True)
        assert results.xs.shape == (len(ts), *results.samples.shape)
        torch.cuda.synchronize(self.device)
        start = time.time()
        extra = self._compute_results(ts, x, use_ema=use_ema, compute_weights=False, return_traj=False)
        torch.cuda.synchronize(self.device)
        results.metrics["eval/sample_time"] = time.time() - start
        results.metrics.update(extra.metrics)
        results.log_norm_const_preds.update(extra.log_norm_const_preds)
        return results

    def _plain_grid(self):
        return "sde" not in self.eval_timesteps.keywords

    @torch.no_grad()
    def evaluate(self, use_ema=True) -> Results:
        return self.compute_results(use_ema=use_ema)

    def state_dict(self):
        return {"generative_ctrl": self.generative_ctrl.state_dict(), "loss": self.loss.state_dict()}

    def load_state_dict(self, sd):
        self.generative_ctrl.load_state_dict(sd["generative_ctrl"])
        if "loss" in sd:
            self.loss.load_state_dict(sd["loss"])

    # -- training (solver/base.py:265-310 optimiser set-up, :401-457 step; solver/oc.py:119-127 compute_loss) ----------------
    def trainable_parameters(self):
        return [p for p in self.generative_ctrl.parameters() if p.requires_grad]

    def setup_optim(self):
        o = dict(self.cfg.get("optim") or {})
        params = self.trainable_parameters()
        if "fused" not in o and "foreach" not in o and params and all(p.is_cuda for p in params):
            o["fused"] = True  # one kernel for the whole update (same arithmetic as the default foreach implementation)
        self.optim = torch.optim.Adam(params, **{"lr": 3e-4, **o})  # conf/solver/basic_oc_base.yaml:19-21
        self.train_steps = self.cfg.get("train_steps", 0)
        self.max_loss, self.max_grad, self.scale_loss = self.cfg.get("max_loss"), self.cfg.get("max_grad"), self.cfg.get("scale_loss")
        self.grad_clip_norm = self.cfg.get("grad_clip_norm")
        self.n_steps, self.n_steps_skip, self.train_ts = 0, 0, None

    def compute_loss(self):
        x = self.prior.sample((self.train_batch_size,)).to(self.device)
        if self.train_ts is None:
            self.train_ts = self.train_timesteps(device=self.device) if self._plain_grid() else self.train_timesteps().to(self.device)
        return self._compute_loss(self.train_ts, x)

    def _compute_loss(self, ts, x):
        raise NotImplementedError

    def step(self, step_id):
        """One stochastic gradient step (solver/base.py:401-457): the loss's forward simulation is a HIP launch, its gradient
        one batched autograd pass of the control (losses/oc.py ``__call__``)."""
        import time as _time
        if getattr(self, "optim", None) is None:
            self.setup_optim()
        t0 = _time.time()
        self.optim.zero_grad()
        loss, metrics = self.compute_loss()
        if self.scale_loss is not None:
            loss = self.scale_loss * loss
        loss.backward()
        params = self.trainable_parameters()
        loss_ok = bool(loss.isfinite()) if self.max_loss is None else bool(loss.abs() <= self.max_loss)
        # one read-back for all parameters (a per-parameter bool() costs a stream synchronisation each): max |grad| is NaN / inf
        # exactly when some gradient entry is (solver/base.py:425-433 checks every parameter)
        grads = [p.grad for p in params if p.grad is not None]
        mg = float(torch.stack(torch._foreach_norm(grads, float("inf"))).max()) if grads else 0.0  # (two kernels, not two per parameter)
        if self.max_grad is None:
            grad_ok = math.isfinite(mg)
        else:
            grad_ok = mg <= self.max_grad
            metrics["train/max_grad"] = mg
        if loss_ok and grad_ok:
            if self.grad_clip_norm is not None:
                metrics["train/grad_clip_norm"] = float(torch.nn.utils.clip_grad_norm_(params, self.grad_clip_norm))
            self.optim.step()
            if self.use_ema and step_id % self.ema_steps == 0:
                self.generative_ctrl_ema.update_parameters(self.generative_ctrl)
        else:
            self.n_steps_skip += 1
        metrics.update({"train/time_per_step": _time.time() - t0, "train/loss": loss.item(), "train/skipped_steps": self.n_steps_skip,
                        "train/no_grad": sum(p.grad is None for p in params)})
        self.n_steps += 1
        return metrics

    def run(self):
        if getattr(self, "optim", None) is None:
            self.setup_optim()
        for step_id in range(self.n_steps, self.train_steps):
            self.step(step_id)
        return self.evaluate(use_ema=self.use_ema)


class _InitialLogProbSolver(TrainableDiff):
    def _compute_loss(self, ts, x):  # solver/oc.py:220-234
        return self.loss(ts, x, self.clipped_target_unnorm_log_prob, initial_log_prob=self.prior.log_prob)

    def _compute_results(self, ts, x, use_ema=True, compute_weights=True, return_traj=True):
        return self.loss.eval(ts, x, self.clipped_target_unnorm_log_prob, use_ema=use_ema,
                              initial_log_prob=self.prior.log_prob, compute_weights=compute_weights, return_traj=return_traj)


class Bridge(_InitialLogProbSolver):
    """DIS (solver/oc.py:185-261, without a learned inference control)."""

    def setup_models(self):
        super().setup_models()
        if not isinstance(self.prior, Gauss):
            raise ValueError("Can only be used with Gaussian prior.")
        self.loss = self.make_loss(**({"inference_ctrl": None} if self.cfg["loss"]["kind"] == "TimeReversalLoss" else {}))
        self.eubo_available = self.cfg["loss"]["kind"] == "DiscreteTimeReversalLossEI"  # the loss whose compute_eubo is a HIP launch


class CMCD(_InitialLogProbSolver):
    """solver/oc.py:264-346."""

    def setup_models(self, skip_prior=False):
        super().setup_models(langevin_based=True, skip_prior=skip_prior)
        if not isinstance(self.prior, (Gauss, GaussFull)):
            raise ValueError("Can only be used with gaussian prior.")
        self.loss = self.make_loss()
        self.eubo_available = hasattr(self.target, "loc") and hasattr(self.target, "sample")  # mixture targets: HIP noising loop

    def update_prior(self, mean, var):
        dim = mean.shape[0]
        self.prior = GaussFull(dim=dim, loc=mean, cov=var) if var.dim() == 2 else Gauss(dim=dim, loc=mean, scale=var.sqrt())
        self.setup_models(skip_prior=True)
        self.to(self.device)


class _ReferenceLogProbSolver(TrainableDiff):
    def _compute_loss(self, ts, x):  # solver/oc.py:380-396, 453-465, 593-605
        return self.loss(ts, x, self.clipped_target_unnorm_log_prob, self.reference_distr.log_prob)

    def _compute_results(self, ts, x, use_ema=True, compute_weights=True, return_traj=True):
        return self.loss.eval(ts, x, self.clipped_target_unnorm_log_prob, self.reference_distr.log_prob, use_ema=use_ema,
                              compute_weights=compute_weights, return_traj=return_traj)


class PIS(_ReferenceLogProbSolver):
    """solver/oc.py:349-423."""

    def setup_models(self):
        super().setup_models()
        if not isinstance(self.prior, Delta):
            raise ValueError("Can only be used with dirac delta prior.")
        self.reference_distr = self.sde.marginal_distr(t=self.sde.terminal_t, x_init=self.prior.loc)
        self.loss = self.make_loss()


class DDS(_ReferenceLogProbSolver):
    """solver/oc.py:426-492."""

    def setup_models(self):
        super().setup_models()
        if not isinstance(self.prior, Gauss):
            raise ValueError("Can only be used with Gaussian prior.")
        self.reference_distr = self.prior
        self.loss = self.make_loss()


class RDS(_ReferenceLogProbSolver):
    """solver/oc.py:495-666."""

    def setup_models(self):
        super().setup_models()
        self.change_reference_type(ref_type="default")
        self.loss = self.make_loss(reference_ctrl=self.reference_ctrl)
        self.eubo_available = True

    def change_reference_type(self, ref_type="default", net=None, eps=None, mean=None, var=None, means=None,
                              variances=None, weights=None):
        if ref_type == "default":
            if isinstance(self.sde, sdes.VP):
                utils = dict(x_init=self.prior.loc.flatten(), var_init=torch.square(self.prior.scale).flatten())
            elif isinstance(self.sde, sdes.PinnedBM):
                utils = dict(x_init=self.prior.loc.flatten(),
                             var_init=(self.sde.terminal_t * self.sde.diff_coeff ** 2 * torch.ones_like(self.prior.loc)).flatten())
            else:
                raise ValueError(f"Default reference for SDE type {type(self.sde)} is not supported.")
            self._reference = MarginalReference(self.sde, "default", **utils)
        elif ref_type == "gaussian":
            self._reference = MarginalReference(self.sde, "gaussian", x_init=mean, var_init=var)
        elif ref_type == "gmm":
            self._reference = MarginalReference(self.sde, "gmm", means_init=means, variances_init=variances, weights_init=weights)
        else:
            raise NotImplementedError(f"Reference type {ref_type}: EBM references need autograd per step (no HIP kernel).")
        dev = getattr(self, "device", None)
        if dev is not None:
            self._reference.to(dev)
        self.ref_type = ref_type
        self.reference_distr_utils = self._reference.reference_distr_utils
        self.reference_distr = self._reference.reference_distr
        self.reference_score_t = self._reference
        if hasattr(self, "loss"):
            self.loss.reference_ctrl = self.reference_ctrl

    def reference_ctrl(self, t, x):
        return self.reference_score_t(t, x)

    def state_dict(self):
        sd = super().state_dict()
        sd.update({f"ref_{k}": v for k, v in self.reference_distr_utils.items()})
        sd["ref_type"] = self.ref_type
        return sd
