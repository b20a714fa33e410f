"""Descriptor compiler + launcher: turns the Python objects the reference's loss classes are handed
(drift-net module, SDE, reference, target / prior distributions, time grid) into one flat
``sdeng_desc`` (include/sdeng.h) and calls ``sdeng_simulate`` on the current HIP stream.

Objects are recognised by duck typing on class name + attributes, so both the reference's own classes
(``sde_sampler.eq.sdes.VP`` ...) and this package's mirrors are accepted.  Anything not recognised raises
``UnsupportedByEngine`` -- there is no PyTorch re-implementation of the step loop to fall back on.

Per-step scalars are computed on the host with the same 0-d fp32 torch expressions the
reference evaluates inside its loop, so the coefficients the kernel consumes are bit-identical to the
reference's; the table is cached per (loss, time grid).
"""
from __future__ import annotations

import copy
import ctypes as C
import math
import weakref

import torch

from . import _lib as L


class UnsupportedByEngine(NotImplementedError):
    pass


def _name(obj):
    return type(obj).__name__


def _dev_f32(t: torch.Tensor, device, keep: list):
    t = t.detach().to(device=device, dtype=torch.float32).contiguous()
    keep.append(t)
    return t.data_ptr()


def _self_of(fn):
    return getattr(fn, "__self__", None)


# ------------------------------------------------------------------------------------------------
# distributions
# ------------------------------------------------------------------------------------------------
def dist_desc(obj, device, keep, clip=None, score_only=False) -> L.Dist:
    """Distribution object -> sdeng_dist (distr/*.py parameters).  ``score_only``: the descriptor is the target of a Score / Lerp /
    CancelDrift control and only its score is evaluated -- the one use a full-covariance mixture has a kernel for."""
    ds = L.Dist()
    ds.clip = float(clip) if clip else 0.0
    if obj is None:
        ds.kind = L.DIST_NONE
        return ds
    n = _name(obj)
    if n == "IsotropicGauss":
        # the four scalars are read back from the device once per parameter version (a read-back synchronises the stream)
        key = (obj.loc._version, obj.scale._version, obj.loc.data_ptr(), obj.scale.data_ptr())
        hit = getattr(obj, "_sdeng_scalars", None)
        if hit is None or hit[0] != key:
            loc, scale = obj.loc[0, 0].detach().float().cpu(), obj.scale[0, 0].detach().float().cpu()
            var = scale ** 2
            hit = (key, (float(loc), float(scale), float(-0.5 * obj.dim * (2.0 * math.pi * var).log()), float(var)))  # distr/gauss.py:759
            obj._sdeng_scalars = hit
        ds.kind = L.DIST_ISO_GAUSS
        ds.p0, ds.p1, ds.p2, ds.p3 = hit[1]
        return ds
    if n in ("GMM", "TwoModes", "ManyModes", "BracketTwoModes", "Gauss", "Delta"):
        if getattr(obj, "mixture_weights", None) is None:
            ds.kind = L.DIST_GAUSS_DIAG
            ds.k = 1
            ds.loc = _dev_f32(obj.loc.reshape(-1), device, keep)
            ds.scale = _dev_f32(obj.scale.reshape(-1), device, keep)
        else:
            ds.kind = L.DIST_GMM_DIAG
            ds.k = int(obj.loc.shape[0])
            ds.loc = _dev_f32(obj.loc, device, keep)
            ds.scale = _dev_f32(obj.scale, device, keep)
            ds.w = _dev_f32(obj.mixture_weights, device, keep)
        return ds
    if n == "PhiFour":
        if getattr(obj, "dim_phys", 1) != 1 or tuple(getattr(obj, "bc", ("dirichlet", 0))) != ("dirichlet", 0) or getattr(obj, "tilt", None):
            raise UnsupportedByEngine("PhiFour: only the 1-D Dirichlet-0 untilted lattice")
        ds.kind = L.DIST_PHI4
        ds.p0, ds.p1, ds.p2 = scalar_of(obj.a), scalar_of(obj.b), scalar_of(obj.beta)
        return ds
    if n == "GaussFull":
        key = (obj.cov._version, obj.cov.data_ptr(), str(device))
        hit = getattr(obj, "_sdeng_chol", None)
        if hit is None or hit[0] != key:  # Cholesky factor, its inverse and log-determinant once per parameter version
            tril = torch.linalg.cholesky(obj.cov.detach().float().cpu())
            hit = (key, torch.linalg.inv(tril).to(device), float(tril.diagonal().log().sum()), tril.to(device))
            obj._sdeng_chol = hit
        ds.kind = L.DIST_GAUSS_FULL
        ds.loc = _dev_f32(obj.loc, device, keep)
        ds.scale = _dev_f32(obj.prec, device, keep)
        ds.w = _dev_f32(hit[1], device, keep)
        ds.p0 = hit[2]
        ds.aux = _dev_f32(hit[3], device, keep)  # L itself: GaussFull.sample (x0_dist)
        return ds
    if n in ("LogisticRegression", "SyntheticLogReg"):
        ds.kind = L.DIST_LOGREG
        ds.k = int(obj.X_train.shape[0])
        ds.loc = _dev_f32(obj.X_train, device, keep)
        ds.scale = _dev_f32(obj.y_train, device, keep)
        ds.p0, ds.p1, ds.p2 = scalar_of(obj.weight_scale), scalar_of(obj.intercept_mean), scalar_of(obj.intercept_scale)
        ds.p3 = scalar_of(obj.threshold)
        return ds
    if n == "Rings":  # distr/rings.py:38-109
        ds.kind = L.DIST_RINGS
        ds.k = int(obj.radiuses.shape[0])
        if ds.k > 8:
            raise UnsupportedByEngine("Rings: at most 8 radii")
        ds.loc = _dev_f32(obj.radiuses, device, keep)
        ds.w = _dev_f32(obj.radius_dist.mixture_distribution.probs, device, keep)
        ds.p0 = float(obj.radius_dist.component_distribution.scale.reshape(-1)[0])
        return ds
    if n in ("GMMFull", "TwoModesFull") and score_only:
        # score_mog_full inside the step loop (distr/gauss.py:110-121): the kernel takes each covariance in its eigen form, decomposed
        # once per parameter version in fp64 (the reference inverts the covariances once, in fp32, at construction)
        cov = getattr(obj, "cov", None)
        if cov is not None:
            eig = _EIGH.get(cov, lambda v: torch.linalg.eigh(v.detach().double()))
            evals, evecs = eig[0].float(), eig[1].float()
        else:
            eig = _EIGH.get(obj.prec, lambda v: torch.linalg.eigh(v.detach().double()))
            evals, evecs = (1.0 / eig[0]).float(), eig[1].float()
        ds.kind = L.DIST_GMM_FULL
        ds.k = int(obj.loc.shape[0])
        ds.loc = _dev_f32(obj.loc, device, keep)
        ds.scale = _dev_f32(evals, device, keep)
        ds.w = _dev_f32(obj.mixture_weights, device, keep)
        ds.aux = _dev_f32(evecs, device, keep)
        return ds
    raise UnsupportedByEngine(f"no HIP log-density/score for distribution {n}")


class InitialDraw:
    """``prior.sample((B,))`` left to the engine: pass it wherever a loss takes ``x`` and the kernel draws
    x0 = loc + scale * z itself (z = Philox stream 1 at step 0, keyed by the loss's seed and the global particle index), in
    registers for IsotropicGauss / Gauss / Delta priors -- x0 never exists in HBM.  ``tensor(seed, particle0)`` gives the same
    x0 as a device tensor (``sdeng_sample_x0``), for callers that need it.  Reference: IsotropicGauss.sample distr/gauss.py:772-787,
    Delta.sample distr/delta.py:27-31, GaussFull.sample distr/gauss.py:709-713 (all on torch's global generator upstream)."""

    def __init__(self, prior, batch_size: int, device=None):
        if _name(prior) == "IsotropicGauss" and getattr(prior, "truncate_quartile", None) is not None:
            raise UnsupportedByEngine("truncated IsotropicGauss prior: no native sampler")
        self.prior, self.batch_size = prior, int(batch_size)
        self.device = torch.device(device) if device is not None else next(iter(prior.buffers())).device
        self.shape = (self.batch_size, int(prior.dim))
        self.is_cuda = self.device.type == "cuda"

    def desc(self, keep) -> L.Dist:
        ds = dist_desc(self.prior, self.device, keep)
        if _name(self.prior) == "Delta":
            ds.scale = None  # Delta.sample repeats loc (its tiny scale only enters the log-density)
        if ds.kind not in (L.DIST_ISO_GAUSS, L.DIST_GAUSS_DIAG, L.DIST_GAUSS_FULL):
            raise UnsupportedByEngine(f"no native sampler for prior {_name(self.prior)}")
        return ds

    def tensor(self, seed: int, particle0: int = 0) -> torch.Tensor:
        keep = []
        ds = self.desc(keep)
        B, d = self.shape
        out = torch.empty(B, d, dtype=torch.float32, device=self.device)
        L.check(L.lib().sdeng_sample_x0(C.byref(ds), int(seed), int(particle0), B, d, out.data_ptr(), _stream_ptr(self.device)))
        return out


def sample_prior(prior, batch_size: int, seed: int, particle0: int = 0, device=None) -> torch.Tensor:
    """prior.sample((B,)) from the engine's counter-based stream: a pure function of (seed, global particle index)."""
    return InitialDraw(prior, batch_size, device).tensor(seed, particle0)


def resolve_logp(fn):
    """A log-density callable handed to the loss -> (distribution object, clip) or None if opaque."""
    owner = _self_of(fn)
    if owner is None:
        return None
    fname = getattr(fn, "__name__", "")
    if fname == "clipped_target_unnorm_log_prob" and hasattr(owner, "target"):  # solver/oc.py:80-87
        return owner.target, getattr(owner, "clip_target", None)
    if fname in ("log_prob", "unnorm_log_prob") and hasattr(owner, "dim"):
        if fname == "log_prob" and getattr(owner, "log_norm_const", 0.0) not in (0.0, None):
            return None
        return owner, None
    return None


# ------------------------------------------------------------------------------------------------
# drift net
# ------------------------------------------------------------------------------------------------
def _time_embed(te, device, keep) -> L.TimeEmbed:
    out = L.TimeEmbed()
    if _name(te) != "TimeEmbed" or te.channels != 64:
        raise UnsupportedByEngine("time embedding must be TimeEmbed(channels=64)")
    if _name(te.activation) != "GELU" or getattr(te.activation, "approximate", "none") != "none":
        raise UnsupportedByEngine("activation must be exact-erf GELU")
    out.coeff = _dev_f32(te.timestep_coeff.reshape(-1), device, keep)
    out.phase = _dev_f32(te.timestep_phase.reshape(-1), device, keep)
    n_hidden = len(te.hidden_layer)
    if n_hidden > 4:
        raise UnsupportedByEngine("TimeEmbed with more than 5 layers")
    for i, layer in enumerate(te.hidden_layer):
        out.w[i] = _dev_f32(layer.weight, device, keep)
        out.b[i] = _dev_f32(layer.bias, device, keep)
    out.n_hidden = n_hidden
    out.dim_out = int(te.out_layer.weight.shape[0])
    out.w_out = _dev_f32(te.out_layer.weight, device, keep)
    out.b_out = _dev_f32(te.out_layer.bias, device, keep)
    return out


def unwrap_ctrl(ctrl):
    """(control module the kernels know, RemoveReferenceCtrl wrapper or None).  Peels the EMA wrapper (``AveragedModel``,
    solver/oc.py:69-78) and ``RemoveReferenceCtrl`` (models/reparam.py:46-64: ``score(t, x) - ref_score(t, x)``), which the step loop
    applies as SDENG_FLAG_REMOVE_REF -- only in the form upstream's forward can evaluate (``use_rescaling=False``)."""
    if _name(ctrl) == "AveragedModel":
        ctrl = ctrl.module
    wrapper = None
    if _name(ctrl) == "RemoveReferenceCtrl":
        wrapper = ctrl
        if ctrl.use_rescaling:
            raise UnsupportedByEngine("RemoveReferenceCtrl(use_rescaling=True): upstream's forward needs the `sde` its constructor refuses "
                                      "(models/reparam.py:52, :60) and raises AttributeError; use_rescaling=False is on the HIP path")
        ctrl = ctrl.score
        if _name(ctrl) == "AveragedModel":
            ctrl = ctrl.module
    return ctrl, wrapper


def net_desc(ctrl, device, keep) -> L.Net:
    """ClippedCtrl / ScoreCtrl / LerpCtrl / CancelDriftCtrl around FourierMLP(4 layers, 64 channels, GELU) -> sdeng_net."""
    ctrl, _ = unwrap_ctrl(ctrl)
    kinds = {"ClippedCtrl": L.CTRL_CLIPPED, "ScoreCtrl": L.CTRL_SCORE, "LerpCtrl": L.CTRL_LERP, "CancelDriftCtrl": L.CTRL_CANCEL_DRIFT}
    if _name(ctrl) not in kinds:
        raise UnsupportedByEngine(f"control wrapper {_name(ctrl)} has no HIP kernel")
    net = ctrl.base_model
    if _name(net) != "FourierMLP" or net.channels != 64 or len(net.hidden_layer) != 2 or _name(net.input_embed) != "Linear":
        raise UnsupportedByEngine("drift net must be FourierMLP(num_layers=4, channels=64, use_angle_encoding=False)")
    if _name(net.activation) != "GELU" or getattr(net.activation, "approximate", "none") != "none":
        raise UnsupportedByEngine("activation must be exact-erf GELU")
    if getattr(ctrl, "hard_constrain", False):
        raise UnsupportedByEngine("LerpCtrl(hard_constrain=True)")
    n = L.Net()
    n.ctrl_kind = kinds[_name(ctrl)]
    n.w_in, n.b_in = _dev_f32(net.input_embed.weight, device, keep), _dev_f32(net.input_embed.bias, device, keep)
    n.w_h1, n.b_h1 = _dev_f32(net.hidden_layer[0].weight, device, keep), _dev_f32(net.hidden_layer[0].bias, device, keep)
    n.w_h2, n.b_h2 = _dev_f32(net.hidden_layer[1].weight, device, keep), _dev_f32(net.hidden_layer[1].bias, device, keep)
    n.w_out, n.b_out = _dev_f32(net.out_layer.weight, device, keep), _dev_f32(net.out_layer.bias, device, keep)
    n.t_embed = _time_embed(net.timestep_embed, device, keep)
    n.clip_model = float(ctrl.clip_model) if ctrl.clip_model else 0.0
    if n.ctrl_kind != L.CTRL_CLIPPED:
        n.clip_score = float(ctrl.clip_score) if ctrl.clip_score else 0.0
        n.scale_score = float(ctrl.scale_score)
        if ctrl.score_model is not None:
            n.score_model = _time_embed(ctrl.score_model, device, keep)
    return n


def ctrl_target(ctrl):
    """Distribution whose score ScoreCtrl/LerpCtrl mixes in (``target_score`` is a bound ``Distribution.score``)."""
    ctrl, _ = unwrap_ctrl(ctrl)
    if _name(ctrl) == "ClippedCtrl":
        return None, None
    tgt = _self_of(ctrl.target_score)
    if tgt is None:
        raise UnsupportedByEngine("ScoreCtrl.target_score must be a bound Distribution.score")
    prior = None
    if _name(ctrl) == "LerpCtrl":
        prior = _self_of(ctrl.prior_score)
        if prior is None or _name(prior) != "IsotropicGauss":
            raise UnsupportedByEngine("LerpCtrl.prior_score must be IsotropicGauss.score")
    return tgt, prior


# ------------------------------------------------------------------------------------------------
# reference drift
# ------------------------------------------------------------------------------------------------
def resolve_reference(reference_ctrl):
    """``reference_ctrl`` callable -> ('none'|'gaussian'|'gmm', params).  Recognises the bound
    ``RDS.reference_ctrl`` (solver/oc.py:590-592 + ``reference_distr_utils``) and this package's
    ``MarginalReference`` objects."""
    if reference_ctrl is None:
        return "none", {}
    owner = _self_of(reference_ctrl) or reference_ctrl
    utils = getattr(owner, "reference_distr_utils", None)
    if utils is None:
        raise UnsupportedByEngine("reference_ctrl is an opaque callable: pass RDS.reference_ctrl or a MarginalReference")
    if "means_init" in utils:
        return "gmm", utils  # diagonal [K,d], full [K,d,d] or the eigen form (D, P): ref_desc picks the kernel
    if "x_init" in utils:
        return "gaussian", utils  # diagonal [d], a covariance matrix [d,d] or its eigen form (D, P)
    raise UnsupportedByEngine("EBM ('nn') references need autograd inside the step: not on the HIP path")


class _TensorCache:
    """Derived facts about a caller's tensor (shared-variance test, eigendecomposition), keyed by the tensor OBJECT: the entry holds a
    weak reference, so it dies with the tensor and a new tensor the allocator places at the same address can never match; the
    (version, shape, device) part of the key catches in-place edits.  Nothing is ever written into the caller's own containers
    (``reference_distr_utils`` is iterated by ``RDS.state_dict()`` and ``MarginalReference.to()``)."""

    def __init__(self):
        self._d = {}

    def get(self, t: torch.Tensor, compute):
        key = (t._version, tuple(t.shape), str(t.device), t.data_ptr())
        hit = self._d.get(id(t))
        if hit is not None and hit[0]() is t and hit[1] == key:
            return hit[2]
        val = compute(t)
        ident = id(t)
        self._d[ident] = (weakref.ref(t, lambda _r, i=ident: self._d.pop(i, None)), key, val)
        return val


_SHARED_VAR = _TensorCache()
_EIGH = _TensorCache()
_SCALARS = _TensorCache()


def scalar_of(v) -> float:
    """float(v) for a 0-d parameter / buffer that lives on the GPU, read back ONCE per tensor version: a read-back waits for everything
    queued on the stream, so a descriptor that reads four scalars per call serialises consecutive passes (cfg 4: 6.6 ms of GPU work
    per pass became 9.7 ms of wall time)."""
    if not torch.is_tensor(v):
        return float(v)
    if not v.is_cuda:
        return float(v)
    return _SCALARS.get(v, lambda t: float(t.detach().float().cpu()))


def same_reference(a, b) -> bool:
    """Do two ``reference_distr_utils`` dicts describe the same reference (same tensors, or equal contents)?"""
    if a is b:
        return True
    if not isinstance(a, dict) or not isinstance(b, dict) or a.keys() != b.keys():
        return False
    for k in a:
        va, vb = (a[k] if isinstance(a[k], tuple) else (a[k],)), (b[k] if isinstance(b[k], tuple) else (b[k],))
        if len(va) != len(vb):
            return False
        for x, y in zip(va, vb):
            if x is not y and not (x.shape == y.shape and torch.equal(x.detach().cpu(), y.detach().cpu())):
                return False
    return True


def ref_desc(kind, utils, device, keep) -> L.Ref:
    r = L.Ref()
    if kind == "none":
        r.kind = L.REF_NONE
    elif kind == "gaussian" and (isinstance(utils["var_init"], tuple) or utils["var_init"].dim() == 2):
        # score_gauss_full (distr/gauss.py:129-135, eq/sdes.py:274-277) = the full-covariance mixture kernel with one component
        var = utils["var_init"]
        full = dict(means_init=utils["x_init"].reshape(1, -1), weights_init=torch.ones(1),
                    variances_init=(var[0].unsqueeze(0), var[1].unsqueeze(0)) if isinstance(var, tuple) else var.unsqueeze(0))
        return ref_desc("gmm", full, device, keep)
    elif kind == "gaussian":
        r.kind, r.k = L.REF_GAUSS_DIAG, 1
        r.means_init = _dev_f32(utils["x_init"].reshape(-1), device, keep)
        r.vars_init = _dev_f32(utils["var_init"].reshape(-1), device, keep)
    else:
        var = utils["variances_init"]
        r.k = int(utils["means_init"].shape[0])
        r.means_init = _dev_f32(utils["means_init"], device, keep)
        r.weights = _dev_f32(utils["weights_init"], device, keep)
        if isinstance(var, tuple) or var.dim() == 3:
            # full covariances (score_mog_full): the kernel takes the eigen form (D, P) the reference also accepts
            # (eq/sdes.py:228-238); covariance matrices are decomposed once per parameter version
            if isinstance(var, tuple):
                evals, evecs = var
            else:
                eig = _EIGH.get(var, lambda v: torch.linalg.eigh(v.detach().double()))  # once per parameter version
                evals, evecs = eig[0].float(), eig[1].float()
            r.kind = L.REF_GMM_FULL
            r.vars_init = _dev_f32(evals, device, keep)
            r.eigvecs = _dev_f32(evecs, device, keep)
        else:
            r.kind = L.REF_GMM_DIAG
            r.vars_init = _dev_f32(var, device, keep)
            # do all components share one variance vector?  (one read-back per parameter version)
            r.shared_var = 1 if _SHARED_VAR.get(var, lambda v: bool(torch.equal(v, v[:1].expand_as(v)))) else 0
    return r



# ------------------------------------------------------------------------------------------------
# per-step scalar tables
# ------------------------------------------------------------------------------------------------
def _cpu_sde(sde):
    if sde is None:
        return None
    try:
        return copy.deepcopy(sde).to("cpu")
    except Exception:  # modules holding bound methods of other modules (ControlledLangevinSDE)
        return sde


def _transition_gains(sde, s, t, ddpm):
    """(x gain, ctrl gain, noise gain) of the EI / DDPM-like kernels, with the reference's expressions
    (eq/sdes.py:532-555 VP, :658-678 PinnedBM)."""
    n = _name(sde)
    if n in ("VP", "CosineVP"):
        sig, lam = sde.scale_diff_coeff, sde.lambda_(s, t)
        if not ddpm:
            return torch.sqrt(1.0 + lam), 2.0 * sig ** 2 * (torch.sqrt(1.0 + lam) - 1.0), sig * torch.sqrt(lam)
        T = sde.terminal_t
        lam_b = 1.0 - torch.exp(sde.alpha_(T - t) - sde.alpha_(T - s))
        la = 1.0 - torch.exp(-sde.alpha_(T - s))
        lb = 1.0 - torch.exp(-sde.alpha_(T - t))
        half = (sde.alpha_(sde.terminal_t - s) - sde.alpha_(sde.terminal_t - t)) / 2.0
        var = sig ** 2 * lam_b * (lb / la)
        return torch.sqrt(1.0 + lam), 2.0 * sig ** 2 * torch.sinh(half), torch.sqrt(var)
    if n == "PinnedBM":
        g, T = sde.diff_coeff, sde.terminal_t
        var = g ** 2 * ((T - t) / (T - s)) * (t - s) if ddpm else g ** 2 * (t / s) * (t - s)
        return t / s, g ** 2 * (t - s), torch.sqrt(var)
    raise UnsupportedByEngine(f"{n} has no closed-form EI/DDPM transition kernel")


def coef_table(kind, ts, sde=None, *, with_ref=False, lerp=False, cancel=None, ctrl_sde=None, alpha=None, sigma=None, train=False, dim=1,
               rescale=True) -> torch.Tensor:
    """[N,16] fp32 table (column meaning: include/sdeng.h).  ``kind``: 'ei' | 'ddpm' | 'dis_ei' | 'em' |
    'time_reversal' | 'dds' | 'eubo_ei' | 'eubo_em'.  ``ts`` is a CPU tensor; every entry is produced by the reference's scalar formula."""
    ts = ts.detach().to("cpu", torch.float32)
    N = ts.numel() - 1
    if kind == "cmcd_eubo":  # N+1 rows in iteration order: row k is the evaluation at time ts[N-k] (losses/oc.py:782-823)
        out = torch.zeros(N + 1, L.NCOEF, dtype=torch.float32)
        Tstar = sde.terminal_t
        for k in range(N + 1):
            tk = ts[N - k]
            out[k, 0] = tk
            out[k, 4], out[k, 5] = tk / Tstar, 1.0 - tk / Tstar  # eq/sdes.py:103 at this evaluation's time
            out[k, 6], out[k, 7] = out[k, 4], out[k, 5]
            if k < N:
                dt = ts[N - k] - ts[N - 1 - k]
                out[k, 2], out[k, 3] = dt, dt.sqrt()
        return out
    if kind == "cmcd":  # N+1 rows: row k holds the times/weights of step k; the last row only ts[N] (net time)
        out = torch.zeros(N + 1, L.NCOEF, dtype=torch.float32)
        Tstar = sde.terminal_t
        for k in range(N + 1):
            out[k, 0] = ts[k]
            if k < N:
                s, t = ts[k], ts[k + 1]
                dt = t - s
                out[k, 1], out[k, 2], out[k, 3] = t, dt, dt.sqrt()
                out[k, 4], out[k, 5] = s / Tstar, 1.0 - s / Tstar  # eq/sdes.py:103
                out[k, 6], out[k, 7] = t / Tstar, 1.0 - t / Tstar
        return out
    out = torch.zeros(N, L.NCOEF, dtype=torch.float32)
    T = ts[-1]
    for k in range(N):
        s, t = ts[k], ts[k + 1]
        row = out[k]
        row[7] = 1.0
        if kind in ("ei", "ddpm", "dis_ei"):
            tau = T - s
            omega = sde.omega_ddpm(s, t) if kind == "ddpm" else sde.omega(s, t)
            g1, g2, g3 = _transition_gains(sde, s, t, kind == "ddpm")
            row[0], row[1], row[2], row[3] = tau, g1, g2, g3
            row[4], row[5] = 0.5 * omega, torch.sqrt(omega)
        elif kind == "em":
            tau = T - s
            dt = t - s
            g = sde.diff(tau, None)
            row[0], row[1], row[2], row[3] = tau, -sde.drift_coeff_t(tau), g, torch.square(g)
            row[4], row[5] = dt, dt.sqrt()
        elif kind == "time_reversal":
            tau = s
            dt = t - s
            g = sde.diff(s, None)
            row[0], row[1], row[2], row[3] = s, sde.drift_coeff_t(s), g, torch.square(g)
            row[4], row[5] = dt, dt.sqrt()
            if not train:
                row[6] = -(sde.int_drift_coeff_t(s, t) * dim)  # losses/oc.py:1218-1219, eq/sdes.py:137-141
        elif kind in ("eubo_ei", "eubo_em"):
            # compute_eubo (losses/oc.py:325-358, 539-564): iteration k noises from time T - t to T - s with
            # (s, t) = (ts[N-1-k], ts[N-k]); the rows are laid out in iteration order
            s, t = ts[N - 1 - k], ts[N - k]
            tau = T - s
            mean_f, var_f = sde.transition_params(T - t, T - s)
            std_f = var_f.sqrt()
            row[0], row[1], row[2], row[3] = tau, mean_f, 1.0, std_f
            if kind == "eubo_ei":
                omega = sde.omega(s, t)
                row[4], row[5] = omega, torch.sqrt(omega)
            else:
                g = sde.diff(tau, None)
                dt = t - s
                if rescale:
                    row[2] = 1.0 / g
                row[4] = dt * g ** 2
                row[5] = std_f / mean_f
                row[6] = 1.0 / mean_f - 1.0 + sde.drift_coeff_t(tau) * dt
        elif kind == "dds":
            tau = s
            dt = t - s
            beta_k = torch.clip(alpha * dt.sqrt(), 0, 1)
            alpha_k = torch.sqrt(1.0 - beta_k ** 2)
            row[0], row[1] = s, alpha_k
            row[2] = (beta_k ** 2) * (sigma ** 2)
            row[3] = sigma * beta_k
            row[4] = 0.5 * (beta_k ** 2 * sigma ** 2)
            row[5] = sigma * beta_k
        else:
            raise ValueError(kind)
        csde = ctrl_sde if ctrl_sde is not None else sde  # the control's own SDE (DDS has no loss-level SDE)
        if lerp:  # LerpCtrl: weight t/T (reparam.py:175) and gain g(t) (reparam.py:199), at the net's time
            t_net = row[0].clone()
            row[7] = csde.diff(t_net, None)
            row[8] = t_net / csde.terminal_t
        if cancel is not None:  # CancelDriftCtrl (reparam.py:131-145): drift/g + (g/2) score, or drift/g^2 + score/2
            t_net = row[0].clone()
            g = csde.diff(t_net, None)
            row[7] = 0.5 * g if cancel == "rescale" else 0.5
            row[8] = csde.drift_coeff_t(t_net) / (g if cancel == "rescale" else torch.square(g))
        if with_ref:  # eq/sdes.py:228-229, 247
            s_tau = sde.s(tau)
            row[9], row[10], row[11] = s_tau, s_tau ** 2 * sde.sigma_sq(tau), s_tau ** 2
    return out



# ------------------------------------------------------------------------------------------------
# launch
# ------------------------------------------------------------------------------------------------
class Workspace:
    """Grow-only device scratch per (device) -- the C ABI never allocates."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes, device):
        cur = self.buf.get(device)
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self.buf[device] = cur
        return cur


_WS = Workspace()


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError("the sdeng simulate path runs on an MI355X only: move the solver to a cuda device "
                           "(no CPU implementation is shipped)")


def run(desc: L.Desc, x: torch.Tensor, keep: list, return_traj=False, noise=None, events=None):
    """Fill the I/O pointers of ``desc`` and launch.  Returns (x_N [B,d], rnd [B,1], xs or None)."""
    require_gpu(x)
    lib = L.lib()
    device = x.device
    B, d, N = x.shape[0], x.shape[1], desc.N
    if isinstance(x, InitialDraw):  # x0 drawn by the engine: no input array
        desc.x_in = None
        desc.x0_dist = x.desc(keep)
        x_out = torch.empty(B, d, dtype=torch.float32, device=device)
    else:
        xin = x.detach().to(torch.float32).contiguous()
        keep.append(xin)
        desc.x_in = xin.data_ptr()
        x_out = torch.empty_like(xin)
    rnd = torch.empty(B, 1, dtype=torch.float32, device=device)
    xs = torch.empty(N + 1, B, d, dtype=torch.float32, device=device) if return_traj else None
    desc.abi_version = L.ABI_VERSION
    desc.B, desc.d = B, d
    desc.x_out, desc.rnd_out = x_out.data_ptr(), rnd.data_ptr()
    desc.xs_out = xs.data_ptr() if return_traj else None
    if noise is not None:
        nz = noise.detach().to(device=device, dtype=torch.float32).contiguous()
        assert nz.shape == (N, B, d), f"noise must be [N,B,d] = {(N, B, d)}, got {tuple(nz.shape)}"
        keep.append(nz)
        desc.noise_in = nz.data_ptr()
    else:
        desc.noise_in = None
    need = lib.sdeng_workspace_bytes(C.byref(desc))
    ws = _WS.get(need, device)
    desc.workspace, desc.workspace_bytes = ws.data_ptr(), ws.numel()
    if events is not None:
        desc.ev_start, desc.ev_stop = events.start, events.stop
    L.check(lib.sdeng_simulate(C.byref(desc), _stream_ptr(device)))
    return x_out, rnd, xs


def euler_states(sde, timesteps: torch.Tensor, x: torch.Tensor, increments=None, seed: int = 0, particle0: int = 0, events=None):
    """All N+1 Euler-Maruyama states ``[N+1, B, d]`` of ``sde`` on the grid ``timesteps`` in one launch
    (the loop of EulerIntegrator.integrate, eq/integrator.py:113-122: ``xt = xs + drift(s, xs) (t-s) + diff(s, xs) noise``).

    * ``LangevinSDE`` (eq/sdes.py:46-76) and uncontrolled ``OU`` SDEs / ``ControlledSDE(sde, None)`` run the net-free kernel
      (SDENG_CTRL_NONE);
    * ``ControlledSDE(sde, ctrl)`` (eq/sdes.py:681-720; the net is evaluated at ``T - s``) runs the step-loop kernel.
    ``increments`` ``[N,B,d]`` replays Brownian increments (the reference's ``bm(s, t)``); otherwise Philox normals times sqrt(dt).
    """
    require_gpu(x)
    device, keep = x.device, []
    ts = timesteps.detach().to("cpu", torch.float32)
    N = ts.numel() - 1
    desc = L.Desc()
    desc.form, desc.flags, desc.N = L.FORM_EM, 0, N
    desc.seed, desc.particle0 = int(seed), int(particle0)
    coef = torch.zeros(N, L.NCOEF, dtype=torch.float32)
    name = _name(sde)
    ctrl = getattr(sde, "ctrl", None) if name == "ControlledSDE" else None
    if name == "LangevinSDE":
        tgt = _self_of(sde.target_score)
        if tgt is None:
            raise UnsupportedByEngine("LangevinSDE.target_score must be a bound Distribution.score")
        desc.target = dist_desc(tgt, device, keep)
        desc.net.ctrl_kind = L.CTRL_NONE
        desc.net.clip_score = float(sde.clip_score) if sde.clip_score else 0.0
        g = sde.diff_coeff.detach().float().cpu()
        coef[:, 2], coef[:, 7] = g, g ** 2 / 2.0  # eq/sdes.py:65
    else:
        base = _cpu_sde(sde.sde if name == "ControlledSDE" else sde)
        if not hasattr(base, "drift_coeff_t"):
            raise UnsupportedByEngine(f"EulerIntegrator: no HIP kernel for SDE {name}")
        T = base.terminal_t
        if ctrl is None:
            desc.net.ctrl_kind = L.CTRL_NONE
        else:
            desc.net = net_desc(ctrl, device, keep)
            tgt, lerp_prior = ctrl_target(ctrl)
            if tgt is not None:
                desc.target = dist_desc(tgt, device, keep)
            if lerp_prior is not None:
                desc.prior = dist_desc(lerp_prior, device, keep)
        for k in range(N):
            s = ts[k]
            g = base.diff(s, None)
            coef[k, 0], coef[k, 1], coef[k, 2], coef[k, 3] = T - s, base.drift_coeff_t(s), g, torch.square(g)
            if ctrl is not None and _name(ctrl) == "LerpCtrl":  # reparam.py:175, :199 at the net's time T - s
                coef[k, 7], coef[k, 8] = base.diff(T - s, None), (T - s) / T
    dt = ts[1:] - ts[:-1]
    coef[:, 4] = dt
    coef[:, 5] = 1.0 if increments is not None else dt.sqrt()
    coef = coef.to(device)
    keep.append(coef)
    desc.coef = coef.data_ptr()
    _, _, xs = run(desc, x, keep, return_traj=True, noise=increments, events=events)
    return xs


def philox_noise(seed: int, N: int, B: int, d: int, particle0: int, device) -> torch.Tensor:
    """[N,B,d] normals of the step loop's counter-based stream (the kernel draws exactly these when no noise is injected)."""
    lib = L.lib()
    out = torch.empty(N, B, d, dtype=torch.float32, device=device)
    L.check(lib.sdeng_philox_normal_steps(int(seed), 0, N, int(particle0), B, d, 0, out.data_ptr(), _stream_ptr(device)))
    return out


def logz_stats(rnd: torch.Tensor, want_weights=True):
    """sdeng_logz -> (stats[8] device tensor, weights [B,1] or None)."""
    require_gpu(rnd)
    lib = L.lib()
    device = rnd.device
    r = rnd.detach().to(torch.float32).contiguous().view(-1)
    stats = torch.empty(8, dtype=torch.float32, device=device)
    w = torch.empty(r.numel(), 1, dtype=torch.float32, device=device) if want_weights else None
    ws = torch.empty(lib.sdeng_logz_workspace_bytes(), dtype=torch.uint8, device=device)
    L.check(lib.sdeng_logz(r.data_ptr(), r.numel(), stats.data_ptr(), w.data_ptr() if want_weights else None, ws.data_ptr(),
                           ws.numel(), _stream_ptr(device)))
    return stats, w


# ------------------------------------------------------------------------------------------------
# standalone pieces of the ABI (unit parity tests; also handy for inspection)
# ------------------------------------------------------------------------------------------------
def dist_eval(dist, x: torch.Tensor, want_logp=True, want_score=True):
    """sdeng_dist_eval: log-density [B,1] and score [B,d] of a distribution object, computed in HIP."""
    require_gpu(x)
    lib = L.lib()
    device, keep = x.device, []
    ds = dist_desc(dist, device, keep)
    xin = x.detach().to(torch.float32).contiguous()
    B, d = xin.shape
    logp = torch.empty(B, 1, dtype=torch.float32, device=device) if want_logp else None
    score = torch.empty(B, d, dtype=torch.float32, device=device) if want_score else None
    need = lib.sdeng_dist_workspace_bytes(C.byref(ds), d)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=device)
    L.check(lib.sdeng_dist_eval(C.byref(ds), B, d, xin.data_ptr(), logp.data_ptr() if want_logp else None,
                                score.data_ptr() if want_score else None, ws.data_ptr(), ws.numel(), _stream_ptr(device)))
    return logp, score


def langevin_moves(target, prior, x, lp, grad, step, n_moves, *, t=None, keep_from=0, unadjusted=False, target_acceptance=0.0, noise="torch",
                   seed=0, chain0=0, want_samples=True):
    """sdeng_langevin_moves: ``n_moves`` MALA / ULA moves of all chains in one launch (include/sdeng.h).  ``x`` [B,d], ``lp`` [B], ``grad``
    [B,d], ``step`` [B] are updated IN PLACE.  ``noise='torch'``: the normals (and uniforms) are drawn from torch's generator move by move in
    the order additions/mcmc.py consumes them (randn((B,d)) then rand_like(lp)), so a chain is the one the reference's loop produces;
    ``noise='philox'``: drawn in the kernel.  Returns (samples [n_moves - keep_from, B, d] or None, acc_sum [B] or None, acc_last [B] or None)."""
    require_gpu(x)
    lib = L.lib()
    device, keep = x.device, []
    B, d = x.shape
    for name, v in (("x", x), ("lp", lp), ("grad", grad), ("step", step)):
        if v.dtype != torch.float32 or not v.is_contiguous() or not v.is_cuda:
            raise ValueError(f"langevin_moves: {name} must be a contiguous float32 CUDA tensor (it is updated in place)")
    dt = dist_desc(target, device, keep)
    dp = dist_desc(prior, device, keep) if prior is not None else None
    z = u = None
    if noise == "torch":
        z = torch.empty(n_moves, B, d, dtype=torch.float32, device=device)
        u = torch.empty(n_moves, B, dtype=torch.float32, device=device) if not unadjusted else None
        for m in range(n_moves):  # the reference's consumption order: one randn per proposal, then one rand per accept test
            z[m] = torch.randn((B, d), device=device)
            if u is not None:
                u[m] = torch.rand((B,), device=device)
    elif noise != "philox":
        raise ValueError("noise: 'torch' or 'philox'")
    tt = None if t is None else t.detach().to(device=device, dtype=torch.float32).reshape(-1).expand(B).contiguous()
    samples = torch.empty(n_moves - keep_from, B, d, dtype=torch.float32, device=device) if want_samples else None
    acc = torch.empty(B, dtype=torch.float32, device=device) if not unadjusted else None
    last = torch.empty(B, dtype=torch.float32, device=device) if not unadjusted else None
    need = lib.sdeng_langevin_moves_workspace_bytes(C.byref(dp) if dp is not None else None, C.byref(dt), d)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=device)
    ptr = lambda v: None if v is None else v.data_ptr()  # noqa: E731
    L.check(lib.sdeng_langevin_moves(C.byref(dp) if dp is not None else None, C.byref(dt), B, d, int(n_moves), int(keep_from), int(bool(unadjusted)),
                                     float(target_acceptance), ptr(tt), x.data_ptr(), lp.data_ptr(), grad.data_ptr(), step.data_ptr(), ptr(z), ptr(u),
                                     int(seed), int(chain0), ptr(samples), ptr(acc), ptr(last), ws.data_ptr(), ws.numel(), _stream_ptr(device)))
    return samples, acc, last


def ctrl_vjp(ctrl, t_unique: torch.Tensor, xs: torch.Tensor, cot: torch.Tensor, want_gx=False):
    """sdeng_ctrl_vjp: fused forward + backward of a ClippedCtrl / FourierMLP over all (time, state) rows.  ``xs`` and ``cot`` are
    [N, B, d] (states at the N times ``t_unique`` and the cotangent of the control there).  Returns the per-row arrays the parameter
    gradients are built from (include/sdeng.h): dict(a0, a1, a2, d0, d1, d2 [N*B, 64], dout [N*B, d], gx [N*B, d] or None)."""
    require_gpu(xs)
    lib = L.lib()
    device, keep = xs.device, []
    N, B, d = xs.shape
    desc = L.Desc()
    desc.abi_version = L.ABI_VERSION
    desc.net = net_desc(ctrl, device, keep)
    if desc.net.ctrl_kind != L.CTRL_CLIPPED:
        raise UnsupportedByEngine("ctrl_vjp: ClippedCtrl only")
    desc.d = d
    coef = torch.zeros(N, L.NCOEF, dtype=torch.float32, device=device)
    coef[:, 0] = t_unique.detach().to(device=device, dtype=torch.float32).reshape(-1)
    desc.coef = coef.data_ptr()
    x2 = xs.detach().to(torch.float32).contiguous().view(N * B, d)
    c2 = cot.detach().to(torch.float32).contiguous().view(N * B, d)
    hid = torch.empty(6, N * B, 64, dtype=torch.float32, device=device)
    dout = torch.empty(N * B, d, dtype=torch.float32, device=device)
    gx = torch.empty(N * B, d, dtype=torch.float32, device=device) if want_gx else None
    ws = _WS.get(lib.sdeng_ctrl_vjp_workspace_bytes(d, N), device)
    desc.workspace, desc.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(lib.sdeng_ctrl_vjp(C.byref(desc), N, B, x2.data_ptr(), c2.data_ptr(), hid[0].data_ptr(), hid[1].data_ptr(), hid[2].data_ptr(),
                               hid[3].data_ptr(), hid[4].data_ptr(), hid[5].data_ptr(), dout.data_ptr(), gx.data_ptr() if want_gx else None,
                               None, _stream_ptr(device)))
    return dict(x=x2, a0=hid[0], a1=hid[1], a2=hid[2], d0=hid[3], d1=hid[4], d2=hid[5], dout=dout, gx=gx)


class VjpSession:
    """sdeng_ctrl_vjp for a caller that walks the N times one by one (the adjoint recursion of KL training): the weight images are packed
    once, ``forward_u`` evaluates the control on all N * B rows, ``step(k, cot_k)`` runs the fused forward + backward of time k alone
    (cotangent [B, d]) -- its per-row arrays land in row block k of the session's [N * B, .] arrays -- and returns the state gradient
    [B, d].  After the walk the arrays are exactly what ``ctrl_vjp`` returns for the whole batch."""

    def __init__(self, ctrl, t_unique: torch.Tensor, xs: torch.Tensor):
        require_gpu(xs)
        self.lib, self.device, self.keep = L.lib(), xs.device, []
        self.N, self.B, self.d = xs.shape
        N, B, d = xs.shape
        self.desc = L.Desc()
        self.desc.abi_version = L.ABI_VERSION
        self.desc.net = net_desc(ctrl, self.device, self.keep)
        if self.desc.net.ctrl_kind != L.CTRL_CLIPPED:
            raise UnsupportedByEngine("ctrl_vjp: ClippedCtrl only")
        self.desc.d = d
        self.coef = torch.zeros(N, L.NCOEF, dtype=torch.float32, device=self.device)
        self.coef[:, 0] = t_unique.detach().to(device=self.device, dtype=torch.float32).reshape(-1)
        self.x = xs.detach().to(torch.float32).contiguous().view(N * B, d)
        self.hid = torch.empty(6, N * B, 64, dtype=torch.float32, device=self.device)
        self.dout = torch.empty(N * B, d, dtype=torch.float32, device=self.device)
        self.gx = torch.empty(B, d, dtype=torch.float32, device=self.device)
        self.ws = torch.empty(self.lib.sdeng_ctrl_vjp_workspace_bytes(d, N), dtype=torch.uint8, device=self.device)  # its own: the images must survive
        self.desc.workspace, self.desc.workspace_bytes = self.ws.data_ptr(), self.ws.numel()
        self.packed = False

    def forward_u(self) -> torch.Tensor:
        u = torch.empty(self.N * self.B, self.d, dtype=torch.float32, device=self.device)
        self.desc.coef, self.desc.flags = self.coef.data_ptr(), 0
        L.check(self.lib.sdeng_ctrl_vjp(C.byref(self.desc), self.N, self.B, self.x.data_ptr(), None, None, None, None, None, None, None, None, None,
                                        u.data_ptr(), _stream_ptr(self.device)))
        self.packed = True
        return u.view(self.N, self.B, self.d)

    def step(self, k: int, cot: torch.Tensor) -> torch.Tensor:
        B, d = self.B, self.d
        c = cot.detach().to(torch.float32).contiguous()
        self.keep.append(c)
        self.desc.coef = self.coef.data_ptr() + 4 * L.NCOEF * k
        self.desc.flags = L.FLAG_REUSE_PACK if self.packed else 0
        row = lambda t, width: t.data_ptr() + 4 * width * B * k  # noqa: E731
        L.check(self.lib.sdeng_ctrl_vjp(C.byref(self.desc), 1, B, row(self.x, d), c.data_ptr(), row(self.hid[0], 64), row(self.hid[1], 64),
                                        row(self.hid[2], 64), row(self.hid[3], 64), row(self.hid[4], 64), row(self.hid[5], 64), row(self.dout, d),
                                        self.gx.data_ptr(), None, _stream_ptr(self.device)))
        self.packed = True
        self.keep.clear()
        return self.gx

    def arrays(self):
        h = self.hid
        return dict(x=self.x, a0=h[0], a1=h[1], a2=h[2], d0=h[3], d1=h[4], d2=h[5], dout=self.dout, gx=None)


def diagonal_reference(kind, utils) -> bool:
    """No reference drift, or a Gaussian / mixture reference with diagonal covariances (what sdeng_kl_adjoint differentiates in closed form)."""
    if kind == "none":
        return True
    if kind == "gaussian":
        v = utils["var_init"]
        return not isinstance(v, tuple) and v.dim() <= 1
    if kind == "gmm":
        v = utils["variances_init"]
        return not isinstance(v, tuple) and v.dim() == 2
    return False


# targets whose ``score`` is the base class's autograd evaluation WITHOUT a graph (distr/base.py:146-154): back-propagation through the
# trajectory sees it as a constant of x, in the reference and here
_GRAPHLESS_SCORE = ("LogisticRegression", "SyntheticLogReg")


def adjoint_ctrl_ok(ctrl) -> bool:
    """Does sdeng_kl_adjoint differentiate this control?  ClippedCtrl over a FourierMLP, or a plain ScoreCtrl over one whose target is a
    diagonal mixture or the phi^4 lattice (BASELINE configs 1 and 3: DDS on TwoModes, PIS on PhiFour) and whose score model, if any, is a TimeEmbed."""
    name = type(ctrl).__name__
    if type(getattr(ctrl, "base_model", None)).__name__ != "FourierMLP":
        return False
    if name == "ClippedCtrl":
        return True
    if name not in ("ScoreCtrl", "LerpCtrl", "CancelDriftCtrl"):
        return False
    if name == "LerpCtrl" and (getattr(ctrl, "hard_constrain", False) or _name(_self_of(ctrl.prior_score)) != "IsotropicGauss"):
        return False
    tgt = _self_of(ctrl.target_score)
    if tgt is None:
        return False
    if _name(tgt) == "PhiFour":
        try:
            dist_desc(tgt, "cpu", [])  # (the 1-D Dirichlet-0 untilted lattice only)
        except UnsupportedByEngine:
            return False
    elif _name(tgt) in _GRAPHLESS_SCORE:
        pass  # the score of every row comes from the HIP score kernel and is a constant of x, as upstream's autograd-made score is
    elif _name(tgt) not in ("GMM", "TwoModes", "ManyModes", "BracketTwoModes") or getattr(tgt, "mixture_weights", None) is None:
        return False
    return ctrl.score_model is None or _name(ctrl.score_model) == "TimeEmbed"


def kl_adjoint(ctrl, coef: torch.Tensor, xs: torch.Tensor, z, w: torch.Tensor, lam_n: torch.Tensor, *, lin: bool, ito: bool, ref=("none", {})):
    """sdeng_kl_adjoint: the whole adjoint recursion of KL training in one launch (ClippedCtrl, or ScoreCtrl on a diagonal mixture target; no
    reference, or a diagonal Gaussian / mixture reference).  ``coef`` [N,16] on the device, ``xs`` [N,B,d] = x_0 .. x_{N-1}, ``z`` [N,B,d]
    the trajectory's normals (or None), ``w`` [B,1] = d loss / d rnd, ``lam_n`` [B,d] = lambda_N.  Returns (per-row arrays as ``ctrl_vjp``
    -- plus ``dst`` [N,B], the per-particle cotangent of s_theta(t_k), for a ScoreCtrl --, lambda_0)."""
    require_gpu(xs)
    lib = L.lib()
    device, keep = xs.device, []
    N, B, d = xs.shape
    desc = L.Desc()
    desc.abi_version = L.ABI_VERSION
    desc.form = L.FORM_LIN if lin else L.FORM_EM
    desc.flags = L.FLAG_ITO if ito else 0
    desc.B, desc.d, desc.N = B, d, N
    desc.net = net_desc(ctrl, device, keep)
    if not adjoint_ctrl_ok(ctrl):
        raise UnsupportedByEngine("kl_adjoint: ClippedCtrl, or ScoreCtrl on a diagonal mixture target")
    score = desc.net.ctrl_kind != L.CTRL_CLIPPED
    ext_score = None
    if score:
        tgt, lerp_prior = ctrl_target(ctrl)
        if _name(tgt) in _GRAPHLESS_SCORE:
            _, ext_score = dist_eval(tgt, xs.reshape(N * B, d), want_logp=False, want_score=True)
        else:
            desc.target = dist_desc(tgt, device, keep)
        if lerp_prior is not None:
            desc.prior = dist_desc(lerp_prior, device, keep)
    desc.ref = ref_desc(ref[0], ref[1], device, keep)
    if desc.ref.kind not in (L.REF_NONE, L.REF_GAUSS_DIAG, L.REF_GMM_DIAG):
        raise UnsupportedByEngine("kl_adjoint: diagonal references only")
    cf = coef.detach().to(device=device, dtype=torch.float32).contiguous()
    desc.coef = cf.data_ptr()
    x = xs.detach().to(torch.float32).contiguous().view(N * B, d)
    hid = torch.empty(6, N * B, 64, dtype=torch.float32, device=device)
    dout = torch.empty(N * B, d, dtype=torch.float32, device=device)
    lam0 = torch.empty(B, d, dtype=torch.float32, device=device)
    wv = w.detach().to(torch.float32).contiguous().view(B)
    ln = lam_n.detach().to(torch.float32).contiguous()
    zz = z.detach().to(torch.float32).contiguous() if (ito and z is not None) else None
    ws = _WS.get(lib.sdeng_kl_adjoint_workspace_bytes(C.byref(desc)), device)
    desc.workspace, desc.workspace_bytes = ws.data_ptr(), ws.numel()
    adj = L.Adjoint()
    adj.xs, adj.noise, adj.w, adj.lam_in, adj.lam_out = x.data_ptr(), (zz.data_ptr() if zz is not None else None), wv.data_ptr(), ln.data_ptr(), lam0.data_ptr()
    adj.a0, adj.a1, adj.a2, adj.d0, adj.d1, adj.d2 = (hid[i].data_ptr() for i in range(6))
    adj.dout = dout.data_ptr()
    dst = torch.empty(N, B, dtype=torch.float32, device=device) if score else None
    if score:
        adj.dst, adj.detach_score = dst.data_ptr(), int(bool(ctrl.detach_score))
        adj.score = ext_score.data_ptr() if ext_score is not None else None
    L.check(lib.sdeng_kl_adjoint(C.byref(desc), C.byref(adj), _stream_ptr(device)))
    return dict(x=x, a0=hid[0], a1=hid[1], a2=hid[2], d0=hid[3], d1=hid[4], d2=hid[5], dout=dout, gx=None, dst=dst), lam0


def ctrl_forward(ctrl, t: float, x: torch.Tensor, score_gain=1.0, lerp_w=0.0):
    """sdeng_ctrl_forward: u = ctrl(t, x) for a ClippedCtrl / ScoreCtrl / LerpCtrl module, computed in HIP."""
    require_gpu(x)
    lib = L.lib()
    device, keep = x.device, []
    desc = L.Desc()
    desc.abi_version = L.ABI_VERSION
    desc.net = net_desc(ctrl, device, keep)
    tgt, prior = ctrl_target(ctrl)
    desc.target = dist_desc(tgt, device, keep)
    desc.prior = dist_desc(prior, device, keep)
    xin = x.detach().to(torch.float32).contiguous()
    desc.B, desc.d, desc.N = xin.shape[0], xin.shape[1], 1
    out = torch.empty_like(xin)
    need = lib.sdeng_workspace_bytes(C.byref(desc))
    ws = _WS.get(need, device)
    desc.workspace, desc.workspace_bytes = ws.data_ptr(), ws.numel()
    L.check(lib.sdeng_ctrl_forward(C.byref(desc), float(t), float(score_gain), float(lerp_w), xin.data_ptr(), out.data_ptr(),
                                   _stream_ptr(device)))
    return out
