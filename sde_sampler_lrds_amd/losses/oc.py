"""Drop-in loss classes for the reference's ``sde_sampler/losses/oc.py`` (``conf/loss/*.yaml:_target_``).

Same class names, constructor arguments, ``simulate`` / ``eval`` / ``state_dict`` surface and return types as
the reference (``BaseOCLoss`` :14-200 and its seven subclasses), but ``simulate`` is ONE call into the HIP
engine (csrc/) instead of a Python step loop: every step's drift-net forward, scores, noise, integrator
update and log-RND accumulation run inside a single persistent gfx950 kernel.

What is on the HIP path: the eval / sampling direction (``change_sde_ctrl=False``), i.e. what
``Trainable.evaluate`` times as ``eval/sample_time`` (solver/oc.py:148-158); ``x`` may be an ``engine.InitialDraw`` (x0 drawn by the
kernel) and ``loss.dist`` a torch.distributed module (sharded run: global estimators and weights).  ``compute_eubo`` (the noising loops of
SURVEY.md 8f-2) is a HIP launch too (RDS losses, DiscreteTimeReversalLossEI, CMCD on mixture targets).  The training direction (``__call__``,
8f-1) is built for the log-variance methods (``_lv_loss``: HIP step loop + one batched autograd pass of the control); KL
training raises instead of silently running a PyTorch loop.

Extra, engine-only knobs (keyword-only, default to the reference behaviour):
  * ``noise=[N,B,d]`` injects the normals (replays the reference's ``randn_like`` stream, parity mode);
    without it the kernel draws counter-based Philox normals keyed by ``self.seed`` and the global
    particle index ``self.particle0 + row`` (independent of how the batch is sharded over GPUs).
"""
from __future__ import annotations

import logging
from typing import Callable

import torch

from .. import _lib as L
from .. import engine as E
from ..utils.common import Results, make_results


def ctrl_batched(ctrl, t_unique: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``ctrl(t, x)`` (with autograd) for M distinct times and B states per time, x [M,B,d] -> [M*B,d].  Same operations
    as the modules' own forward (models/mlp.py, models/reparam.py), except that the time embeddings -- functions of t only --
    are evaluated once per distinct time instead of once per row (the reference recomputes them for every particle,
    models/mlp.py:136-137); other control types fall back to the plain per-row call."""
    from ..models.reparam import _clip
    M, B, d = x.shape
    flat = x.reshape(M * B, d)
    t_rows = t_unique.repeat_interleave(B).view(-1, 1)
    name, net = type(ctrl).__name__, getattr(ctrl, "base_model", None)
    if name not in ("ClippedCtrl", "ScoreCtrl", "LerpCtrl", "CancelDriftCtrl") or type(net).__name__ != "FourierMLP":
        return ctrl(t_rows, flat)
    h = net.input_embed(flat) + net.timestep_embed(t_unique.view(-1, 1)).repeat_interleave(B, dim=0)
    for layer in net.hidden_layer:
        h = layer(net.activation(h))
    out = _clip(net.out_layer(net.activation(h)), ctrl.clip_model)
    if name == "ClippedCtrl":
        return out
    if name in ("ScoreCtrl", "CancelDriftCtrl"):
        score = ctrl.scale_score * ctrl.clipped_target_score(t_rows, flat)
    else:
        score = ctrl.scale_score * ctrl.clipped_interpolated_score(t_rows, flat)
    if ctrl.score_model is not None:
        score = score * _clip(ctrl.score_model(t_unique.view(-1, 1)), ctrl.clip_model).repeat_interleave(B, dim=0)
    if name == "CancelDriftCtrl":  # reparam.py:142-145
        g, f = ctrl.sde.diff(t_rows, flat), ctrl.sde.drift(t_rows, flat)
        return out + (f / g) + 0.5 * g * score if ctrl.use_rescaling else out + (f / torch.square(g)) + 0.5 * score
    return out + (ctrl.sde.diff(t_rows, flat) * score if name == "LerpCtrl" else score)


class _FusedIntegral(torch.autograd.Function):
    """s_b = sum_k <u_theta(t_k, x_kb), zc_kb> for a ClippedCtrl over a FourierMLP, with the gradient w.r.t. the net's parameters from
    ONE fused HIP forward + backward over all N * B rows (``sdeng_ctrl_vjp``, csrc/grad_kernel.hpp) and six skinny GEMMs -- instead of
    the ~150 small kernels of the eager torch pass.  The VALUE of s is not needed by the losses (it enters as ``s - s.detach()``), so
    ``forward`` returns zeros and all the work happens in ``backward``, where the cotangent of s_b (one number per particle) is known."""

    @staticmethod
    def forward(ctx, ctrl, t_unique, xs, zc, *params):
        ctx.ctrl, ctx.shape = ctrl, tuple(xs.shape)
        ctx.save_for_backward(t_unique, xs, zc)
        return torch.zeros(xs.shape[1], dtype=xs.dtype, device=xs.device)

    @staticmethod
    def backward(ctx, grad_s):
        t_unique, xs, zc = ctx.saved_tensors
        N, B, d = ctx.shape
        cot = zc.view(N, B, d) * grad_s.view(1, B, 1)  # d loss / d u_kb
        ctrl = ctx.ctrl
        net_view = fused_net_view(ctrl)
        grads = vjp_param_grads(net_view, t_unique, E.ctrl_vjp(net_view, t_unique, xs, cot), N, B)
        sm = getattr(ctrl, "score_model", None)
        sm_params = [p for p in sm.parameters() if p.requires_grad] if (net_view is not ctrl and sm is not None) else []
        if sm_params:
            # ScoreCtrl: u = clip(net) + scale clip(score_pi(x)) s_theta(t).  The states are constants, so the score part only reaches the
            # score model: d loss / d s_theta(t_k) = sum_b <cot_kb, scale clip(score_pi(x_kb))> (HIP score kernel, one launch for all rows)
            from ..models.reparam import _clip
            _, sc = E.dist_eval(E.ctrl_target(ctrl)[0], xs.reshape(N * B, d), want_logp=False, want_score=True)
            dst = (cot.reshape(N * B, d) * (ctrl.scale_score * _clip(sc, ctrl.clip_score))).sum(-1).view(N, B).sum(1)
            with torch.enable_grad():
                st = ctrl.clipped_score_model(t_unique.view(-1, 1), None).view(N)
                sm_grads = torch.autograd.grad(st, sm_params, grad_outputs=dst, allow_unused=True)
            grads.update({p: g for p, g in zip(sm_params, sm_grads) if g is not None})
        params = [p for p in ctrl.parameters() if p.requires_grad]
        return (None, None, None, None) + tuple(grads.get(p) for p in params)


_NET_VIEWS = {}


def fused_training_ok(ctrl) -> bool:
    """Controls whose batched log-variance pass is the fused HIP forward + backward: ClippedCtrl over a FourierMLP, and a plain ScoreCtrl
    over one (its score part has no state gradient to take in this pass) on a target the score kernel knows."""
    if type(getattr(ctrl, "base_model", None)).__name__ != "FourierMLP":
        return False
    if type(ctrl).__name__ == "ClippedCtrl":
        return True
    if type(ctrl).__name__ != "ScoreCtrl" or not (ctrl.score_model is None or type(ctrl.score_model).__name__ == "TimeEmbed"):
        return False
    try:
        E.dist_desc(E.ctrl_target(ctrl)[0], "cpu", [])
    except E.UnsupportedByEngine:
        return False
    return True


def fused_net_view(ctrl):
    """The ClippedCtrl part of a ScoreCtrl (same drift net, same clip) as sdeng_ctrl_vjp wants it; a ClippedCtrl is its own view."""
    if type(ctrl).__name__ == "ClippedCtrl":
        return ctrl
    view = _NET_VIEWS.get(id(ctrl))
    if view is None or view[0]() is not ctrl or view[1].clip_model != ctrl.clip_model:
        import weakref

        from ..models.reparam import ClippedCtrl
        view = (weakref.ref(ctrl), ClippedCtrl(base_model=ctrl.base_model, clip_model=ctrl.clip_model))
        _NET_VIEWS[id(ctrl)] = view
    return view[1]


def vjp_param_grads(ctrl, t_unique, r, N, B):
    """Per-row arrays of ``sdeng_ctrl_vjp`` (include/sdeng.h) -> {parameter: gradient} for a ClippedCtrl over a FourierMLP."""
    net = ctrl.base_model

    def outer(dl, act):
        # dl^T act over all N * B rows.  As ONE GEMM this is 64 x 64 (or d x 64) with K = N * B: hipBLASLt runs it on a handful of
        # workgroups (180 us each at 512 x 100 rows, rocprofv3); batched over the N times and summed it fills the chip (~10 us).
        return torch.bmm(dl.view(N, B, -1).transpose(1, 2), act.view(N, B, -1)).sum(0)
    grads = {net.out_layer.weight: outer(r["dout"], r["a2"]), net.out_layer.bias: r["dout"].sum(0),
             net.hidden_layer[1].weight: outer(r["d2"], r["a1"]), net.hidden_layer[1].bias: r["d2"].sum(0),
             net.hidden_layer[0].weight: outer(r["d1"], r["a0"]), net.hidden_layer[0].bias: r["d1"].sum(0),
             net.input_embed.weight: outer(r["d0"], r["x"]), net.input_embed.bias: r["d0"].sum(0)}
    # time embedding e_t = timestep_embed(t_k): its cotangent is the sum over the particles of d0; the small module itself (2 layers on
    # N rows) is differentiated by torch
    te_params = [p for p in net.timestep_embed.parameters() if p.requires_grad]
    if te_params:
        with torch.enable_grad():
            e = net.timestep_embed(t_unique.view(-1, 1))
            te_grads = torch.autograd.grad(e, te_params, grad_outputs=r["d0"].view(N, B, 64).sum(1), allow_unused=True)
        grads.update({p: g for p, g in zip(te_params, te_grads) if g is not None})
    return grads


class _IntegralPass(torch.nn.Module):
    """s_b = sum_k <u(t_k, x_kb), zc_kb>: the one part of the log-variance loss that carries a graph (BaseOCLoss._lv_loss)."""

    def __init__(self, ctrl):
        super().__init__()
        self.ctrl = ctrl

    def forward(self, t_unique, xs, zc):
        u = ctrl_batched(self.ctrl, t_unique, xs)
        return (u * zc).sum(dim=-1).view(xs.shape[0], xs.shape[1]).sum(dim=0)


class BaseOCLoss:
    """Mirror of losses/oc.py:14-200."""

    kind = None  # coefficient-table kind of the subclass

    def __init__(self, generative_ctrl: Callable, generative_ctrl_ema: Callable, sde=None, method: str = "kl",
                 traj_per_sample: int = 1, filter_samples: Callable | None = None, max_rnd: float | None = None,
                 sde_ctrl_dropout: float | None = None, sde_ctrl_noise: float | None = None, **kwargs):
        self.generative_ctrl = generative_ctrl
        self.generative_ctrl_ema = generative_ctrl_ema
        self.sde = sde
        if method not in ["kl", "kl_ito", "lv", "lv_traj"]:
            raise ValueError("Unknown loss method.")
        self.method = method
        if traj_per_sample == 1 and self.method == "lv_traj":
            raise ValueError("Cannot compute variance over a single trajectory.")
        self.traj_per_sample = traj_per_sample
        self.filter_samples = filter_samples
        self.max_rnd = max_rnd
        self.sde_ctrl_noise = sde_ctrl_noise
        self.sde_ctrl_dropout = sde_ctrl_dropout
        if self.method in ["kl", "kl_ito"]:
            for attr in ["sde_ctrl_noise", "sde_ctrl_dropout"]:
                if getattr(self, attr) is not None:
                    logging.warning("%s should only be used for the log-variance loss.", attr)
        self.n_filtered = 0
        # engine state
        self.seed = 1
        self.particle0 = 0
        self.train_calls = 0
        self._coef_cache = {}
        self._cpu_sde = None
        self.timing_events = None  # optional _lib.HipEvents: times the step-loop kernel alone
        self.graph_adjoint = True  # KL training where the adjoint runs step by step in torch (CMCD; controls / targets / references the adjoint kernel does not cover): each step replayed as a hipGraph; False: eager
        self.native_adjoint = True  # KL training of a ClippedCtrl with no / a diagonal reference: the adjoint recursion as ONE launch (sdeng_kl_adjoint); False: one sdeng_ctrl_vjp per step
        self.fused_training = True  # ClippedCtrl over a FourierMLP: the batched control pass of training as ONE fused HIP forward + backward (sdeng_ctrl_vjp); False: the eager torch pass
        self.graph_training = False  # True: the batched control pass of log-variance training (forward + backward) is replayed as a hipGraph (_IntegralPass)
        self._graphed = {}
        self.split_tiles = False  # True: small batches (<= _lib.SPLIT_TILES_MAX_B = 8 192, the library's own gate) ask for the low-latency kernels (SDENG_FLAG_SPLIT_TILES; fp32-round-off, not bit, equal to the standard kernel); the solvers switch it on (cfg 'split_tiles', default True)
        self.dist = None  # torch.distributed of a sharded run: eval() then returns global estimators and globally normalised weights

    # ---- reference surface -----------------------------------------------------------------
    def filter(self, rnd, samples=None):
        mask = True
        if samples is not None and self.filter_samples is not None:
            mask = self.filter_samples(samples)
        if self.max_rnd is None:
            return mask & rnd.isfinite()
        return mask & (rnd < self.max_rnd)

    def compute_loss(self, rnd, samples=None):
        mask = self.filter(rnd, samples=samples)
        assert mask.shape == rnd.shape
        if self.method == "lv_traj":
            rnd = rnd.reshape(self.traj_per_sample, -1, 1)
            mask = mask.reshape(self.traj_per_sample, -1, 1).all(dim=0)
            self.n_filtered += self.traj_per_sample * (mask.numel() - mask.sum()).item()
            loss = rnd[:, mask].var(dim=0).mean()
        else:
            self.n_filtered += (mask.numel() - mask.sum()).item()
            loss = rnd[mask].var() if self.method == "lv" else rnd[mask].mean()
        return loss, {"train/n_filtered_cumulative": self.n_filtered}

    @staticmethod
    def compute_results(rnd, compute_weights=False, ts=None, samples=None, xs=None, dist=None) -> Results:
        """losses/oc.py:134-173 on the device: one sdeng_logz call gives elbo, logsumexp, variance, weights.
        ``dist`` (an initialised torch.distributed module, set as ``loss.dist`` by a sharded caller): ``rnd`` is this rank's shard;
        the estimators and the importance weights are then those of ALL ranks' particles (one 36-byte all-gather,
        ``parallel.global_weights``), so ``Results.weights`` concatenated over the ranks is the reference's ``softmax(-rnd, 0)``."""
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from .. import parallel
            if compute_weights:
                w, res = parallel.global_weights(rnd, dist)
            else:
                w, res = None, parallel.global_results(rnd, dist)
            metrics = {"eval/elbo": res["elbo"]}
            preds = {}
            if compute_weights:
                preds["log_norm_const_is"] = res["log_norm_const_is"]
                metrics["eval/lv_loss"] = res["lv_loss"]
                metrics["eval/norm_effective_sample_size"] = res["ess"]
            return make_results(samples=samples, weights=w, log_norm_const_preds=preds, ts=ts, xs=xs, metrics=metrics)
        stats, w = E.logz_stats(rnd, want_weights=compute_weights)
        host = stats.cpu()
        metrics = {"eval/elbo": host[0].item()}
        preds = {}
        if compute_weights:
            preds["log_norm_const_is"] = host[1].item()
            metrics["eval/lv_loss"] = host[2].item()
            metrics["eval/norm_effective_sample_size"] = host[3].item()  # eval/metrics.py:135-140
        return make_results(samples=samples, weights=w, log_norm_const_preds=preds, ts=ts, xs=xs, metrics=metrics)

    def __call__(self, ts, x, *args, **kwargs):
        raise E.UnsupportedByEngine(
            f"{type(self).__name__} has no training call: the subclasses implement loss(ts, x, ...) for the log-variance methods "
            "(BaseOCLoss._lv_loss: HIP step loop + one batched autograd pass, SURVEY.md 8f-1)")

    def load_state_dict(self, state_dict: dict):
        self.n_filtered = state_dict["n_filtered"]

    def state_dict(self) -> dict:
        return {"n_filtered": self.n_filtered}

    def _lv_loss(self, ts, x, simulate, terminal, *, lin, coef_kw=None, rnd0=None, ito=True):
        """Log-variance training value (losses/oc.py ``__call__`` of every loss, method in ('lv', 'lv_traj')).  The
        trajectories are driven by the DETACHED control (generative_and_sde_ctrl, :83-103), so the states carry no graph
        and the step loop is exactly the eval path: it runs as one HIP launch (``simulate(x, z)`` with the trajectory and
        the noise kept) and its log-weights ARE the loss's
            rnd = rnd0 + sum_k c_k <u_k, u_k.detach() - u_k/2> + c'_k <u_k, z_k> + terminal(x_N);
        their gradient comes from ONE batched pass of the control over all N*B (time, state) pairs
        (:269-271, :284 EM; :490-491, :499 EI / DDPM-like; :957-958, :965 DIS; :1361-1383 DDS).  KL training differentiates
        through the trajectory and is not on this path."""
        assert self.method in ("lv", "lv_traj")
        if self.sde_ctrl_noise is not None or self.sde_ctrl_dropout is not None:
            raise E.UnsupportedByEngine("sde_ctrl_noise / sde_ctrl_dropout perturb the simulated control (losses/oc.py:97-101): not built "
                                        "(every conf/loss/*.yaml leaves both empty; listed under 'raises' in INTEGRATION.md)")
        # a fresh stream per training call (the reference consumes torch's global generator): call c uses the engine's Philox
        # streams keyed by seed + c * golden-ratio increment -- for the step noise AND for an x0 left to the engine (an InitialDraw
        # materialised with the loss's fixed seed would hand every training step the same batch); call 0 is the eval stream of ``seed``
        E.require_gpu(x)
        seed_c = self._next_train_seed()
        x = self._x0(x, seed_c)
        if self.traj_per_sample != 1:
            x = x.repeat(self.traj_per_sample, 1, 1).reshape(-1, x.shape[-1])
        N, (B, d) = ts.numel() - 1, x.shape
        z = E.philox_noise(seed_c, N, B, d, self.particle0, x.device)
        # VALUE: the step loop already integrates exactly this rnd (the detached control is the control), terminal terms included.
        # (the kernel draws the normals of stream seed_c itself -- bit for bit the z above -- so nothing is injected and small
        # batches may take the low-latency split-tile kernel, which writes the trajectory but does not replay noise)
        seed_eval, self.seed = self.seed, seed_c
        try:
            with torch.no_grad():
                x_n, rnd_sim, xs = simulate(x, None)
        finally:
            self.seed = seed_eval
        with torch.no_grad():
            rnd_val = rnd_sim.reshape(B, 1)
            if rnd0 is not None:
                rnd_val = rnd_val + self._logp(rnd0, x)
        # GRADIENT: d rnd_b / d u_kb = c_k (u.detach() - u) + c'_k z_kb = c'_k z_kb -- the running-cost term has zero gradient at
        # u.detach() = u (in fp32 too: (u - u/2) - u/2 = 0 exactly, which is what the reference's autograd produces).  So the only
        # part of rnd that needs a graph is the stochastic integral s_b = sum_k c'_k <u_kb, z_kb>, from one batched pass of the
        # control; it enters with value zero (s - s.detach()), and compute_loss / backward() do the rest as upstream.
        coef = self._coef(ts, x.device, **(coef_kw or {}))  # coef[:, 0]: the net's time of step k, coef[:, 5]: c'_k
        zc = (z.view(N, B, d) * (coef[:, 5] if ito else torch.zeros_like(coef[:, 5])).view(N, 1, 1)).view(N * B, d)
        s = self._integral_pass(coef[:, 0].contiguous(), xs[:-1], zc)
        return self.compute_loss(rnd_val + (s - s.detach()).view(B, 1), samples=x_n)

    def _integral_pass(self, t_unique, xs, zc):
        """The batched control pass, eager or -- ``graph_training`` -- as a captured graph per (shape, control): at the reference's
        training sizes the ~150 small kernels of its forward + backward are host-launch bound (tools/probe_training.py).  Only for the
        pure-torch controls (ClippedCtrl over a FourierMLP); anything else, or a capture that fails, runs eagerly."""
        # RemoveReferenceCtrl: u = inner - ref_score and ref_score has no parameters, so s - s.detach() of the inner control carries the
        # same (zero) value and the same gradient
        ctrl = E.unwrap_ctrl(self.generative_ctrl)[0] if type(self.generative_ctrl).__name__ == "RemoveReferenceCtrl" else self.generative_ctrl
        if self.fused_training and fused_training_ok(ctrl):
            # the drift net of every RDS / LRDS solver (conf/model/basic.yaml), and the ScoreCtrl of PIS / DDS / DIS (conf/model/score.yaml):
            # fused HIP forward + backward of the net (csrc/grad_kernel.hpp); the score model's N cotangents from the HIP score kernel
            params = [p for p in ctrl.parameters() if p.requires_grad]
            return _FusedIntegral.apply(ctrl, t_unique, xs.contiguous(), zc.view(xs.shape), *params)
        key = (id(ctrl), tuple(xs.shape), str(xs.device))
        fn = self._graphed.get(key)
        if fn is None:
            fn = _IntegralPass(ctrl)
            if self.graph_training and type(ctrl).__name__ == "ClippedCtrl" and type(getattr(ctrl, "base_model", None)).__name__ == "FourierMLP":
                try:
                    sample = (t_unique.detach().clone(), xs.detach().clone(), zc.detach().clone())
                    fn = torch.cuda.make_graphed_callables(fn, sample)
                except Exception as e:  # noqa: BLE001 -- capture is an optimisation: fall back to the eager pass, say so once
                    import warnings
                    warnings.warn(f"log-variance training: graph capture of the batched control pass failed ({type(e).__name__}: {e}); running it eagerly")
                    fn = _IntegralPass(ctrl)
            self._graphed[key] = fn
        return fn(t_unique, xs.contiguous(), zc)

    def _kl_loss(self, ts, x, simulate, terminal, *, lin, coef_kw=None, reference_ctrl=None, ito=True):
        """KL training value and gradient (``method in ('kl', 'kl_ito')``): loss = mean of the log-weights over the particles that pass
        ``filter``, differentiated THROUGH the trajectory (BaseOCLoss.compute_loss losses/oc.py:105-131 on ``simulate(change_sde_ctrl=
        False)``, where the control that drives the SDE is the un-detached control: :258-260, :478-484, :1361-1365).

        VALUE: the HIP step loop -- one launch with the trajectory kept (and the noise of this call's Philox stream, redrawn below).
        GRADIENT: the discrete adjoint of exactly the recursion the kernel integrates (include/sdeng.h, FORM_LIN / FORM_EM with the
        host's per-step coefficients):
            lambda_N = d terminal / d x_N ;   for k = N-1 .. 0:  S_k = <lambda_{k+1}, x_{k+1}(x_k, theta)> + sum_b w_b d rnd_k,b(x_k, theta)
            lambda_k = dS_k/dx_k ,   dL/dtheta += dS_k/dtheta
        with the HIP states x_k as constants -- N vector-Jacobian products of ONE step each (torch autograd on the control module and the
        reference score), no graph through the loop.  This is the same number the reference's ``loss.backward()`` produces (fixtures
        ``train_kl_*``: loss to 6 digits, gradients to fp32 round-off); it is not yet a fused kernel."""
        if self.sde_ctrl_noise is not None or self.sde_ctrl_dropout is not None:
            raise E.UnsupportedByEngine("sde_ctrl_noise / sde_ctrl_dropout perturb the simulated control (losses/oc.py:97-101): not built")
        E.require_gpu(x)
        seed_c = self._next_train_seed()
        x = self._x0(x, seed_c)
        if self.traj_per_sample != 1:
            x = x.repeat(self.traj_per_sample, 1, 1).reshape(-1, x.shape[-1])
        N, (B, d) = ts.numel() - 1, x.shape
        seed_eval, self.seed = self.seed, seed_c
        try:
            with torch.no_grad():
                x_n, rnd_sim, xs = simulate(x)
        finally:
            self.seed = seed_eval
        z = E.philox_noise(seed_c, N, B, d, self.particle0, x.device)  # bit for bit the normals the kernel drew
        rnd_val = rnd_sim.reshape(B, 1)
        mask = self.filter(rnd_val, samples=x_n)
        assert mask.shape == rnd_val.shape
        self.n_filtered += (mask.numel() - mask.sum()).item()
        w = mask.to(rnd_val.dtype) / mask.sum()  # d mean(rnd[mask]) / d rnd_b
        value = rnd_val[mask].mean()
        ctrl = self.generative_ctrl
        params = [p for p in ctrl.parameters() if p.requires_grad]
        grads = [torch.zeros_like(p) for p in params]
        coef = self._coef(ts, x.device, **(coef_kw or {}))
        fused = self.fused_training and type(ctrl).__name__ == "ClippedCtrl" and type(getattr(ctrl, "base_model", None)).__name__ == "FourierMLP"
        with torch.enable_grad():
            xN = x_n.detach().requires_grad_(True)
            lam, = torch.autograd.grad((w * terminal(xN).view(B, 1)).sum(), xN)
            ref_kind = E.resolve_reference(reference_ctrl) if reference_ctrl is not None else ("none", {})
            native = self.native_adjoint and E.adjoint_ctrl_ok(ctrl) and E.diagonal_reference(*ref_kind)
            if native:
                # ClippedCtrl (every RDS / LRDS solver at its defaults) or ScoreCtrl on a diagonal mixture target (DDS / PIS on the mixture
                # benchmarks, BASELINE config 1), no / a diagonal reference: the whole recursion below is ONE launch (sdeng_kl_adjoint:
                # lambda in registers, u recomputed, Hessian-vector products of the mixtures in closed form); the parameter gradients come
                # from the per-row arrays, as in log-variance training, the score model's from its N cotangents.
                arrays, _ = E.kl_adjoint(ctrl, coef, xs[:-1], z if ito else None, w, lam, lin=lin, ito=ito, ref=ref_kind)
                found = vjp_param_grads(ctrl, coef[:, 0].contiguous(), arrays, N, B)
                sm = getattr(ctrl, "score_model", None)
                sm_params = [p for p in sm.parameters() if p.requires_grad] if (arrays["dst"] is not None and sm is not None) else []
                if sm_params:
                    st = ctrl.clipped_score_model(coef[:, 0].contiguous().view(-1, 1), None).view(N)
                    sm_grads = torch.autograd.grad(st, sm_params, grad_outputs=arrays["dst"].sum(1), allow_unused=True)
                    found.update({p: g for p, g in zip(sm_params, sm_grads) if g is not None})
                grads = [found.get(p, torch.zeros_like(p)) for p in params]
                fused = True
            elif fused:
                # ClippedCtrl over the FourierMLP (every RDS / LRDS solver): the control's part of each step's vector-Jacobian product is
                # the fused HIP forward + backward of that time step (sdeng_ctrl_vjp; weights packed once per call), the reference score's
                # part a small torch VJP; the parameter gradients come from the per-row arrays at the end, as in log-variance training.
                sess = E.VjpSession(ctrl, coef[:, 0], xs[:-1])
                u_all = sess.forward_u()
                for k in range(N - 1, -1, -1):
                    c, u, zk = coef[k], u_all[k], z[k]
                    if lin:
                        g = c[2] * lam + w * (2.0 * c[4] * u + (c[5] * zk if ito else 0.0))
                    else:
                        g = (c[2] * c[4]) * lam + w * (c[4] * u + (c[5] * zk if ito else 0.0))
                    gx = sess.step(k, g)
                    jl = None
                    if reference_ctrl is not None:
                        xk = xs[k].detach().requires_grad_(True)
                        jl, = torch.autograd.grad((reference_ctrl(c[0], xk) * lam).sum(), xk)
                    if lin:
                        lam = c[1] * lam + gx if jl is None else c[1] * lam + c[2] * jl + gx
                    else:
                        lam = (1.0 + c[4] * c[1]) * lam + gx if jl is None else (1.0 + c[4] * c[1]) * lam + (c[4] * c[3]) * jl + gx
                found = vjp_param_grads(ctrl, coef[:, 0].contiguous(), sess.arrays(), N, B)
                grads = [found.get(p, torch.zeros_like(p)) for p in params]
            def step(lam_in, x_in, z_in, c, w_in):
                """One step of the adjoint in torch (controls / targets / references the adjoint kernel does not cover)."""
                with torch.enable_grad():
                    xk = x_in.detach().requires_grad_(True)
                    u = ctrl(c[0], xk)
                    ref = reference_ctrl(c[0], xk) if reference_ctrl is not None else None
                    uu, uz = (u * u).sum(-1, keepdim=True), (u * z_in).sum(-1, keepdim=True)
                    if lin:   # x' = c1 x + c2 (u [+ ref]) + c3 z ;  rnd += c4 <u,u> + c5 <u,z>
                        x_next = c[1] * xk + c[2] * (u if ref is None else ref + u) + c[3] * z_in
                        dr = c[4] * uu + (c[5] * uz if ito else 0.0)
                    else:     # x' = x + ((c1 x [+ c3 ref]) + c2 u) c4 + c2 (c5 z) ;  rnd += 0.5 <u,u> c4 + c5 <u,z>
                        drift = c[1] * xk if ref is None else c[1] * xk + c[3] * ref
                        x_next = xk + (drift + c[2] * u) * c[4] + c[2] * (c[5] * z_in)
                        dr = 0.5 * uu * c[4] + (c[5] * uz if ito else 0.0)
                    got = torch.autograd.grad((lam_in * x_next).sum() + (w_in * dr).sum(), [xk] + params, allow_unused=True)
                for acc, gk in zip(grads, got[1:]):
                    if gk is not None:
                        acc.add_(gk)
                return got[0]

            if not (fused or native):
                # launch-bound (~60 small kernels per step): replayed as a hipGraph per (shape, control, form); eager if capture fails
                runner = _graphed_step(self, ("kl", id(ctrl), id(getattr(reference_ctrl, "__self__", reference_ctrl)), B, d, bool(lin), bool(ito), str(x.device)), step, grads,
                                            (lam, xs[0], z[0], coef[0], w)) if (self.graph_adjoint and _capturable(ctrl)) else None
                for k in range(N - 1, -1, -1):
                    args = (lam, xs[k], z[k], coef[k], w)
                    lam = runner(*args) if runner is not None else step(*args)
                if runner is not None:
                    grads = [gr.clone() for gr in runner.grads]
        # hand the gradient to autograd: value + sum <p - p.detach(), dL/dp>  (zero-valued, gradient dL/dp)
        surrogate = sum(((p - p.detach()) * g).sum() for p, g in zip(params, grads))
        return value.detach() + surrogate, {"train/n_filtered_cumulative": self.n_filtered}

    # ---- engine plumbing ---------------------------------------------------------------------
    @staticmethod
    def _logp(fn, x):
        """Gradient-free log-density ``fn(x)`` as [B,1]: one ``sdeng_dist_eval`` launch when ``fn`` is a method of a
        distribution the engine knows (a torch.distributions evaluation costs ~1 ms of host time per call, mostly argument
        validation with device read-backs); the callable itself otherwise."""
        res = E.resolve_logp(fn)
        if res is not None:
            try:
                logp, _ = E.dist_eval(res[0], x, want_logp=True, want_score=False)
                return logp if not res[1] else logp.clip(min=-res[1], max=res[1])  # clipped_target_unnorm_log_prob, solver/oc.py:80-87
            except E.UnsupportedByEngine:
                pass
        return fn(x).view((-1, 1))

    def _ctrl(self, use_ema):
        return self.generative_ctrl_ema if use_ema else self.generative_ctrl

    def _x0(self, x, seed=None):
        """``x`` as a tensor: an ``engine.InitialDraw`` (x0 left to the engine) is materialised with this loss's seed (a training call:
        that call's seed) and shard offset -- the same x0 sdeng_simulate draws itself when handed an InitialDraw."""
        return x.tensor(self.seed if seed is None else seed, self.particle0) if isinstance(x, E.InitialDraw) else x

    def _next_train_seed(self):
        seed_c = (int(self.seed) + 0x9E3779B97F4A7C15 * self.train_calls) & 0xFFFFFFFFFFFFFFFF
        self.train_calls += 1
        return seed_c

    def _sde_cpu(self):
        """CPU copy of the SDE for the per-step scalar tables, rebuilt when a buffer of the SDE changes (load_state_dict, edits)."""
        sig = tuple((b.data_ptr(), b._version) for b in self.sde.buffers()) if isinstance(self.sde, torch.nn.Module) else ()
        if self._cpu_sde is None or self._cpu_sde[0] != sig:
            self._cpu_sde = (sig, E._cpu_sde(self.sde))
            self._coef_cache = {}
        return self._cpu_sde[1]

    def _coef(self, ts, device, **kw):
        # per-step gains of the control wrapper itself (LerpCtrl: g(t) and t/T; CancelDriftCtrl: drift/g and g/2), whatever the loss
        ctrl, _ = E.unwrap_ctrl(self.generative_ctrl)
        cname = type(ctrl).__name__
        if cname in ("LerpCtrl", "CancelDriftCtrl") and kw.get("kind", self.kind) not in ("cmcd", "cmcd_eubo"):
            if cname == "LerpCtrl":
                kw.setdefault("lerp", True)
            else:
                kw.setdefault("cancel", "rescale" if ctrl.use_rescaling else "plain")
            if getattr(self, "_ctrl_sde_cpu", None) is None or self._ctrl_sde_cpu[0] is not ctrl.sde:
                self._ctrl_sde_cpu = (ctrl.sde, E._cpu_sde(ctrl.sde))
            kw["ctrl_sde"] = self._ctrl_sde_cpu[1]
        sde_cpu = self._sde_cpu()  # (may clear the cache)
        # the entry holds `ts` itself: while it is cached its address cannot be handed to another grid, so (address, version) is an
        # identity -- a fresh ts per call with other values can never hit a stale table
        key = (ts.data_ptr(), ts._version, ts.numel(), str(device), tuple(sorted((k, str(v)) for k, v in kw.items())))
        hit = self._coef_cache.get(key)
        if hit is not None and hit[1] is ts:
            return hit[0]
        # another tensor object: compare by CONTENT (one small read-back) before paying for a new table
        kwkey = key[3:]
        ts_cpu = ts.detach().to("cpu", torch.float32)
        for old in self._coef_cache.values():
            if old[3] == kwkey and old[2].shape == ts_cpu.shape and torch.equal(old[2], ts_cpu):
                hit = (old[0], ts, old[2], kwkey)
                break
        else:
            kw2 = dict(kw)
            table = E.coef_table(kw2.pop("kind", self.kind), ts_cpu, sde_cpu, **kw2)
            hit = (table.to(device), ts, ts_cpu, kwkey)
        self._coef_cache = {key: hit}
        return hit[0]

    def _terminal(self, desc, keep, device, terminal_unnorm_log_prob, reference_log_prob):
        """Fill target / ref_dist + flags; returns the callables that stay opaque (evaluated with torch after)."""
        late = []
        res = E.resolve_logp(terminal_unnorm_log_prob) if terminal_unnorm_log_prob is not None else None
        ctrl_tgt, _ = E.ctrl_target(self.generative_ctrl)
        if res is not None:
            try:
                desc.target = E.dist_desc(res[0], device, keep, clip=res[1])
                desc.flags |= L.FLAG_TERM_TARGET
            except E.UnsupportedByEngine:
                res = None
        if res is None and terminal_unnorm_log_prob is not None:
            late.append((-1.0, terminal_unnorm_log_prob))
            if ctrl_tgt is not None:
                desc.target = E.dist_desc(ctrl_tgt, device, keep, score_only=True)
        if reference_log_prob is not None:
            rr = E.resolve_logp(reference_log_prob)
            ok = False
            if rr is not None and desc.flags & L.FLAG_TERM_TARGET:
                try:
                    desc.ref_dist = E.dist_desc(rr[0], device, keep)
                    desc.flags |= L.FLAG_TERM_REF
                    ok = True
                except E.UnsupportedByEngine:
                    pass
            if not ok:
                late.append((1.0, reference_log_prob))
        return late

    @staticmethod
    def _apply_late(rnd, x, late):
        # opaque user callables: evaluated once on the device tensors, like the reference does (losses/oc.py:290)
        if late:
            term = 0.0
            for sign, fn in sorted(late, key=lambda p: -p[0]):
                term = term + sign * fn(x).view((-1, 1))
            rnd += term
        return rnd

    def _simulate(self, ts, x, *, terminal_unnorm_log_prob, reference_log_prob=None, initial_log_prob=None, form,
                  flags, use_ema, return_traj, noise, ref=("none", {}), coef_kw=None):
        E.require_gpu(x)
        if isinstance(x, E.InitialDraw) and (form == L.FORM_EUBO or (initial_log_prob is not None and E.resolve_logp(initial_log_prob) is None)):
            x = self._x0(x)  # an opaque initial log-density needs x0 as a tensor; the noising loops start from data anyway
        device = x.device
        keep = []
        desc = L.Desc()
        desc.form = form
        desc.flags = flags | (L.FLAG_SPLIT_TILES if self.split_tiles and x.shape[0] <= L.SPLIT_TILES_MAX_B else 0)
        desc.N = ts.numel() - 1
        desc.seed = int(self.seed)
        desc.particle0 = int(self.particle0)
        ctrl = self._ctrl(use_ema)
        desc.net = E.net_desc(ctrl, device, keep)
        desc.ref = E.ref_desc(ref[0], ref[1], device, keep)
        _, remove = E.unwrap_ctrl(ctrl)
        if remove is not None:  # RemoveReferenceCtrl: u = ctrl - ref_score, with the reference score the step loop already evaluates
            if ref[0] == "none" or not E.same_reference(E.resolve_reference(remove.ref_score)[1], ref[1]):
                raise E.UnsupportedByEngine("RemoveReferenceCtrl.ref_score must be the loss's own reference_ctrl (the kernel subtracts the "
                                            "reference score it evaluates for the drift)")
            desc.flags |= L.FLAG_REMOVE_REF
        late = self._terminal(desc, keep, device, terminal_unnorm_log_prob, reference_log_prob)
        if desc.target.kind == L.DIST_GMM_FULL and (ref[0] != "none" or form == L.FORM_EUBO):
            raise E.UnsupportedByEngine("a full-covariance mixture target inside a Score / Lerp / CancelDrift control is evaluated in the kernel's "
                                        "reference slot: forward passes of the solvers without a reference drift (PIS / DDS / DIS) only")
        _, lerp_prior = E.ctrl_target(ctrl)
        rnd0 = None
        if initial_log_prob is not None:
            pr = E.resolve_logp(initial_log_prob)
            try:
                if pr is None:
                    raise E.UnsupportedByEngine("opaque")
                desc.prior = E.dist_desc(pr[0], device, keep)
                desc.flags |= L.FLAG_INIT_LOGP
            except E.UnsupportedByEngine:
                rnd0 = initial_log_prob if form == L.FORM_EUBO else initial_log_prob(x).view((-1, 1))  # EUBO: at the noised samples
                if lerp_prior is not None:
                    desc.prior = E.dist_desc(lerp_prior, device, keep)
        elif lerp_prior is not None:
            desc.prior = E.dist_desc(lerp_prior, device, keep)
        coef = self._coef(ts, device, **(coef_kw or {}))
        keep.append(coef)
        desc.coef = coef.data_ptr()
        x_out, rnd, xs = E.run(desc, x, keep, return_traj=return_traj, noise=noise, events=self.timing_events)
        if callable(rnd0):
            rnd0 = rnd0(x_out).view((-1, 1))
        if rnd0 is not None:
            rnd += rnd0
        rnd = self._apply_late(rnd, x if form == L.FORM_EUBO else x_out, late)  # EUBO: the cost at the data, x_in
        assert rnd.shape == (x.shape[0], 1)
        return x_out, rnd, xs

    def _no_train(self, change_sde_ctrl):
        if change_sde_ctrl:
            raise E.UnsupportedByEngine("simulate(change_sde_ctrl=True) is the reference's internal training call: log-variance training "
                                        "runs through loss(ts, x, ...) here (BaseOCLoss._lv_loss: the HIP step loop with the detached "
                                        "control + one batched autograd pass, SURVEY.md 8f-1)")


class EMReferenceSDELoss(BaseOCLoss):
    """losses/oc.py:203-428 (RDS with Euler-Maruyama; PIS when ``reference_ctrl`` is None)."""

    kind = "em"

    def __init__(self, *args, reference_ctrl: Callable | None = None, use_rescaling: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        self.reference_ctrl = reference_ctrl
        self.use_rescaling = use_rescaling

    def simulate(self, ts, x, terminal_unnorm_log_prob, reference_log_prob, change_sde_ctrl=False, return_traj=False,
                 use_ema=False, *, noise=None):
        self._no_train(change_sde_ctrl)
        if not self.use_rescaling and type(self) is EMReferenceSDELoss:
            raise E.UnsupportedByEngine("EMReferenceSDELoss(use_rescaling=False) scales the control twice upstream "
                                        "(losses/oc.py:265-267); not reproduced")
        ref = E.resolve_reference(self.reference_ctrl)
        return self._simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob, reference_log_prob=reference_log_prob,
                              form=L.FORM_EM if self.kind == "em" else L.FORM_LIN, flags=L.FLAG_ITO, use_ema=use_ema,
                              return_traj=return_traj, noise=noise, ref=ref, coef_kw=dict(with_ref=ref[0] != "none"))

    def __call__(self, ts, x, terminal_unnorm_log_prob, reference_log_prob):
        """[TRAINING] losses/oc.py:364-394 (see BaseOCLoss._lv_loss)."""
        def sim(xx, z):
            return self.simulate(ts, xx, terminal_unnorm_log_prob=terminal_unnorm_log_prob, reference_log_prob=reference_log_prob,
                                 change_sde_ctrl=False, return_traj=True, use_ema=False, noise=z)
        with_ref = E.resolve_reference(self.reference_ctrl)[0] != "none"
        if self.method in ("kl", "kl_ito"):  # :364-394 with change_sde_ctrl = False; the Ito term is always part of these losses (:284, :499)
            return self._kl_loss(ts, x, lambda xx: sim(xx, None),
                                 lambda xn: reference_log_prob(xn).view(-1, 1) - terminal_unnorm_log_prob(xn).view(-1, 1),
                                 lin=self.kind != "em", coef_kw=dict(with_ref=with_ref), reference_ctrl=self.reference_ctrl if with_ref else None)
        return self._lv_loss(ts, x, sim, lambda xn: self._logp(reference_log_prob, xn) - self._logp(terminal_unnorm_log_prob, xn),
                             lin=self.kind != "em", coef_kw=dict(with_ref=with_ref))

    def compute_eubo(self, ts, x, terminal_unnorm_log_prob, reference_log_prob, use_ema=False, *, noise=None):
        """losses/oc.py:298-362 (EM; inherited by the DDPM-like loss) and :512-568 (EI): noising trajectories started at
        target samples ``x`` and the density log-ratio along them, as ONE HIP launch (SDENG_FORM_EUBO).  Like the
        reference, ``x`` is noised in place."""
        ref = E.resolve_reference(self.reference_ctrl)
        if ref[0] == "none":
            raise E.UnsupportedByEngine("compute_eubo needs a reference control (RDS losses)")
        kind = "eubo_ei" if self.kind == "ei" else "eubo_em"
        x_out, rnd, _ = self._simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob, reference_log_prob=reference_log_prob,
                                       form=L.FORM_EUBO, flags=0, use_ema=use_ema, return_traj=False, noise=noise, ref=ref,
                                       coef_kw=dict(with_ref=True, kind=kind, rescale=bool(self.use_rescaling)))
        x.copy_(x_out)
        return rnd

    def eval(self, ts, x, terminal_unnorm_log_prob, reference_log_prob=None, compute_weights=True, return_traj=True,
             use_ema=True, *, noise=None) -> Results:
        samples, rnd, xs = self.simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                         reference_log_prob=reference_log_prob, change_sde_ctrl=False,
                                         return_traj=return_traj, use_ema=use_ema, noise=noise)
        return BaseOCLoss.compute_results(rnd, compute_weights=compute_weights, ts=ts, samples=samples, xs=xs, dist=self.dist)


class EIReferenceSDELoss(EMReferenceSDELoss):
    """losses/oc.py:431-568 (RDS with the exponential integrator)."""

    kind = "ei"

    def __init__(self, *args, reference_ctrl: Callable | None = None, **kwargs):
        kwargs.pop("use_rescaling", None)
        super().__init__(*args, reference_ctrl=reference_ctrl, use_rescaling=False, **kwargs)


class DDPMLikeReferenceSDELoss(EMReferenceSDELoss):
    """losses/oc.py:571-651 (RDS with DDPM-like transition kernels)."""

    kind = "ddpm"

    def __init__(self, *args, reference_ctrl: Callable | None = None, **kwargs):
        kwargs.pop("use_rescaling", None)
        super().__init__(*args, reference_ctrl=reference_ctrl, use_rescaling=False, **kwargs)


class _InitialLogProbLoss(BaseOCLoss):
    """Shared eval() of the losses that start from ``initial_log_prob`` (DIS / CMCD families)."""

    def compute_eubo(self, *a, **k):
        raise E.UnsupportedByEngine("no HIP noising loop for this loss's compute_eubo (TimeReversalLoss / ExponentialIntegratorSDELoss; the "
                                    "reference switches EUBO off for PIS and DDS too, solver/oc.py:356,436); the RDS losses', "
                                    "DiscreteTimeReversalLossEI's and ControlledLangevinSDELoss's are HIP launches")

    def eval(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None, compute_weights=True, return_traj=True,
             use_ema=True, *, noise=None) -> Results:
        kw = dict(compute_ito_int=compute_weights) if isinstance(self, TimeReversalLoss) else {}
        samples, rnd, xs = self.simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                         initial_log_prob=initial_log_prob, train=False, return_traj=return_traj,
                                         use_ema=use_ema, noise=noise, **kw)
        return BaseOCLoss.compute_results(rnd, compute_weights=compute_weights, ts=ts, samples=samples, xs=xs, dist=self.dist)


class ControlledLangevinSDELoss(_InitialLogProbLoss):
    """losses/oc.py:654-894 (CMCD)."""

    kind = "cmcd"

    def __init__(self, *args, use_rescaling: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        self.use_rescaling = use_rescaling

    def simulate(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None, train=True, change_sde_ctrl=False,
                 return_traj=False, use_ema=False, *, noise=None, eubo=False):
        self._no_train(change_sde_ctrl)
        if train and self.method in ["kl", "kl_ito"]:
            raise E.UnsupportedByEngine("CMCD KL-training start (rnd0 = 0) is part of the training direction")
        if not self.use_rescaling:
            raise E.UnsupportedByEngine("ControlledLangevinSDELoss(use_rescaling=False) scales the control twice upstream "
                                        "(losses/oc.py:716-719); not reproduced")
        E.require_gpu(x)
        if eubo:
            x = self._x0(x)
        device = x.device
        keep = []
        desc = L.Desc()
        desc.form = L.FORM_CMCD_EUBO if eubo else L.FORM_CMCD
        desc.flags = L.FLAG_ITO | L.FLAG_INIT_LOGP | L.FLAG_TERM_TARGET
        desc.N = ts.numel() - 1
        desc.seed, desc.particle0 = int(self.seed), int(self.particle0)
        desc.net = E.net_desc(self._ctrl(use_ema), device, keep)
        target = getattr(self.sde.target_score, "__self__", None)
        prior = getattr(self.sde.prior_score, "__self__", None)
        if target is None or prior is None:
            raise E.UnsupportedByEngine("ControlledLangevinSDE.target_score / prior_score must be bound Distribution.score methods")
        res = E.resolve_logp(terminal_unnorm_log_prob)
        if res is None or res[0] is not target:
            raise E.UnsupportedByEngine("CMCD: terminal_unnorm_log_prob must be the log-density of sde.target_score's distribution")
        pr = E.resolve_logp(initial_log_prob) if initial_log_prob is not None else None
        if pr is None or pr[0] is not prior:
            raise E.UnsupportedByEngine("CMCD: initial_log_prob must be the log-density of sde.prior_score's distribution")
        desc.target = E.dist_desc(target, device, keep, clip=res[1])
        desc.prior = E.dist_desc(prior, device, keep)
        desc.cmcd_g = E.scalar_of(self.sde.diff_coeff)
        desc.cmcd_clip = E.scalar_of(self.sde.clip_score) if self.sde.clip_score else 0.0
        coef = self._coef(ts, device, kind="cmcd_eubo") if eubo else self._coef(ts, device)
        keep.append(coef)
        desc.coef = coef.data_ptr()
        x_out, rnd, xs = E.run(desc, x, keep, return_traj=return_traj, noise=noise, events=self.timing_events)
        return x_out, rnd, xs

    def compute_eubo(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None, use_ema=False, *, noise=None):
        """losses/oc.py:757-828: the CMCD loop run from target samples with the control subtracted, one HIP launch
        (SDENG_FORM_CMCD_EUBO; diagonal Gaussian / mixture targets -- the ones that can be sampled).  ``x`` is NOT modified
        (upstream rebinds it, :820)."""
        _, rnd, _ = self.simulate(ts, x, terminal_unnorm_log_prob, initial_log_prob=initial_log_prob, train=False, use_ema=use_ema,
                                  noise=noise, eubo=True)
        return rnd


    def __call__(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None):
        """[TRAINING] losses/oc.py:830-857, log-variance methods.  The trajectory is driven by the detached control
        (:704-705, :722), so it is the eval trajectory: one HIP launch (trajectory and noise kept).  The target / prior
        scores along it come from the HIP distribution kernels (no graph needed: the states are constants), and one batched
        autograd pass of the control over the (N+1)*B (time, state) pairs rebuilds (:736-742)
            cost_k = (b_k + b'_k)/g + u_k - u_{k+1},   rnd = rnd0 + sum_k 0.5|cost_k|^2 dt + <cost_k, u_k.detach() - u_k> dt + <cost_k, db_k> - log pi~(x_N)
        with b_k = drift(t_k, x_k), b'_k = drift(t_{k+1}, x_{k+1})."""
        if self.sde_ctrl_noise is not None or self.sde_ctrl_dropout is not None:
            raise E.UnsupportedByEngine("sde_ctrl_noise / sde_ctrl_dropout perturb the simulated control: not built")
        E.require_gpu(x)
        if self.method in ("kl", "kl_ito"):
            return self._kl_loss_cmcd(ts, x, terminal_unnorm_log_prob, initial_log_prob)
        seed_c = self._next_train_seed()
        x = self._x0(x, seed_c)
        if self.traj_per_sample != 1:
            x = x.repeat(self.traj_per_sample, 1, 1).reshape(-1, x.shape[-1])
        N, (B, d) = ts.numel() - 1, x.shape
        z = E.philox_noise(seed_c, N, B, d, self.particle0, x.device)
        target = getattr(self.sde.target_score, "__self__", None)
        prior = getattr(self.sde.prior_score, "__self__", None)
        g, T = float(self.sde.diff_coeff), float(self.sde.terminal_t)
        with torch.no_grad():
            x_n, _, xs = self.simulate(ts, x, terminal_unnorm_log_prob, initial_log_prob=initial_log_prob, train=False,
                                       return_traj=True, use_ema=False, noise=z)
            flat = xs.reshape((N + 1) * B, d)
            _, s_tgt = E.dist_eval(target, flat, want_logp=False)
            _, s_pri = E.dist_eval(prior, flat, want_logp=False)
            s_tgt, s_pri = s_tgt.view(N + 1, B, d), s_pri.view(N + 1, B, d)

            def drift(k_time, k_state):  # eq/sdes.py:101-110 at (ts[k_time], xs[k_state])
                w = (ts[k_time] / T).view(-1, 1, 1)
                out = (s_tgt[k_state] * w + s_pri[k_state] * (1.0 - w)) * (0.5 * g ** 2)
                return out if not self.sde.clip_score else out.clip(-float(self.sde.clip_score), float(self.sde.clip_score))
            idx = torch.arange(N, device=x.device)
            b_s, b_t = drift(idx, idx), drift(idx + 1, idx + 1)
            const = -terminal_unnorm_log_prob(x_n)
            if initial_log_prob is not None:
                const = const + initial_log_prob(x).view((-1, 1))
        u = ctrl_batched(self.generative_ctrl, ts.to(x.device), xs).view(N + 1, B, d)
        dt = (ts[1:] - ts[:-1]).view(N, 1, 1)
        db = dt.sqrt() * z
        cost = (b_s + b_t) / g + u[:-1] - u[1:]
        rnd = (0.5 * (cost ** 2).sum(-1) * dt.view(N, 1) + (cost * (u[:-1].detach() - u[:-1])).sum(-1) * dt.view(N, 1)
               + (cost * db).sum(-1)).sum(0).view(B, 1) + const
        return self.compute_loss(rnd, samples=x_n)


    def _kl_loss_cmcd(self, ts, x, terminal_unnorm_log_prob, initial_log_prob):
        """[TRAINING] KL methods of CMCD (losses/oc.py:830-857 on simulate(train=True): rnd0 = 0, :695-699; the un-detached control drives
        the SDE, :706-709).  VALUE: the HIP step loop (its log-weight minus the initial log-density it adds).  GRADIENT: the discrete
        adjoint of the reference's own step (:711-742) -- y = x + (b_s(x) + g u_s(x)) dt + g db, cost = (b_s(x) + b_t(y))/g + u_s(x) - u_t(y),
        rnd += 0.5 |cost|^2 dt + <cost, db> -- one torch vector-Jacobian product per step over the HIP states (two control evaluations
        share the state y, so the step is differentiated as a whole; the target / prior scores are differentiated exactly where the
        reference's are: closed-form scores carry a graph, autograd-made ones do not, distr/base.py:146-154)."""
        seed_c = self._next_train_seed()
        x = self._x0(x, seed_c)
        if self.traj_per_sample != 1:
            x = x.repeat(self.traj_per_sample, 1, 1).reshape(-1, x.shape[-1])
        N, (B, d) = ts.numel() - 1, x.shape
        seed_eval, self.seed = self.seed, seed_c
        try:
            with torch.no_grad():
                x_n, rnd_sim, xs = self.simulate(ts, x, terminal_unnorm_log_prob, initial_log_prob=initial_log_prob, train=False, return_traj=True)
        finally:
            self.seed = seed_eval
        z = E.philox_noise(seed_c, N, B, d, self.particle0, x.device)  # bit for bit the normals the kernel drew
        rnd_val = rnd_sim.reshape(B, 1) - self._logp(initial_log_prob, x)  # (the evaluation pass starts from log p_prior(x0), training from 0)
        mask = self.filter(rnd_val, samples=x_n)
        assert mask.shape == rnd_val.shape
        self.n_filtered += (mask.numel() - mask.sum()).item()
        w = mask.to(rnd_val.dtype) / mask.sum()
        value = rnd_val[mask].mean()
        ctrl = self.generative_ctrl
        params = [p for p in ctrl.parameters() if p.requires_grad]
        grads = [torch.zeros_like(p) for p in params]
        g = self.sde.diff_coeff
        tdev = ts.to(x.device)
        def step(lam_in, x_in, z_in, s, t, w_in):
            """One step of the adjoint: lambda_k and the parameter-gradient contributions of step k (accumulated into ``grads``)."""
            with torch.enable_grad():
                dt = t - s
                db = dt.sqrt() * z_in
                xk = x_in.detach().requires_grad_(True)
                u_s, b_s = ctrl(s, xk), self.sde.drift(s, xk)
                y = xk + (b_s + u_s * g) * dt + g * db
                cost = (b_s + self.sde.drift(t, y)) / g + u_s - ctrl(t, y)
                dr = 0.5 * (cost ** 2).sum(-1, keepdim=True) * dt + (cost * db).sum(-1, keepdim=True)
                got = torch.autograd.grad((lam_in * y).sum() + (w_in * dr).sum(), [xk] + params, allow_unused=True)
            for acc, gk in zip(grads, got[1:]):
                if gk is not None:
                    acc.add_(gk)
            return got[0]

        with torch.enable_grad():
            xN = x_n.detach().requires_grad_(True)
            lam, = torch.autograd.grad((w * (-terminal_unnorm_log_prob(xN).view(B, 1))).sum(), xN)
        # The ~130 small kernels of one step are launch-bound (2048 x 100: 5 ms per step of the recursion): the step is captured once per
        # (shape, control) as a hipGraph and replayed N times; a capture that fails runs the steps eagerly, and says so.
        runner = _graphed_step(self, ("cmcd", id(ctrl), B, d, str(x.device)), step, grads,
                                    (lam, xs[0], z[0], tdev[0], tdev[1], w)) if (self.graph_adjoint and _capturable(ctrl, self.sde.target_score)) else None
        for k in range(N - 1, -1, -1):
            args = (lam, xs[k], z[k], tdev[k], tdev[k + 1], w)
            lam = runner(*args) if runner is not None else step(*args)
        if runner is not None:
            grads = [gr.clone() for gr in runner.grads]
        surrogate = sum(((p - p.detach()) * gr).sum() for p, gr in zip(params, grads))
        return value.detach() + surrogate, {"train/n_filtered_cumulative": self.n_filtered}



def _capturable(ctrl, *score_fns) -> bool:
    """Can one adjoint step of this control be captured as a hipGraph?  Not when a score in it is the base class's autograd evaluation
    (distr/base.py:146-154: a nested torch.autograd.grad on a freshly flagged leaf -- LogisticRegression): capture refuses it."""
    fns = list(score_fns)
    inner = getattr(ctrl, "score", ctrl)  # RemoveReferenceCtrl wraps the score control
    if hasattr(inner, "target_score"):
        fns.append(inner.target_score)
    return all(type(getattr(f, "__self__", None)).__name__ not in E._GRAPHLESS_SCORE for f in fns)


def _graphed_step(loss, key, step, grads, example):
    """``step`` captured as a hipGraph with static inputs / outputs (torch.cuda.graphs), cached on the loss per key; None if capture is not
    possible (the caller then runs the step eagerly)."""
    import weakref
    cache = loss.__dict__.setdefault("_step_graphs", {})
    ctrl_now = loss.generative_ctrl
    hit = cache.get(key)
    if hit is not None and hit is not False and hit.owner() is not ctrl_now:
        hit = None  # (another control object at a recycled id: the captured graph reads the old one's parameters)
    if hit is None:
        try:
            hit = _GraphedAdjointStep(step, grads, example)
            hit.owner = weakref.ref(ctrl_now)
        except Exception as e:  # noqa: BLE001 -- capture is an optimisation: run the steps eagerly, say so once
            import warnings
            warnings.warn(f"KL training: graph capture of the adjoint step failed ({type(e).__name__}: {e}); running it eagerly")
            hit = False
        cache[key] = hit
    if hit is False:
        return None
    hit.reset()
    return hit


class _GraphedAdjointStep:
    """One adjoint step (forward of the step's formulas + torch.autograd.grad) captured as a hipGraph.  The step function closes over
    per-call tensors (the accumulators); the captured graph keeps its own static accumulators and inputs, refreshed per call."""

    def __init__(self, step, grads, example):
        self.static_in = [t.detach().clone() for t in example]
        self.grads = [torch.zeros_like(g) for g in grads]
        grads_backup = [g.clone() for g in grads]
        self._swap(grads, self.grads)  # the closure accumulates into `grads`: make those our static buffers during warm-up / capture
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step(*self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = step(*self.static_in)
        finally:  # whatever happened, the caller's accumulators get their own storage and their values back
            torch.cuda.synchronize()
            self._unswap(grads, grads_backup)

    def _swap(self, grads, mine):
        self._held = [g.data for g in grads]
        for g, m in zip(grads, mine):
            g.data = m.data

    def _unswap(self, grads, backup):
        for g, h, b in zip(grads, self._held, backup):
            g.data = h
            g.copy_(b)

    def reset(self):
        for g in self.grads:
            g.zero_()

    def __call__(self, *inputs):
        for s, i in zip(self.static_in, inputs):
            s.copy_(i)
        self.graph.replay()
        return self.static_out


class DiscreteTimeReversalLossEI(_InitialLogProbLoss):
    """losses/oc.py:897-1102 (DIS with the exponential integrator)."""

    kind = "dis_ei"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.use_rescaling = False

    def simulate(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None, train=True, change_sde_ctrl=False,
                 return_traj=False, use_ema=False, *, noise=None):
        self._no_train(change_sde_ctrl)
        init = None if (train and self.method in ["kl", "kl_ito"]) else initial_log_prob
        return self._simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob, initial_log_prob=init,
                              form=L.FORM_LIN, flags=L.FLAG_ITO, use_ema=use_ema, return_traj=return_traj, noise=noise)

    def __call__(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None):
        """[TRAINING] losses/oc.py:1038-1066, log-variance methods (rnd0 = log p_prior(x0), terminal -log pi~)."""
        def sim(xx, z):
            return self.simulate(ts, xx, terminal_unnorm_log_prob=terminal_unnorm_log_prob, initial_log_prob=None, train=True,
                                 change_sde_ctrl=False, return_traj=True, use_ema=False, noise=z)
        if self.method in ("kl", "kl_ito"):  # :1038-1066; rnd0 = 0 for the KL methods (:935-939)
            return self._kl_loss(ts, x, lambda xx: sim(xx, None), lambda xn: -terminal_unnorm_log_prob(xn).view(-1, 1), lin=True)
        return self._lv_loss(ts, x, sim, lambda xn: -self._logp(terminal_unnorm_log_prob, xn), lin=True, rnd0=initial_log_prob)

    def compute_eubo(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None, use_ema=False, *, noise=None):
        """losses/oc.py:980-1036: noising trajectories from target samples (no reference; cost 0.5|u|^2 omega, Ito term,
        + log p_prior of the noised samples), one HIP launch.  ``x`` is noised in place like upstream."""
        if initial_log_prob is None:
            raise ValueError("compute_eubo needs initial_log_prob (the reference calls it unconditionally, losses/oc.py:1032)")
        x_out, rnd, _ = self._simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob, initial_log_prob=initial_log_prob,
                                       form=L.FORM_EUBO, flags=0, use_ema=use_ema, return_traj=False, noise=noise,
                                       coef_kw=dict(kind="eubo_ei"))
        x.copy_(x_out)
        return rnd


class TimeReversalLoss(_InitialLogProbLoss):
    """losses/oc.py:1105-1307 (original DIS, without inference control)."""

    kind = "time_reversal"

    def __init__(self, *args, inference_ctrl: Callable | None = None, div_estimator: str | None = None,
                 use_rescaling: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        if not use_rescaling:
            raise ValueError("use_rescaling must be True for TimeReversalLoss.")
        self.inference_ctrl, self.div_estimator, self.use_rescaling = inference_ctrl, div_estimator, use_rescaling

    def simulate(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None, train=True, compute_ito_int=False,
                 change_sde_ctrl=False, return_traj=False, use_ema=False, *, noise=None):
        self._no_train(change_sde_ctrl)
        if self.inference_ctrl is not None:
            raise E.UnsupportedByEngine("a learned inference control needs the divergence of a net (autograd): not on the HIP path")
        init = None if (train and self.method in ["kl", "kl_ito"]) else initial_log_prob
        lerp = type(self.generative_ctrl).__name__ == "LerpCtrl"
        # quirk kept: TimeReversalLoss.simulate never uses the EMA net (losses/oc.py:1177-1180)
        return self._simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob, initial_log_prob=init,
                              form=L.FORM_EM, flags=L.FLAG_ITO if compute_ito_int else 0, use_ema=False,
                              return_traj=return_traj, noise=noise, coef_kw=dict(train=train, dim=x.shape[-1], lerp=lerp))

    def __call__(self, ts, x, terminal_unnorm_log_prob, initial_log_prob=None):
        """[TRAINING] losses/oc.py:1240-1272, log-variance methods, no inference control: cost <u, u.detach() - u/2> dt + <u, db>."""
        lerp = type(self.generative_ctrl).__name__ == "LerpCtrl"
        kl = self.method in ("kl", "kl_ito")
        ito = self.method != "kl"  # :1251 compute_ito_int = self.method != "kl"

        def sim(xx, z):
            return self.simulate(ts, xx, terminal_unnorm_log_prob=terminal_unnorm_log_prob, initial_log_prob=None, train=True,
                                 compute_ito_int=ito, change_sde_ctrl=False, return_traj=True, use_ema=False, noise=z)
        if kl:
            if self.inference_ctrl is not None:
                raise E.UnsupportedByEngine("a learned inference control needs the divergence of a net (autograd): not on the HIP path")
            return self._kl_loss(ts, x, lambda xx: sim(xx, None), lambda xn: -terminal_unnorm_log_prob(xn).view(-1, 1), lin=False,
                                 coef_kw=dict(train=True, dim=x.shape[-1], lerp=lerp), ito=ito)
        return self._lv_loss(ts, x, sim, lambda xn: -self._logp(terminal_unnorm_log_prob, xn), lin=False, rnd0=initial_log_prob,
                             coef_kw=dict(train=True, dim=x.shape[-1], lerp=lerp))


class ExponentialIntegratorSDELoss(BaseOCLoss):
    """losses/oc.py:1310-1467 (DDS)."""

    kind = "dds"

    def __init__(self, *args, alpha: float, sigma: float, **kwargs):
        super().__init__(*args, **kwargs)
        self.alpha, self.sigma = alpha, sigma

    def simulate(self, ts, x, terminal_unnorm_log_prob, reference_log_prob, compute_ito_int=False, change_sde_ctrl=False,
                 return_traj=False, use_ema=False, *, noise=None):
        self._no_train(change_sde_ctrl)
        return self._simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob, reference_log_prob=reference_log_prob,
                              form=L.FORM_LIN, flags=L.FLAG_ITO if compute_ito_int else 0, use_ema=use_ema,
                              return_traj=return_traj, noise=noise, coef_kw=dict(alpha=self.alpha, sigma=self.sigma))

    def __call__(self, ts, x, terminal_unnorm_log_prob, reference_log_prob):
        """[TRAINING] losses/oc.py:1399-1428, log-variance methods (compute_ito_int = True for them)."""
        ito = self.method != "kl"  # :1414 compute_ito_int = self.method != "kl"

        def sim(xx, z):
            return self.simulate(ts, xx, terminal_unnorm_log_prob=terminal_unnorm_log_prob, reference_log_prob=reference_log_prob,
                                 compute_ito_int=ito, change_sde_ctrl=False, return_traj=True, use_ema=False, noise=z)
        if self.method in ("kl", "kl_ito"):
            return self._kl_loss(ts, x, lambda xx: sim(xx, None),
                                 lambda xn: reference_log_prob(xn).view(-1, 1) - terminal_unnorm_log_prob(xn).view(-1, 1), lin=True,
                                 coef_kw=dict(alpha=self.alpha, sigma=self.sigma), ito=ito)
        return self._lv_loss(ts, x, sim, lambda xn: self._logp(reference_log_prob, xn) - self._logp(terminal_unnorm_log_prob, xn), lin=True,
                             coef_kw=dict(alpha=self.alpha, sigma=self.sigma))

    def eval(self, ts, x, terminal_unnorm_log_prob, reference_log_prob=None, compute_weights=True, return_traj=True,
             use_ema=True, *, noise=None) -> Results:
        samples, rnd, xs = self.simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                         reference_log_prob=reference_log_prob, compute_ito_int=compute_weights,
                                         change_sde_ctrl=False, return_traj=return_traj, use_ema=use_ema, noise=noise)
        return BaseOCLoss.compute_results(rnd, compute_weights=compute_weights, ts=ts, samples=samples, xs=xs, dist=self.dist)
