"""Build the golden cases on top of the PRODUCT classes (sde_sampler_lrds_amd.*) -- the GPU parity tests
drive the engine exactly the way a user of the reference's API would."""
from __future__ import annotations


import torch


from sde_sampler_lrds_amd.distr.gauss import GMM, GMMFull, Gauss, GaussFull, IsotropicGauss
from sde_sampler_lrds_amd.distr.logistic_regression import LogisticRegression
from sde_sampler_lrds_amd.distr.phi_four import PhiFour
from sde_sampler_lrds_amd.distr.rings import Rings
from sde_sampler_lrds_amd.eq.sdes import VP, ControlledLangevinSDE, ControlledSDE, LangevinSDE, PinnedBM, ScaledBM
from sde_sampler_lrds_amd.losses import oc
from sde_sampler_lrds_amd.models.mlp import FourierMLP, TimeEmbed
from sde_sampler_lrds_amd.models.reparam import CancelDriftCtrl, ClippedCtrl, LerpCtrl, RemoveReferenceCtrl, ScoreCtrl
from sde_sampler_lrds_amd.reference import MarginalReference


def _mlp(d):
    return FourierMLP(dim=d, activation=torch.nn.GELU(), num_layers=4, channels=64)


def _score_model():
    return TimeEmbed(dim_out=1, activation=torch.nn.GELU(), num_layers=4, channels=64)


def make_sde(m):
    if m.get("sde", "vp") == "pbm":
        return PinnedBM(diff_coeff=m["diff_coeff"], terminal_t=m["T"])
    return VP(m["beta_min"], m["beta_max"], m["sigma"], terminal_t=m["T"])


def build(c, device):
    """-> dict(loss, ts, x0, simulate_kwargs) with every module on ``device``."""
    m, kind, d = c.meta, c.meta["kind"], c.meta["d"]
    out = {}
    if kind == "eubo_gmm":
        m = dict(m, kind="rds_gmm")
        kind = "rds_gmm"
    if kind == "eubo_dis":
        kind = "dis_ei"
    if kind == "eubo_cmcd":
        kind = "cmcd_gmm"
    if kind in ("rds_gmm", "rds_default"):
        sde = make_sde(m)
        target = GMM(dim=d, loc=c["tgt_loc"], scale=c["tgt_scale"], mixture_weights=c["tgt_w"].clone())
        ctrl = ClippedCtrl(base_model=_mlp(d), clip_model=m["clip_model"])
        if m.get("remove_ref"):
            ctrl = CancelDriftCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                                   clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"], sde=sde)
        ctrl.load_state_dict(c.params("ctrl."))
        if kind == "rds_gmm":
            cov_kind = m.get("cov", "diag")
            variances = c["ref_cov"] if cov_kind == "full" else ((c["ref_D"], c["ref_P"]) if cov_kind == "eigen" else c["ref_vars"])
            ref = MarginalReference(sde, "gmm", means_init=c["ref_means"], variances_init=variances, weights_init=c["ref_w"].clone())
        else:
            ref = MarginalReference(sde, "gaussian", x_init=c["ref_x_init"], var_init=c["ref_var_init"])
        cls = {"ei": oc.EIReferenceSDELoss, "ddpm_like": oc.DDPMLikeReferenceSDELoss, "em": oc.EMReferenceSDELoss}[m["integrator"]]
        mods = [sde, target, ctrl, ref]
        for mod in mods:
            mod.to(device)
        if m.get("remove_ref"):
            ctrl = RemoveReferenceCtrl(ctrl, ref, use_rescaling=False)
        loss = cls(ctrl, ctrl, sde=sde, method="kl", reference_ctrl=ref)
        out.update(loss=loss, args=(target.unnorm_log_prob, ref.reference_distr.to(device).log_prob), kwargs={})
    elif kind == "pis_phi4":
        sde = ScaledBM(diff_coeff=m["diff_coeff"], terminal_t=m["T"])
        target = PhiFour(a=m["a"], b=m["b"], dim=d, beta=m["beta"])
        ctrl = ScoreCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                         clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"])
        ctrl.load_state_dict(c.params("ctrl."))
        refd = Gauss(dim=d, loc=c["ref_loc"], scale=c["ref_scale"])
        for mod in (sde, target, ctrl, refd):
            mod.to(device)
        loss = oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
        out.update(loss=loss, args=(target.unnorm_log_prob, refd.log_prob), kwargs={})
    elif kind == "pis_full":
        sde = ScaledBM(diff_coeff=m["diff_coeff"], terminal_t=m["T"])
        target = GMMFull(dim=d, loc=c["tgt_loc"], cov=c["tgt_cov"], mixture_weights=c["tgt_w"].clone())
        ctrl = ScoreCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                         clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"])
        ctrl.load_state_dict(c.params("ctrl."))
        refd = Gauss(dim=d, loc=c["ref_loc"], scale=c["ref_scale"])
        for mod in (sde, target, ctrl, refd):
            mod.to(device)
        loss = oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
        out.update(loss=loss, args=(target.unnorm_log_prob, refd.log_prob), kwargs={})
    elif kind in ("dds", "dds_rings", "dds_full"):
        if kind == "dds_rings":
            target = Rings(dim=2, lower_rad=m["lower_rad"], upper_rad=m["upper_rad"], num_rad=m["num_rad"], scale=m["scale"],
                           n_reference_samples=10)
        elif kind == "dds_full":
            target = GMMFull(dim=d, loc=c["tgt_loc"], cov=c["tgt_cov"], mixture_weights=c["tgt_w"].clone())
        else:
            target = GMM(dim=d, loc=c["tgt_loc"], scale=c["tgt_scale"], mixture_weights=c["tgt_w"].clone())
        prior = IsotropicGauss(dim=d, scale=m["sigma"])
        ctrl = ScoreCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                         clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"])
        ctrl.load_state_dict(c.params("ctrl."))
        for mod in (target, prior, ctrl):
            mod.to(device)
        loss = oc.ExponentialIntegratorSDELoss(ctrl, ctrl, sde=None, method="kl", alpha=m["alpha"], sigma=m["sigma"])
        out.update(loss=loss, args=(target.unnorm_log_prob, prior.log_prob), kwargs=dict(compute_ito_int=True))
    elif kind in ("dis_ei", "dis_orig"):
        sde = make_sde(m)
        target = GMM(dim=d, loc=c["tgt_loc"], scale=c["tgt_scale"], mixture_weights=c["tgt_w"].clone())
        prior = IsotropicGauss(dim=d, scale=1.0)
        if kind == "dis_ei" and m.get("cancel_drift"):
            ctrl = CancelDriftCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                                   clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"], sde=sde)
        elif kind == "dis_ei":
            ctrl = ScoreCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                             clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"])
        else:
            ctrl = LerpCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                            clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"], sde=sde,
                            prior_score=prior.score)
        ctrl.load_state_dict(c.params("ctrl."))
        for mod in (sde, target, prior, ctrl):
            mod.to(device)
        if kind == "dis_ei":
            loss = oc.DiscreteTimeReversalLossEI(ctrl, ctrl, sde=sde, method="kl")
            kwargs = dict(initial_log_prob=prior.log_prob, train=False)
        else:
            loss = oc.TimeReversalLoss(ctrl, ctrl, sde=sde, method="kl", inference_ctrl=None)
            kwargs = dict(initial_log_prob=prior.log_prob, train=False, compute_ito_int=True)
        out.update(loss=loss, args=(target.unnorm_log_prob,), kwargs=kwargs)
    elif kind in ("logreg_pis", "logreg_dds"):
        target = LogisticRegression(c["X"], c["y"], intercept_mean=m["intercept_mean"], intercept_scale=m["intercept_scale"],
                                    weight_scale=m["weight_scale"])
        ctrl = ScoreCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                         clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"])
        ctrl.load_state_dict(c.params("ctrl."))
        if kind == "logreg_pis":
            sde = ScaledBM(diff_coeff=m["diff_coeff"], terminal_t=m["T"])
            refd = Gauss(dim=d, loc=c["ref_loc"], scale=c["ref_scale"])
            for mod in (sde, target, ctrl, refd):
                mod.to(device)
            loss = oc.EMReferenceSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
            out.update(loss=loss, args=(target.unnorm_log_prob, refd.log_prob), kwargs={})
        else:
            prior = IsotropicGauss(dim=d, scale=m["sigma"])
            for mod in (target, prior, ctrl):
                mod.to(device)
            loss = oc.ExponentialIntegratorSDELoss(ctrl, ctrl, sde=None, method="kl", alpha=m["alpha"], sigma=m["sigma"])
            out.update(loss=loss, args=(target.unnorm_log_prob, prior.log_prob), kwargs=dict(compute_ito_int=True))
    elif kind in ("cmcd_gmm", "cmcd_phi4"):
        if kind == "cmcd_phi4":
            target = PhiFour(a=m["a"], b=m["b"], dim=d, beta=m["beta"])
            prior = IsotropicGauss(dim=d, scale=m["prior_scale"])
        else:
            target = GMM(dim=d, loc=c["tgt_loc"], scale=c["tgt_scale"], mixture_weights=c["tgt_w"].clone())
            prior = (IsotropicGauss(dim=d, scale=m["prior_scale"]) if m["prior_kind"] == "iso"
                     else Gauss(dim=d, loc=c["prior_loc"], scale=c["prior_scale_vec"]))
        sde = ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=m["diff_coeff"],
                                    terminal_t=m["T"], clip_score=m["clip_langevin"])
        ctrl = ScoreCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                         clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"])
        ctrl.load_state_dict(c.params("ctrl."))
        for mod in (target, prior, sde, ctrl):
            mod.to(device)
        loss = oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
        out.update(loss=loss, args=(target.unnorm_log_prob,), kwargs=dict(initial_log_prob=prior.log_prob, train=False))
    elif kind == "cmcd_logreg":
        target = LogisticRegression(c["X"], c["y"], intercept_mean=m["intercept_mean"], intercept_scale=m["intercept_scale"],
                                    weight_scale=m["weight_scale"])
        prior = GaussFull(dim=d, loc=c["prior_loc"], cov=c["prior_cov"])
        sde = ControlledLangevinSDE(target_score=target.score, prior_score=prior.score, diff_coeff=m["diff_coeff"],
                                    terminal_t=m["T"], clip_score=m["clip_langevin"])
        ctrl = ScoreCtrl(base_model=_mlp(d), score_model=_score_model(), target_score=target.score, detach_score=False,
                         clip_score=m["clip_score"], clip_model=m["clip_model"], scale_score=m["scale_score"])
        ctrl.load_state_dict(c.params("ctrl."))
        for mod in (target, prior, sde, ctrl):
            mod.to(device)
        loss = oc.ControlledLangevinSDELoss(ctrl, ctrl, sde=sde, method="lv", max_rnd=1e8)
        out.update(loss=loss, args=(target.unnorm_log_prob,), kwargs=dict(initial_log_prob=prior.log_prob, train=False))
    else:
        raise KeyError(kind)
    out["ts"] = c["ts"].to(device)
    out["x0"] = c["x0"].to(device)
    out["loss"].seed = m["seed"]
    return out


def build_euler(c, device):
    """The SDE object of an Euler fixture (tests/golden/gen_golden.py:case_euler) from the product classes."""
    m, d, kind = c.meta, c.meta["d"], c.meta["sde_kind"]
    if kind.startswith("langevin"):
        if kind == "langevin_gmm":
            target = GMM(dim=d, loc=c["tgt_loc"], scale=c["tgt_scale"], mixture_weights=c["tgt_w"].clone())
        elif kind == "langevin_phi4":
            target = PhiFour(a=m["phi_a"], b=m["phi_b"], dim=d, beta=m["phi_beta"])
        else:
            target = Rings(dim=2, lower_rad=m["lower_rad"], upper_rad=m["upper_rad"], num_rad=m["num_rad"], scale=m["scale"],
                           n_reference_samples=10)
        target = target.to(device)
        return LangevinSDE(target_score=target.score, diff_coeff=m["diff_coeff"], clip_score=m["clip_score"], terminal_t=m["T"]).to(device)
    base = make_sde(m).to(device)
    if kind == "controlled_vp":
        ctrl = ClippedCtrl(base_model=_mlp(d), clip_model=m["clip_model"])
        ctrl.load_state_dict(c.params("ctrl."))
        return ControlledSDE(sde=base, ctrl=ctrl.to(device))
    return base
