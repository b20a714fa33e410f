"""``make_model`` is the drop-in boundary: it must accept and reject exactly what the reference's does
(experiments/benchmark_utils.py:96-160, :260-262).  ``tests/golden/make_model_grid.json`` holds the outcome of the REFERENCE's own
checks for all 13 824 argument combinations (tests/golden/gen_make_model_grid.py runs the reference's statements); the mirror is held
to it entry by entry.  CPU only: nothing here launches a kernel."""
import itertools
import json
import os

import pytest
import torch

from sde_sampler_lrds_amd import engine as E
from sde_sampler_lrds_amd.experiments import benchmark_utils as bu
from sde_sampler_lrds_amd.models import reparam

GRID = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "make_model_grid.json")))
AXES = GRID["axes"]
COMBOS = [dict(zip(AXES, c)) for c in itertools.product(*AXES.values())]
EXPECTED = [GRID["outcomes"][ord(ch) - ord("a")] for ch in GRID["index"]]


def test_grid_is_the_full_product():
    assert len(COMBOS) == len(EXPECTED) == 13824
    assert list(AXES["solver_type"]) == list(bu.solver_types) and list(AXES["model_type"]) == list(bu.model_types)


def test_validation_matches_the_reference_on_every_combination():
    wrong = []
    for kw, want in zip(COMBOS, EXPECTED):
        try:
            bu.validate_make_model_args(**kw)
            got = "ok"
        except ValueError as e:
            got = "ValueError: " + str(e)
        if got != want:
            wrong.append((kw, want, got))
    assert not wrong, f"{len(wrong)} combinations differ from the reference, first: {wrong[0]}"


def _details(dim, k=3):
    g = torch.Generator().manual_seed(0)
    return dict(sigma=1.3, mean_ref=torch.randn(dim, generator=g), var_ref=torch.rand(dim, generator=g) + 0.5,
                weights_ref=torch.ones(k) / k, means_ref=torch.randn(k, dim, generator=g), variances_ref=torch.rand(k, dim, generator=g) + 0.5,
                mean=torch.zeros(dim), var=torch.ones(dim))


TRAIN = dict(train_steps=10, train_batch_size=8, eval_batch_size=16)


def test_every_accepted_combination_builds_or_says_why_not():
    """Of the combinations the reference accepts, the mirror builds a solver (on the CPU: construction launches nothing) or raises
    UnsupportedByEngine -- exactly for the UNet drift nets and the 'nn' references, which have no kernel."""
    dim = 4
    target = bu.make_target_details("many_modes", dim=dim, n_modes=3)
    built = unsupported = 0
    for kw, want in zip(COMBOS, EXPECTED):
        if want != "ok" or kw["loss_type"] == "kl" and kw["force_vp20"]:  # (halve the work: vp20 x kl adds nothing over vp20 x lv)
            continue
        no_kernel = "unet" in kw["model_type"] or (kw["ref_type"] == "nn" and "ref" in kw["solver_type"])
        try:
            model = bu.make_model(**kw, solver_details=_details(dim), target_details=target, training_details=TRAIN, n_steps=4, device="cpu")
        except E.UnsupportedByEngine:
            assert no_kernel, kw
            unsupported += 1
            continue
        assert not no_kernel, kw
        built += 1
        wrapped = kw["model_type"] == "target_informed_langevin_init" and "ref" in kw["solver_type"]
        assert isinstance(model.generative_ctrl, reparam.RemoveReferenceCtrl) == wrapped, kw
    assert built > 300 and unsupported > 100, (built, unsupported)


@pytest.mark.parametrize("kw,message", [
    (dict(solver_type="dds_orig", model_type="base_zero_init"), "Only target_informed_zero_init model is supported."),
    (dict(solver_type="pis_orig", model_type="target_informed_lerp_tempering"), "Only target_informed_zero_init model is supported."),
    (dict(solver_type="dis_orig", model_type="base_zero_init"), "Model base_zero_init is not supported."),
    (dict(solver_type="cmcd", model_type="base_zero_init"), "Only base_zero_init is supported for CMCD."),
    (dict(solver_type="dds_orig", force_vp20=True), "Can't use vp_20 for orig models other than DIS."),
    (dict(solver_type="dis_orig", model_type="target_informed_lerp_tempering", force_vp_cosine=True), "Can't use vp_cosine for orig models."),
    (dict(solver_type="vp-ref", model_type="target_informed_lerp_tempering"), "Model target_informed_lerp_tempering is not supported."),
    (dict(solver_type="pbm-ref", time_type="snr", force_vp20=True), "Can't use vp_20 or vp_cosine with PBM."),
    (dict(solver_type="cmcd", ref_type="gmm"), "Can't use ref other than gaussian for CMCD."),
    (dict(solver_type="vp-ref", model_type="target_informed_langevin_init", integrator_type="ei"), "Can't use EI or DDPM-like with Langevin score."),
])
def test_make_model_itself_raises_the_reference_message(kw, message):
    """The exceptions the round-2 mirror had dropped (VERDICT weak #11), raised by make_model itself with upstream's text."""
    args = dict(solver_type="vp-ref", ref_type="default", loss_type="lv", integrator_type="em", model_type="target_informed_zero_init",
                time_type="uniform")
    args.update(kw)
    with pytest.raises(ValueError) as e:
        bu.make_model(**args, solver_details=_details(4), target_details=bu.make_target_details("many_modes", dim=4),
                      training_details=TRAIN, device="cpu")
    assert str(e.value) == message


def test_langevin_init_on_a_reference_solver_is_wired_like_upstream():
    """benchmark_utils.py:260-262 rebinds ``model.generative_ctrl`` to a RemoveReferenceCtrl AFTER the loss was constructed
    (solver/oc.py:504-511), so upstream the loss keeps simulating with the un-wrapped CancelDriftCtrl; only the solver's own attribute
    (checkpoints, EMA source, direct calls) sees the wrapper -- and a direct call fails (use_rescaling=True needs the sde the constructor
    refuses, models/reparam.py:52-60).  Same object graph here."""
    model = bu.make_model("vp-ref", "gmm", "lv", "em", "target_informed_langevin_init", "uniform", _details(4),
                          bu.make_target_details("many_modes", dim=4), TRAIN, n_steps=4, device="cpu")
    wrap = model.generative_ctrl
    assert isinstance(wrap, reparam.RemoveReferenceCtrl) and wrap.use_rescaling and wrap.sde is None
    assert isinstance(wrap.score, reparam.CancelDriftCtrl) and wrap.ref_score is model.reference_score_t
    assert model.loss.generative_ctrl is wrap.score and model.loss.generative_ctrl_ema is wrap.score
    assert all(k.startswith("score.") for k in model.state_dict()["generative_ctrl"])
    assert {id(p) for p in model.trainable_parameters()} == {id(p) for p in wrap.score.parameters() if p.requires_grad}
    with pytest.raises(AttributeError):
        wrap(torch.tensor(0.5), torch.zeros(2, 4))
    # the form upstream CAN evaluate is a plain subtraction
    plain = reparam.RemoveReferenceCtrl(wrap.score, wrap.ref_score, use_rescaling=False)
    t, x = torch.tensor(0.5), torch.randn(3, 4)
    with torch.no_grad():
        assert torch.equal(plain(t, x), wrap.score(t, x) - wrap.ref_score(t, x))


def test_engine_caches_do_not_leak_into_the_reference_utils():
    """the shared-variance / eigendecomposition caches lived in the caller's ``reference_distr_utils`` dict, which
    ``RDS.state_dict()`` iterates and ``MarginalReference.to()`` maps ``.to`` over.  They are private to the engine now."""
    dim = 4
    model = bu.make_model("vp-ref", "gmm", "lv", "ei", "base_zero_init", "uniform", _details(dim),
                          bu.make_target_details("many_modes", dim=dim), TRAIN, n_steps=4, device="cpu")
    keys_before = set(model.reference_distr_utils)
    for _ in range(2):  # second call: served from the cache
        r = E.ref_desc(*E.resolve_reference(model.loss.reference_ctrl), "cpu", [])
        assert r.shared_var == 0
    cov = torch.diag_embed(_details(dim)["variances_ref"])
    model.change_reference_type("gmm", weights=torch.ones(3) / 3, means=_details(dim)["means_ref"], variances=cov)
    for _ in range(2):
        assert E.ref_desc(*E.resolve_reference(model.loss.reference_ctrl), "cpu", []).kind == E.L.REF_GMM_FULL
    assert set(model.reference_distr_utils) == keys_before == {"means_init", "variances_init", "weights_init"}
    model._reference.to("cpu")  # crashed with "'tuple' object has no attribute 'to'" when the cache tuple sat in the dict
    model.to("cpu")
    assert set(model.state_dict()) == {"generative_ctrl", "loss", "ref_means_init", "ref_variances_init", "ref_weights_init", "ref_type"}  # solver/oc.py:637
    # a version bump invalidates the cached fact
    var = torch.ones(3, dim)
    utils = dict(means_init=torch.zeros(3, dim), variances_init=var, weights_init=torch.ones(3))
    assert E.ref_desc("gmm", utils, "cpu", []).shared_var == 1
    var[1, 2] = 2.0
    assert E.ref_desc("gmm", utils, "cpu", []).shared_var == 0
