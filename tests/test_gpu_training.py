"""Log-variance training direction: the HIP step loop + one batched autograd pass of the control must
reproduce the reference's loss value and its gradient w.r.t. every drift-net parameter (fixtures generated from the real
reference by tests/golden/gen_golden.py, same counter-based noise)."""
import pytest
import torch

from sde_sampler_lrds_amd import engine as E
from tests import build_cases as bc
from tests import golden_cases as gc


KINDS = {"train_lv": "rds_gmm", "train_lv_dis": "dis_ei", "train_lv_dis_orig": "dis_orig", "train_lv_dds": "dds", "train_lv_pis": "pis_phi4", "train_lv_cmcd": "cmcd_gmm"}  # objects as in the simulate cases


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["train_lv_ei_gmm_d16", "train_lv_em_gmm_d16", "train_lv_dis_ei_d8", "train_lv_dds_d2", "train_lv_pis_phi4_d100", "train_lv_dis_orig_d8", "train_lv_cmcd_gmm_d16"])
def test_lv_training_loss_and_gradients_match_reference(gpu, name):
    c = gc.load(name)
    c.meta["kind"] = KINDS[c.meta["kind"]]
    b = bc.build(c, gpu)
    loss = b["loss"]
    loss.method = "lv"
    ctrl = loss.generative_ctrl
    for p in ctrl.parameters():
        p.grad = None
    kw = {k: v for k, v in b["kwargs"].items() if k == "initial_log_prob"}
    value, metrics = loss(b["ts"], b["x0"], *b["args"], **kw)
    value.backward()
    # tolerances = 10 x the worst achieved over the seven fixtures on MI355X (loss value 1.2e-6 relative, gradients 5.0e-6 of the largest entry)
    loss_err = abs(float(value.detach()) - c.meta["loss"]) / max(1.0, abs(c.meta["loss"]))
    assert loss_err < 1e-5, (float(value.detach()), c.meta["loss"])
    worst = 0.0
    for k, p in ctrl.named_parameters():
        if "grad." + k not in c.a:
            continue
        ref = c["grad." + k]
        err = float((p.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
        worst = max(worst, err)
        assert err < 5e-5, (k, err)
    print(f"{name}: loss {float(value.detach()):.6f} vs {c.meta['loss']:.6f} (rel {loss_err:.1e}); worst relative gradient error {worst:.2e}")
    assert "train/n_filtered_cumulative" in metrics


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["train_kl_dds_d2", "train_kl_ei_gmm_d16", "train_kl_em_gmm_d16", "train_kl_dis_ei_d8", "train_kl_dis_orig_d8",
                                  "train_kl_pis_phi4_d100"])
def test_kl_training_loss_and_gradients_match_reference(gpu, name):
    """method='kl' -- BaseOCLoss's default, BASELINE cfg 1's loss: back-propagation through the whole trajectory (losses/oc.py:105-131).
    Fixtures: the reference's own ``loss(...)`` + ``backward()`` under the replayed noise.  Here: the HIP trajectory + the discrete
    adjoint (BaseOCLoss._kl_loss)."""
    c = gc.load(name)
    assert c.meta["method"] == "kl"
    c.meta["kind"] = KINDS[c.meta["kind"]]
    b = bc.build(c, gpu)
    loss = b["loss"]
    loss.method, loss.max_rnd = "kl", None
    ctrl = loss.generative_ctrl
    for p in ctrl.parameters():
        p.grad = None
    kw = {k: v for k, v in b["kwargs"].items() if k == "initial_log_prob"}
    value, metrics = loss(b["ts"], b["x0"], *b["args"], **kw)
    value.backward()
    loss_err = abs(float(value.detach()) - c.meta["loss"]) / max(1.0, abs(c.meta["loss"]))
    worst, n = 0.0, 0
    for k, p in ctrl.named_parameters():
        if "grad." + k not in c.a:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        ref = c["grad." + k]
        err = float((p.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
        worst, n = max(worst, err), n + 1
    # tolerance: 5e-5 (10 x the worst achieved on the well-conditioned fixtures), or 10 x the fixture's own conditioning -- how far the
    # REFERENCE's gradient moves when every normal is perturbed by 1.2e-6, recorded by gen_golden.py (train_kl_dds_d2: 1.0e-4, the
    # sharp TwoModes target; the others ~1e-6)
    tol = max(5e-5, 10 * c.meta.get("grad_sensitivity", 0.0))
    print(f"{name}: loss {float(value.detach()):.6f} vs {c.meta['loss']:.6f} (rel {loss_err:.1e}); worst relative gradient error {worst:.2e} over {n} "
          f"parameters (tolerance {tol:.1e}; reference's own sensitivity {c.meta.get('grad_sensitivity', 0.0):.1e})")
    assert n >= 8 and loss_err < 1e-5 and worst < tol
    assert "train/n_filtered_cumulative" in metrics


@pytest.mark.gpu
def test_cmcd_kl_training_is_refused(gpu):
    """The one KL path without an adjoint here: ControlledLangevinSDELoss (two control evaluations per step share the state)."""
    c = gc.load("train_lv_cmcd_gmm_d16")
    c.meta["kind"] = KINDS[c.meta["kind"]]
    b = bc.build(c, gpu)
    b["loss"].method = "kl"
    with pytest.raises(E.UnsupportedByEngine):
        b["loss"](b["ts"], b["x0"], *b["args"], initial_log_prob=b["kwargs"]["initial_log_prob"])


@pytest.mark.gpu
def test_training_steps_reduce_the_loss(gpu):
    """make_model -> TrainableWrapper.run(): a few hundred Adam steps of log-variance training on a small mixture."""
    from sde_sampler_lrds_amd.additions.hacking import TrainableWrapper
    from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details
    tgt = make_target_details("many_modes", dim=8, n_modes=4)
    K, d = 4, 8
    model = make_model("vp-ref", "default", "lv", "ei", "base_zero_init", "uniform", dict(sigma=2.0), tgt,
                       dict(train_steps=150, train_batch_size=512, eval_batch_size=2048), optim_details=dict(lr=3e-3), n_steps=32)
    before = model.evaluate().metrics["eval/lv_loss"]
    res, train = TrainableWrapper(model, verbose=False).run(keep_training_metrics=True)
    after = res.metrics["eval/lv_loss"]
    print(f"eval/lv_loss {before:.3f} -> {after:.3f} after {len(train['train/loss'])} steps; eubo {res.metrics.get('eval/eubo')}")
    assert len(train["train/loss"]) == 150 and after < 0.7 * before


@pytest.mark.gpu
def test_ema_weights_are_used_for_evaluation(gpu):
    from sde_sampler_lrds_amd.experiments.benchmark_utils import make_model, make_target_details
    tgt = make_target_details("many_modes", dim=8, n_modes=4)
    model = make_model("vp-ref", "default", "lv", "ei", "base_zero_init", "uniform", dict(sigma=2.0), tgt,
                       dict(train_steps=20, train_batch_size=256, eval_batch_size=512), optim_details=dict(lr=3e-3), n_steps=16, use_ema=True)
    assert type(model.generative_ctrl_ema).__name__ == "AveragedModel"
    model.run()
    w, w_ema = model.generative_ctrl.base_model.out_layer.weight, model.generative_ctrl_ema.module.base_model.out_layer.weight
    assert not torch.equal(w, w_ema) and int(model.generative_ctrl_ema.n_averaged) >= 2
    a = model.evaluate(use_ema=True).metrics["eval/elbo"]
    b = model.evaluate(use_ema=False).metrics["eval/elbo"]
    assert a != b  # the EMA copy and the live net drive different samplers


@pytest.mark.gpu
@pytest.mark.parametrize("d,N,B,clip", [(128, 7, 37, 1e4), (100, 5, 64, 0.05), (16, 9, 20, None), (2, 3, 5, 1e4), (61, 4, 48, 0.03)])
def test_fused_forward_backward_of_the_drift_net_matches_autograd(gpu, d, N, B, clip):
    """sdeng_ctrl_vjp (csrc/grad_kernel.hpp): per-row activations / cotangents of the drift net, the six parameter-gradient products built
    from them and the state gradient, against torch autograd of the same ClippedCtrl in fp64.  Rows per time not a multiple of the
    16-row tile, an ACTIVE clip (its mask must gate the cotangent exactly like torch.clip's backward), small cotangents (1e-6)."""
    import copy

    from sde_sampler_lrds_amd.experiments import baseline_configs as cfgs
    from sde_sampler_lrds_amd.models.reparam import ClippedCtrl
    torch.manual_seed(d + N)
    ctrl = ClippedCtrl(base_model=cfgs._net(d), clip_model=clip).to(gpu)
    ts = torch.linspace(0.05, 0.9, N, device=gpu)
    xs = 1.5 * torch.randn(N, B, d, device=gpu)
    cot = 1e-6 * torch.randn(N, B, d, device=gpu) * (1.0 + 10.0 * torch.rand(N, B, 1, device=gpu))
    r = E.ctrl_vjp(ctrl, ts, xs, cot, want_gx=True)
    net = ctrl.base_model
    got = {"out_layer.weight": r["dout"].t() @ r["a2"], "out_layer.bias": r["dout"].sum(0), "hidden_layer.1.weight": r["d2"].t() @ r["a1"],
           "hidden_layer.1.bias": r["d2"].sum(0), "hidden_layer.0.weight": r["d1"].t() @ r["a0"], "hidden_layer.0.bias": r["d1"].sum(0),
           "input_embed.weight": r["d0"].t() @ r["x"], "input_embed.bias": r["d0"].sum(0)}
    # fp64 autograd of the same module
    c64 = copy.deepcopy(ctrl).double()
    x64 = xs.double().requires_grad_(True)
    u = torch.stack([c64(ts[k].double(), x64[k]) for k in range(N)])
    (u * cot.double()).sum().backward()
    worst = 0.0
    for k, v in got.items():
        ref = dict(c64.base_model.named_parameters())[k].grad
        err = float((v.double() - ref).abs().max() / ref.abs().max().clamp(min=1e-30))
        worst = max(worst, err)
        assert err < 2e-5, (k, err)
    egx = float((r["gx"].view(N, B, d).double() - x64.grad).abs().max() / x64.grad.abs().max())
    with torch.no_grad():  # outputs of the un-clipped net beyond the clip: where torch.clip's backward blocks the cotangent
        n_clipped = int(sum((c64.base_model(ts[k].double(), x64[k]).abs() > clip).sum() for k in range(N))) if clip else 0
    print(f"ctrl_vjp d={d} N={N} B={B} clip={clip}: worst parameter-gradient error {worst:.2e}, state gradient {egx:.2e} (vs fp64 autograd; {n_clipped} clipped outputs)")
    assert egx < 2e-5
    if clip is not None and clip < 1.0:
        assert n_clipped > 0
