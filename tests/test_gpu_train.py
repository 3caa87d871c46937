"""GPU: one training iteration (mtk_train_step / mtk_train_apply) against the oracle's
autograd restatement and the reference trainer's captured run (tests/golden/train_*.npz).

Bars:
  * gradient of every parameter tensor: ||g - g_ref||_2 <= 2e-4 ||g_ref||_2 (fp32 MFMA accumulation order and fp32
    atomic accumulation of the split batch reduction differ from torch-CPU's; the loss-side gradient is the analytic
    derivative with the exact per-block log-sum-exp, the reference differentiates through its cumulative one);
  * AdamW + EMA kernel on given (p, g, m, v): 2e-6 relative;
  * 12 iterations replaying the reference's draws, teacher-forced (every iteration starts from the reference
    trajectory's weights and Adam moments): loss within 1e-4 relative + 3 conditioning floors, gradient within
    max(2e-4, 3x the reference's own sensitivity to a 1e-6 relative weight perturbation (the size of fp32 GEMM summation-order differences)) in L2;
  * the same 12 iterations free-running: each loss mean within 1e-2 relative and the trained weights within
    a few learning-rate quanta (Adam's first updates are lr * sign(g): a gradient entry whose sign is decided by
    rounding moves a weight by up to 2 lr, so no tighter bound on individual weights is well-posed).
"""
import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as orc
from tests.test_oracle_train import flatten, load_train

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def close(got, want, rtol, atol):
    got, want = got.double().cpu(), want.double().cpu()
    assert bool(((got - want).abs() <= rtol * want.abs() + atol).all()), float(((got - want).abs() / (want.abs() + atol)).max())


def build_trainer(c, params, lr=2e-3, wd=1e-4, steps=12):
    import midst_toolkit_amd as mtk

    net = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(c["d_layers"]), 0.0))
    net.load_state_dict(params, strict=True)
    d = mtk.GaussianMultinomialDiffusion(np.asarray(c["K"]), c["Nn"], net, num_timesteps=c["T"], device=dev()).to(dev())
    return mtk.ClavaDDPMTrainer(d, iter(()), lr, wd, steps, device=dev()), d


def named(tensors, n_layers):
    out = {}
    for name, (W, b) in zip(orc.mlp_param_names(n_layers), tensors):
        out[name + ".weight"], out[name + ".bias"] = W, b
    return out


def export(tr, image):
    net = tr.diffusion_model._denoise_fn
    bufs = [(torch.empty_like(W.data), torch.empty_like(b.data)) for W, b in net._weights()]
    tr._engine.train_export(image, bufs)
    return named([(W.cpu(), b.cpu()) for W, b in bufs], len(net.mlp.blocks))


def big_batch(c, B, seed=5):
    g = torch.Generator().manual_seed(seed)
    K, Nn = c["K"], c["Nn"]
    cols = [torch.randn((B, Nn), generator=g).clamp(-5.2, 5.2)]
    S = 0
    if K[0] != 0:
        cols += [torch.randint(0, k, (B, 1), generator=g).float() for k in K]
        S = sum(K)
    x0 = torch.cat(cols, 1)
    t = torch.randint(0, c["T"], (B,), generator=g)
    t[:3] = torch.tensor([0, 1, c["T"] - 1])
    return x0, t, torch.randn((B, Nn), generator=g), torch.rand((B, S), generator=g)


@pytest.mark.parametrize("tag,B,layers", [("B_ragged", 24, None), ("A_numeric", 24, None), ("B_ragged", 1000, [256, 128, 256]),
                                          ("M", 4096, [512] * 4), ("M", 16384, [1024, 1024]),  # 128 x 128 tiles in wgrad too
                                          ("M", 16384, [1024] * 4),  # BASELINE configs[4]: 4 x 1024, global batch 16,384
                                          ("M", 2048, [1024] * 4)])  # ... and the 2,048 rows one of 8 ranks sees
def test_gradients_match_autograd(tag, B, layers):
    if tag == "M":
        c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=layers)
    else:
        c = load_train(tag)
        if layers:
            c["d_layers"] = layers
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    x0, t, eps, u = c["batches"][0] if (B == 24) else big_batch(c, B)
    s = orc.make_schedule(c["T"], "cosine")
    want, lm_ref, lg_ref = orc.loss_and_grads(s, c["K"], c["Nn"], p0, x0, t, eps, u)
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    has_cat = c["K"][0] != 0
    lm = torch.empty(B, device=dev()) if has_cat else None
    lg = torch.empty(B, device=dev())
    eng.train_step(x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()) if has_cat else None, 1.0 / B, lm, lg)
    got = export(tr, "grads")
    assert abs(float(lg.mean()) - float(lg_ref)) <= 1e-5 * abs(float(lg_ref)) + 1e-6
    if has_cat:
        # small batches do not average the reference's own rounding noise out: allow its conditioning floor
        xt, idx = orc.q_sample_fused(s, c["K"], c["Nn"], x0, t, eps, u)
        out = orc.mlp_forward(p0, orc.model_input(xt, idx, c["K"]), t)
        floor = float(orc.multinomial_loss_floor(s, c["K"], c["Nn"], out, x0, idx, t).mean())
        assert abs(float(lm.mean()) - float(lm_ref)) <= 1e-4 * abs(float(lm_ref)) + 3 * floor
    for k in want:
        num = float((got[k].double() - want[k].double()).norm())
        den = float(want[k].double().norm())
        assert num <= 2e-4 * den + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e}"


@pytest.mark.parametrize("tag", ["B_ragged", "A_numeric"])
def test_kl_gaussian_loss_gradients(tag):
    """GaussianLossType.KL in the training iteration (mtk_train_set_gaussian_loss; GMD:431-484, 534-541): per-record bits/dim and the
    gradient of their batch mean through the decoder NLL (t = 0, all three bins of the discretised likelihood) and the KL branch,
    against the oracle's autograd through its fp64 restatement of _vb_terms_bpd (which is bit-exact against the reference's KL
    fixtures).  Also through the autograd seam: mixed_loss(...).backward() on a KL-configured drop-in."""
    import midst_toolkit_amd as mtk

    c = dict(K=[3, 7, 10, 2, 12], Nn=6, T=1000, D=40, d_layers=[128, 128]) if tag == "B_ragged" else dict(K=[0], Nn=8, T=100, D=8, d_layers=[64, 64])
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    B = 768
    x0, t, eps, u = big_batch(c, B)
    t[:200] = 0                       # plenty of decoder-NLL rows ...
    x0[:40, 0] = -1.5                 # ... in the lower open bin,
    x0[40:80, 1] = 1.5                # the upper open bin,
    x0[80:200, : c["Nn"]] *= 0.3      # and the interior
    has_cat = c["K"][0] != 0
    s = orc.make_schedule(c["T"], "cosine")
    want, lm_ref, lg_ref = orc.loss_and_grads(s, c["K"], c["Nn"], p0, x0, t, eps, u, gaussian_loss="kl")
    # conditioning: at small t the bound's 1 / variance factor (up to ~1e5 at T = 1000) multiplies an fp32 difference of O(1)
    # numbers, so the REFERENCE's own gradient moves by ~2e-3 relative under a 1e-7 perturbation of the weights (T = 1000;
    # ~1e-6 at T = 100).  The bar is 3e-4 + 3 x that movement, measured here per tensor.
    g7 = torch.Generator().manual_seed(0)
    p1 = {k: (v * (1 + 1e-7 * torch.randn(v.shape, generator=g7))).float() for k, v in p0.items()}
    moved, _, _ = orc.loss_and_grads(s, c["K"], c["Nn"], p1, x0, t, eps, u, gaussian_loss="kl")
    floor = {k: float((moved[k].double() - want[k].double()).norm()) for k in want}
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    eng.train_set_gaussian_loss("kl")
    lm = torch.empty(B, device=dev()) if has_cat else None
    lg = torch.empty(B, device=dev())
    eng.train_step(x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()) if has_cat else None, 1.0 / B, lm, lg)
    got = export(tr, "grads")
    rows = orc.mixed_loss(s, c["K"], c["Nn"], p0, x0, t, eps, u, gaussian_loss="kl")[1]
    close(lg, rows.float(), rtol=2e-4, atol=1e-5)
    assert abs(float(lg.mean()) - float(lg_ref)) <= 1e-4 * abs(float(lg_ref))
    for k in want:
        num, den = float((got[k].double() - want[k].double()).norm()), float(want[k].double().norm())
        assert num <= 3e-4 * den + 3 * floor[k] + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e} (floor {floor[k] / max(den, 1e-30):.3e})"
    # the autograd seam on a KL-configured drop-in
    net = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(c["d_layers"]), 0.0))
    net.load_state_dict(p0, strict=True)
    dk = mtk.GaussianMultinomialDiffusion(np.asarray(c["K"]), c["Nn"], net, num_timesteps=c["T"], gaussian_loss_type=mtk.GaussianLossType.KL,
                                          device=dev()).to(dev())
    dk._draw_loss_noise = lambda f: (t.to(dev()), eps.to(dev()), u.to(dev()) if has_cat else None)
    l_m, l_g = dk.mixed_loss(x0.to(dev()), {"y": torch.zeros(B, dtype=torch.long, device=dev())})
    (l_m + l_g).backward()
    assert abs(float(l_g.detach()) - float(lg_ref)) <= 1e-4 * abs(float(lg_ref))
    for k, v in dk._denoise_fn.named_parameters():
        num, den = float((v.grad.cpu().double() - want[k].double()).norm()), float(want[k].double().norm())
        assert num <= 3e-4 * den + 3 * floor[k] + 1e-9, f"autograd {k}: relative L2 error {num / max(den, 1e-30):.3e}"


@pytest.mark.parametrize("gaussian_loss", ["mse", "kl"])
@pytest.mark.parametrize("math", ["reference", "fast"])
def test_x0_and_direct_parametrisations_train(gaussian_loss, math):
    """GaussianParametrization.X0 + Parametrization.DIRECT (GMD:399-408, 822-832) in the training iteration: losses and every
    gradient tensor against the oracle's autograd through its restatement of both branches (bit-exact against the reference's own
    fixtures, tests/test_oracle_golden.py).  DIRECT: d KL / d logits through the plain log-softmax; X0 with the KL Gaussian loss:
    d mean / d output = posterior_mean_coef1 (with the MSE loss the Gaussian parametrisation does not enter, GMD:531-532).  Both
    math modes of the loss-gradient kernel; bar as in test_kl_gaussian_loss_gradients (the reference's own movement under a 1e-7
    weight perturbation is measured and allowed three times over)."""
    import midst_toolkit_amd as mtk

    c = dict(K=[3, 7, 10, 2, 12], Nn=6, T=1000, D=40, d_layers=[128, 128])
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    B = 768
    x0, t, eps, u = big_batch(c, B)
    t[:150] = 0
    x0[:40, 0], x0[40:80, 1] = -1.5, 1.5
    s = orc.make_schedule(c["T"], "cosine")
    kw = dict(gaussian_loss=gaussian_loss, x0_param=True, direct=True)
    want, lm_ref, lg_ref = orc.loss_and_grads(s, c["K"], c["Nn"], p0, x0, t, eps, u, **kw)
    g7 = torch.Generator().manual_seed(0)
    p1 = {k: (v * (1 + 1e-7 * torch.randn(v.shape, generator=g7))).float() for k, v in p0.items()}
    moved, _, _ = orc.loss_and_grads(s, c["K"], c["Nn"], p1, x0, t, eps, u, **kw)
    floor = {k: float((moved[k].double() - want[k].double()).norm()) for k in want}
    net = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(c["d_layers"]), 0.0))
    net.load_state_dict(p0, strict=True)
    d = mtk.GaussianMultinomialDiffusion(np.asarray(c["K"]), c["Nn"], net, num_timesteps=c["T"], device=dev(),
                                         gaussian_loss_type=mtk.GaussianLossType(gaussian_loss),
                                         gaussian_parametrization=mtk.GaussianParametrization.X0, parametrization=mtk.Parametrization.DIRECT).to(dev())
    d.math = math
    tr = mtk.ClavaDDPMTrainer(d, iter(()), 2e-3, 1e-4, 12, device=dev())
    eng = tr._engine
    eng.train_set_gaussian_loss(gaussian_loss)
    eng.set_math(math)
    lm, lg = torch.empty(B, device=dev()), torch.empty(B, device=dev())
    eng.train_step(x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()), 1.0 / B, lm, lg)
    got = export(tr, "grads")
    assert abs(float(lm.mean()) - float(lm_ref)) <= 2e-4 * abs(float(lm_ref)) and abs(float(lg.mean()) - float(lg_ref)) <= 2e-4 * abs(float(lg_ref))
    for k in want:
        num, den = float((got[k].double() - want[k].double()).norm()), float(want[k].double().norm())
        assert num <= 4e-4 * den + 3 * floor[k] + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e} (floor {floor[k] / max(den, 1e-30):.3e})"
    # the flags do something: the default parametrisations give other gradients on the same batch
    other, _, _ = orc.loss_and_grads(s, c["K"], c["Nn"], p0, x0, t, eps, u, gaussian_loss=gaussian_loss)
    assert float((other["mlp.head.weight"] - want["mlp.head.weight"]).norm()) > 1e-2 * float(want["mlp.head.weight"].norm())


@pytest.mark.parametrize("B,layers", [(4096, [512] * 4), (2048, [1024] * 4), (1000, [256, 128, 256])])
def test_bf16_training_gradients(B, layers):
    """mtk_train_set_precision(MTK_PREC_BF16): every GEMM of the iteration multiplies bf16-rounded operands with fp32 accumulation
    (fp32 master weights, gradients and optimizer).  Stated tolerance, against the oracle's fp32 autograd: losses within 2e-3,
    every gradient tensor within 3e-2 relative L2 (measured 8e-3 at 4x512, 1.8e-2 at 4x1024: the roundings chain through the layers), and still bit
    identical run to run."""
    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=layers)
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    x0, t, eps, u = big_batch(c, B)
    s = orc.make_schedule(c["T"], "cosine")
    want, lm_ref, lg_ref = orc.loss_and_grads(s, c["K"], c["Nn"], p0, x0, t, eps, u)
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    eng.train_set_precision("bf16")
    lm, lg = torch.empty(B, device=dev()), torch.empty(B, device=dev())
    args = (x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()), 1.0 / B, lm, lg)
    eng.train_step(*args)
    got = export(tr, "grads")
    first = tr.state[1].clone()
    assert abs(float(lg.mean()) - float(lg_ref)) <= 2e-3 * abs(float(lg_ref)) and abs(float(lm.mean()) - float(lm_ref)) <= 2e-3 * abs(float(lm_ref))
    worst = 0.0
    for k in want:
        num, den = float((got[k].double() - want[k].double()).norm()), float(want[k].double().norm())
        worst = max(worst, num / max(den, 1e-30))
        assert num <= 3e-2 * den + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e}"
    print(f"bf16 training gradients, {layers} x {B}: worst relative L2 error {worst:.3e}")
    eng.train_step(*args)
    assert torch.equal(tr.state[1], first)
    eng.train_set_precision("fp32")
    eng.train_step(*args)
    got32 = export(tr, "grads")
    for k in want:
        assert float((got32[k].double() - want[k].double()).norm()) <= 2e-4 * float(want[k].double().norm()) + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_iteration_needs_no_memset_of_its_buffers(precision):
    """The deterministic iteration stores every element of the gradient image (so it is not zeroed per step) and keeps the rows
    past the batch of its input / output-gradient buffers zero by tracking the rows written (zeroed only when a batch is smaller
    than the one before): a gradient image filled with NaN beforehand comes out finite and equal to a clean run, and a small
    batch after a big one equals the small batch alone."""
    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=[256, 256])
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    eng.train_set_precision(precision)
    small, big = big_batch(c, 300), big_batch(c, 1500)
    def run(batch):
        B = batch[0].shape[0]
        lm, lg = torch.empty(B, device=dev()), torch.empty(B, device=dev())
        eng.train_step(*(v.to(dev()) for v in batch), 1.0 / B, lm, lg)
        return tr.state[1].clone(), lm.clone(), lg.clone()
    clean = run(small)
    tr.state[1].fill_(float("nan"))
    again = run(small)
    assert bool(torch.isfinite(again[0]).all())
    assert all(torch.equal(a, b) for a, b in zip(clean, again))
    run(big)
    eng.train_forward(*(v.to(dev()) for v in big), torch.empty(1500, device=dev()), torch.empty(1500, device=dev()))
    after_big = run(small)
    assert all(torch.equal(a, b) for a, b in zip(clean, after_big))


@pytest.mark.gpu
@pytest.mark.parametrize("precision,p", [("fp32", 0.1), ("fp32", 0.5), ("bf16", 0.25)])
def test_dropout_in_the_training_iteration(precision, p):
    """MODEL:162-180 (MLP.Block = Linear -> ReLU -> Dropout, training mode; VERDICT r2 missing 2).  torch's Bernoulli stream cannot
    be matched off-device, so the mask is the library's own Philox draw, and what is checked is everything that follows from it:
    given the masks (mtk_train_dropout_mask), losses and every gradient tensor equal the oracle's autograd through
    relu(linear) * mask / (1 - p) at the tolerance of the precision; the keep rate is 1 - p; the masks move with the seed and do
    not depend on how the rows are split over calls (row0); p = 0 restores the plain iteration bit for bit."""
    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=[256, 512, 256])
    B = 1500
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    # kept units are scaled by 1 / (1 - p): at random init the head then puts out logits in the hundreds and the loss is so badly
    # conditioned that fp32 rounding alone moves the gradients by 5e-3 (measured, p = 0.5); a head scaled down keeps the comparison
    # about the masks (fp32 cases; the bf16 case keeps the plain head, its 3e-2 bar is relative to full-size gradients)
    if precision == "fp32":
        p0["mlp.head.weight"], p0["mlp.head.bias"] = p0["mlp.head.weight"] * 0.05, p0["mlp.head.bias"] * 0.05
    x0, t, eps, u = big_batch(c, B)
    s = orc.make_schedule(c["T"], "cosine")
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    eng.train_set_precision(precision)
    lm, lg = torch.empty(B, device=dev()), torch.empty(B, device=dev())
    args = (x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()), 1.0 / B, lm, lg)
    eng.train_step(*args)
    plain = tr.state[1].clone()
    eng.train_set_dropout(p, seed=1234)
    eng.train_step(*args, row0=64)
    got = export(tr, "grads")
    masks = [eng.train_dropout_mask(i, B, row0=64).cpu() for i in range(3)]
    for m in masks:
        assert abs(float(m.float().mean()) - (1 - p)) < 0.01
    # the same rows as two calls' worth of row offsets; another seed gives another mask
    assert torch.equal(eng.train_dropout_mask(1, 500, row0=64 + 1000).cpu(), masks[1][1000:])
    eng.train_set_dropout(p, seed=1235)
    assert not torch.equal(eng.train_dropout_mask(1, B, row0=64).cpu(), masks[1])
    want, lm_ref, lg_ref = orc.loss_and_grads(s, c["K"], c["Nn"], p0, x0, t, eps, u, drop_masks=masks, drop_p=p)
    tol_l, tol_g = (2e-4, 4e-4) if precision == "fp32" else (2e-3, 3e-2)  # measured: 2.1e-4 worst gradient tensor at p = 0.5 in fp32
    assert abs(float(lg.mean()) - float(lg_ref)) <= tol_l * abs(float(lg_ref)) + 1e-7 and abs(float(lm.mean()) - float(lm_ref)) <= tol_l * abs(float(lm_ref)) + 1e-7
    for k in want:
        num, den = float((got[k].double() - want[k].double()).norm()), float(want[k].double().norm())
        assert num <= tol_g * den + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e}"
    eng.train_set_dropout(0.0)
    eng.train_step(*args)
    assert torch.equal(tr.state[1], plain)


@pytest.mark.gpu
def test_trainer_applies_dropout_in_training_mode_only():
    """A denoiser built with dropout > 0 (MODEL:576-582): the trainer's iteration draws a fresh mask per step while the module is
    in training mode and none in eval mode; a plain forward() in training mode still refuses (the inference kernels carry no mask)."""
    import midst_toolkit_amd as mtk

    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=[256, 256])
    x0, t, eps, u = big_batch(c, 512)

    def run(train_mode):
        torch.manual_seed(0)
        net = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(c["d_layers"]), 0.3))
        net.load_state_dict(orc.init_mlp_params(c["D"], c["d_layers"], seed=0), strict=True)
        net.train(train_mode)
        d = mtk.GaussianMultinomialDiffusion(np.asarray(c["K"]), c["Nn"], net, num_timesteps=c["T"], device=dev()).to(dev())
        net.train(train_mode)
        tr = mtk.ClavaDDPMTrainer(d, iter(()), 2e-3, 1e-4, 10, device=dev())
        losses = [tuple(float(v) for v in tr._train_step(x0, injected=(t, eps, u))) for _ in range(3)]
        return net, losses

    net_eval, eval_losses = run(False)
    net_train, train_losses = run(True)
    assert all(np.isfinite(v) for pair in train_losses for v in pair)
    assert train_losses[0] != eval_losses[0]              # the first step already sees a mask
    _, again = run(True)
    assert again == train_losses                          # and the masks follow the trainer's seed and step count
    with torch.no_grad():
        net_eval(x0[:8, :].to(dev()), t[:8].to(dev()))
        with pytest.raises(NotImplementedError):
            net_train(x0[:8, :].to(dev()), t[:8].to(dev()))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_label_conditioned_denoiser_trains(precision):
    """MODEL:583-585, 610-612 under training (VERDICT r2 missing 1): emb += silu(label_embedding(y)) in the HIP forward, and
    label_embedding.weight's gradient - d act[0] scattered over the labels, times silu' - beside the other thirteen tensors', all
    against the oracle's autograd with the same labels (fp32: <= 4e-4 relative L2 - the UNLABELLED net on this batch measures
    1.2e-4 .. 2.7e-4 per tensor, the loss formulation's conditioning floor at extreme timesteps, and the label path adds nothing
    to it: a zero table reproduces those figures digit for digit - losses <= 1e-4; bf16 mode: 3e-2, losses 6e-3).  The optimizer step then moves the table, sync_modules writes it back into the module, and the module's own
    (label, timestep) table-row context evaluates the trained weights."""
    import midst_toolkit_amd as mtk

    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=[256, 256])
    n_cls, B = 5, 4096
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    g = torch.Generator().manual_seed(11)
    p0["label_embedding.weight"] = 0.3 * torch.randn((n_cls, 128), generator=g)
    x0, t, eps, u = big_batch(c, B)
    y = torch.randint(0, n_cls, (B, 1), generator=g)
    s = orc.make_schedule(c["T"], "cosine")
    want, lm_ref, lg_ref = orc.loss_and_grads(s, c["K"], c["Nn"], p0, x0, t, eps, u, y=y)
    assert float(want["label_embedding.weight"].norm()) > 0
    net = mtk.MLPDiffusion(c["D"], n_cls, mtk.IsTargetConditioned.EMBEDDING, mtk.DiffusionParameters(list(c["d_layers"]), 0.0))
    net.load_state_dict(p0, strict=True)
    d = mtk.GaussianMultinomialDiffusion(np.asarray(c["K"]), c["Nn"], net, num_timesteps=c["T"], device=dev()).to(dev())
    tr = mtk.ClavaDDPMTrainer(d, iter(()), 2e-3, 1e-4, 10, device=dev())
    tr.set_train_precision(precision)
    lm, lg = tr._train_step(x0, {"y": y}, injected=(t, eps, u))
    tol_l, tol_g = (1e-4, 4e-4) if precision == "fp32" else (6e-3, 3e-2)  # bf16 on this small net: Gaussian loss off by 4.1e-3 (measured)
    assert abs(float(lm) - float(lm_ref)) <= tol_l * abs(float(lm_ref)) and abs(float(lg) - float(lg_ref)) <= tol_l * abs(float(lg_ref))
    bufs = [(torch.empty_like(W), torch.empty_like(b)) for W, b in tr._tensors(net)]
    tr._engine.train_export("grads", bufs)
    names = orc.mlp_param_names(len(c["d_layers"]))
    names.insert(3, "label_embedding")
    errs = {}
    for name, (W, b) in zip(names, bufs):
        pairs = [(name + ".weight", W)] + ([(name + ".bias", b)] if name != "label_embedding" else [])
        for k, got in pairs:
            errs[k] = float((got.cpu().double() - want[k].double()).norm()) / max(float(want[k].double().norm()), 1e-30)
    print("label-conditioned gradients, relative L2 errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= tol_g, errs
    assert float(bufs[3][1].abs().max()) == 0.0  # the unused bias slot of the embedding's layer stays zero
    # the step moved the table; the module sees it after sync_modules and serves it through its own inference context
    before = net.label_embedding.weight.detach().clone()
    tr.sync_modules()
    assert not torch.equal(net.label_embedding.weight.detach(), before)
    trained = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    net.eval()
    with torch.no_grad():
        xq = torch.randn((64, c["D"]))
        tq = torch.randint(0, c["T"], (64,))
        got = net(xq.to(dev()), tq.to(dev()), y[:64].to(dev()))
    close(got, orc.mlp_forward(trained, xq, tq, y=y[:64]), rtol=1e-4, atol=1e-5)
    if precision != "fp32":
        return
    # the reference's own loop (zero_grad -> mixed_loss(x, {"y": y}) -> backward -> AdamW.step, TRAINER:70-94) through the autograd seam
    net2 = mtk.MLPDiffusion(c["D"], n_cls, mtk.IsTargetConditioned.EMBEDDING, mtk.DiffusionParameters(list(c["d_layers"]), 0.0))
    net2.load_state_dict(p0, strict=True)
    d2 = mtk.GaussianMultinomialDiffusion(np.asarray(c["K"]), c["Nn"], net2, num_timesteps=c["T"], device=dev()).to(dev())
    d2._draw_loss_noise = lambda f: (t.to(dev()), eps.to(dev()), u.to(dev()))
    opt = torch.optim.AdamW(d2.parameters(), lr=1e-3)
    opt.zero_grad()
    l_m, l_g = d2.mixed_loss(x0.to(dev()), {"y": y.to(dev())})
    (l_m + l_g).backward()
    assert abs(float(l_m.detach()) - float(lm_ref)) <= 1e-4 * abs(float(lm_ref))
    got = {k: v.grad.detach().cpu() for k, v in net2.named_parameters()}
    assert set(got) == set(want)
    for k in want:
        num, den = float((got[k].double() - want[k].double()).norm()), float(want[k].double().norm())
        assert num <= 4e-4 * den + 1e-9, f"autograd {k}: relative L2 error {num / max(den, 1e-30):.3e}"
    w_before = net2.label_embedding.weight.detach().clone()
    opt.step()
    assert not torch.equal(net2.label_embedding.weight.detach(), w_before)
    l_m2, l_g2 = d2.mixed_loss(x0.to(dev()), {"y": y.to(dev())})  # picks the stepped parameters up (image 0 re-imported)
    assert float(l_m2.detach() + l_g2.detach()) != float(l_m.detach() + l_g.detach())
    with pytest.raises(ValueError):
        d2.mixed_loss(x0.to(dev()), {})


def test_bf16_training_berka_run():
    """The 200-step Berka run of test_berka_run_200_steps with the bf16 training GEMMs.  Stated tolerance on the loss curve against
    the fp32 reference curve: the first 10 losses within 5e-3, mean deviation <= 4e-2 and max <= 2e-1 over the 200 steps (measured
    2.2e-2 / 1.0e-1; the run's own chaos floor under a 1e-7 perturbation is 6.4e-3 / 3.3e-2), the last 20 losses' mean within 3 %
    (measured 1.1 %)."""
    from tests.test_oracle_train import load_berka

    c = load_berka()
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    tr, d = build_trainer(c, p0, float(c["lr"]), float(c["wd"]), c["steps"])
    tr._engine.train_set_precision("bf16")
    hist = []
    for step, (x0, t, eps, u) in enumerate(c["batches"]):
        lm, lg = tr._train_step(x0, None, injected=(t.to(dev()), eps.to(dev()), None))
        tr._anneal_learning_rate(step)
        hist.append(float(lg))
    hist, ref = np.asarray(hist), c["losses"][:, 1]
    rel = np.abs(hist - ref) / np.abs(ref)
    print(f"berka 200-step curve, bf16 training: max rel deviation {rel.max():.3e}, mean {rel.mean():.3e}, first 10 {rel[:10].max():.3e}; "
          f"last-20 mean {hist[-20:].mean():.4f} vs {ref[-20:].mean():.4f}")
    assert rel[:10].max() <= 5e-3 and rel.mean() <= 4e-2 and rel.max() <= 2e-1, (rel[:10].max(), rel.mean(), rel.max())
    assert abs(hist[-20:].mean() - ref[-20:].mean()) <= 3e-2 * ref[-20:].mean()


@pytest.mark.parametrize("B,layers", [(4096, [512] * 4), (1000, [256, 128, 256])])
def test_gradients_are_bit_identical_run_to_run(B, layers):
    """The reference pipeline runs with torch.use_deterministic_algorithms (common/random.py:11-55).  Default mode here:
    split-batch reductions added in a fixed order, the time-embedding scatter in fixed point - the same batch gives the
    same gradient image bit for bit, call after call.  (The fp32-atomics mode stays available and within the parity bar.)"""
    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=layers)
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    x0, t, eps, u = [v.to(dev()) for v in big_batch(c, B)]
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    runs = []
    for rep in range(3):
        eng.train_step(x0, t, eps, u, 1.0 / B, None, None)
        runs.append(tr.state[1].clone())
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    eng.train_set_deterministic(False)
    eng.train_step(x0, t, eps, u, 1.0 / B, None, None)
    atomics = tr.state[1].clone()
    rel = float((atomics - runs[0]).norm() / runs[0].norm())
    assert rel < 1e-5, rel  # same sums, another order


@pytest.mark.parametrize("K,layers", [([10] * 10, [256, 128, 256]), ([3, 12, 16, 2], [128, 128]), ([4, 30, 7], [128, 128])])
def test_training_sized_kernels_equal_the_general_ones(monkeypatch, K, layers):
    """Round 3: at the trainer's batch sizes q_sample runs one lane per (record, item) (k_q_sample_items) and the loss-gradient
    kernel keeps a block's logits in registers (k_loss_grad<KC>).  Both are re-arrangements, not other arithmetic: under the A/B
    switches MTK_QS_ROWS / MTK_LG_STREAM (the lane-per-record and from-memory forms, which the parity tests above pinned to the
    oracle) every output is the same bit for bit - noised rows, drawn codes, per-record losses, the whole gradient image; rows
    with a timestep outside [0, T) included.  Blocks of 12 and 16 levels sit on the register variants' edges, 30 is streamed."""
    Nn = 5
    c = dict(K=K, Nn=Nn, T=1000, D=Nn + sum(K), d_layers=layers)
    B = 1337
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    x0, t, eps, _ = [v.to(dev()) for v in big_batch(c, B)]
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    eng.set_math("fast")
    tb = t.clone()
    tb[7], tb[900] = -1, c["T"]
    ld = 1408

    def noised(e):
        x_in = torch.zeros((c["D"], ld), device=dev())
        idx = torch.zeros((len(K), B), dtype=torch.int32, device=dev())
        eng.q_sample(x0, tb, e, None, x_in, (1, ld), idx, seed=77, row0=4096)
        return x_in.view(torch.int32).clone(), idx

    def iteration(e):
        rows = torch.empty((2, B), device=dev())
        eng.train_step(x0, t, e, None, 1.0 / B, rows[0], rows[1], seed=77, row0=4096)
        return rows.clone(), tr.state[1].clone()

    new = [noised(None), noised(eps), iteration(None), iteration(eps)]
    monkeypatch.setenv("MTK_QS_ROWS", "1")
    monkeypatch.setenv("MTK_LG_STREAM", "1")
    old = [noised(None), noised(eps), iteration(None), iteration(eps)]
    for a, b in zip(new, old):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert bool(torch.isfinite(new[2][0]).all()) and float(new[2][1].abs().sum()) > 0


def test_library_draws_the_timesteps_and_takes_the_batch_means():
    """Round 3: with rng = "philox" the trainer hands the iteration no timesteps (mtk_train_step, t = NULL: GMD:1011's uniform
    sample_time drawn in the noising kernel) and reads the two batch means from the library (mtk_train_set_loss_means) - no torch
    launch is left in the iteration.  Checked: the drawn t are in range, uniform (chi-square over 20 bins), differ between seeds
    and between ranks' row offsets; an iteration given those t explicitly reproduces losses and gradients bit for bit in both
    math modes; the means equal the fp64 means of the per-record losses; the trainer's returned pair is that."""
    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=[256, 128, 256])
    B = 6000
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    x0 = big_batch(c, B)[0].to(dev())
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    means = torch.full((2,), -1.0, device=dev())
    eng.train_set_loss_means(means)
    for math in ("fast", "reference"):
        eng.set_math(math)
        rows = torch.empty((2, B), device=dev())
        eng.train_step(x0, None, None, None, 1.0 / B, rows[0], rows[1], seed=11, row0=B)
        got_means, grads, t = means.clone(), tr.state[1].clone(), eng.train_timesteps(B)
        assert int(t.min()) >= 0 and int(t.max()) < c["T"]
        counts = torch.bincount(t // 50, minlength=20).double().cpu()
        assert float(((counts - B / 20) ** 2 / (B / 20)).sum()) < 50.0  # 19 degrees of freedom: 1e-4 tail at 49
        ref = rows.double().mean(dim=1)
        assert float((got_means.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        rows2 = torch.empty((2, B), device=dev())
        eng.train_step(x0, t, None, None, 1.0 / B, rows2[0], rows2[1], seed=11, row0=B)
        assert torch.equal(rows, rows2) and torch.equal(grads, tr.state[1]) and torch.equal(means, got_means)
        if math == "fast":
            t_fast = t
    assert torch.equal(t, t_fast)  # the draw does not depend on the math mode
    eng.train_step(x0, None, None, None, 1.0 / B, rows[0], rows[1], seed=12, row0=B)
    assert not torch.equal(eng.train_timesteps(B), t)
    eng.train_step(x0, None, None, None, 1.0 / B, rows[0], rows[1], seed=11, row0=0)
    assert not torch.equal(eng.train_timesteps(B), t)
    eng.train_set_loss_means(None)
    with pytest.raises(RuntimeError):
        eng.train_step(x0, t, None, None, 1.0 / B, None, None, seed=11)
        eng.train_timesteps(B)  # the last forward was given its timesteps
    # the trainer: nothing but the library's launches, the returned pair is the means of that iteration
    d.rng = "philox"
    lm, lg = tr._train_step(x0)
    assert lm.shape == () and bool(torch.isfinite(lm)) and bool(torch.isfinite(lg)) and float(lg) > 0 and float(lm) > 0
    keep = (float(lm), float(lg))
    tr._train_step(x0)
    assert (float(lm), float(lg)) == keep  # a fresh pair per iteration: what was returned earlier is not overwritten


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_deferred_reduction_gives_the_same_training_run(precision):
    """Round 3: on one device ClavaDDPMTrainer lets the optimizer pass add the split-batch partials of the gradients
    (mtk_train_set_deferred_reduce: k_adamw_ema_segs instead of k_reduce_splits + k_adamw_ema).  Same sums in the same order, same
    per-element update: after several iterations all five images (weights, gradients, both Adam moments, EMA) are the same bits
    as with the separate reduction; an export of the gradients between the two calls completes the image itself; a larger batch
    arriving while partials are pending (the slabs are reallocated) does not lose them."""
    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=[256, 512, 256])
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    B = 1500
    x0, t, eps, u = [v.to(dev()) for v in big_batch(c, B)]
    x1 = big_batch(c, 4000, seed=6)[0].to(dev())
    states = []
    for defer in (False, True):
        tr, d = build_trainer(c, p0)
        eng = tr._engine
        eng.train_set_precision(precision)
        eng.set_math("fast")
        for step in range(1, 5):
            eng.train_set_deferred_reduce(defer)
            eng.train_step(x0, t, None, None, 1.0 / B, None, None, seed=100 + step)
            if step == 2:  # something else reads the gradient image first
                g_mid = export(tr, "grads")
            if step == 3:  # the slabs move while partials are pending
                eng.train_forward(x1, torch.zeros(4000, dtype=torch.int64, device=dev()), None, None, seed=1)
            eng.train_apply(2e-3, 1e-4, step, (0.9, 0.999), 1e-8, 0.999)
            eng.train_set_deferred_reduce(False)
        states.append((tr.state.clone(), g_mid))
    assert torch.equal(states[0][0], states[1][0])
    for k in states[0][1]:
        assert torch.equal(states[0][1][k], states[1][1][k]), k
    assert float(states[0][0][1].abs().sum()) > 0


def test_adamw_ema_kernel():
    c = load_train("B_ragged")
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    n = len(c["d_layers"])
    g = torch.Generator().manual_seed(9)
    names = orc.mlp_param_names(n)

    def rand_like_params(scale, positive=False):
        out = {}
        for k, v in p0.items():
            r = torch.randn(v.shape, generator=g) * scale
            out[k] = r.abs() if positive else r
        return out

    imgs = {"weights": p0, "grads": rand_like_params(0.1), "exp_avg": rand_like_params(0.05), "exp_avg_sq": rand_like_params(0.01, True),
            "ema": rand_like_params(1.0)}
    for image, vals in imgs.items():
        eng.train_import(image, [(vals[k + ".weight"].to(dev()), vals[k + ".bias"].to(dev())) for k in names])
    eng.train_apply(1.7e-3, 1e-2, 7)
    for k in p0:
        p, m, v = imgs["weights"][k].clone(), imgs["exp_avg"][k].clone(), imgs["exp_avg_sq"][k].clone()
        orc.adamw_update(p, imgs["grads"][k], m, v, 7, 1.7e-3, 1e-2)
        imgs["weights"][k], imgs["exp_avg"][k], imgs["exp_avg_sq"][k] = p, m, v
    orc.ema_update(imgs["ema"], imgs["weights"])
    for image in ("weights", "exp_avg", "exp_avg_sq", "ema"):
        got = export(tr, image)
        for k in p0:
            torch.testing.assert_close(got[k], imgs[image][k], rtol=2e-6, atol=1e-9)
    # the forward pass now runs on the updated weights (and a rebuilt time-embedding table)
    x = torch.randn((50, c["D"]), generator=g)
    t = torch.randint(0, c["T"], (50,), generator=g)
    want = orc.mlp_forward(imgs["weights"], x, t)
    got = eng.mlp_forward(x.to(dev()), t.to(dev()))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["B_ragged", "A_numeric"])
def test_training_run_replays_reference(tag):
    c = load_train(tag)
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    lr, wd = float(c["lr"]), float(c["wd"])
    tr, d = build_trainer(c, p0, lr, wd, c["steps"])
    has_cat = c["K"][0] != 0
    hist = []
    for step, (x0, t, eps, u) in enumerate(c["batches"]):
        lm, lg = tr._train_step(x0, None, injected=(t.to(dev()), eps.to(dev()), u.to(dev()) if has_cat else None))
        tr._anneal_learning_rate(step)
        hist.append([float(lm), float(lg)])
    hist = np.asarray(hist)
    ref = c["losses"]
    assert np.all(np.abs(hist - ref) <= 1e-2 * np.abs(ref) + 1e-6), (hist, ref)
    tr.sync_modules()
    n = len(c["d_layers"])
    final = flatten({k: v.detach().cpu() for k, v in d._denoise_fn.state_dict().items()}, n).numpy()
    ema = flatten({k: v.detach().cpu() for k, v in tr.ema_model.state_dict().items()}, n).numpy()
    moved = np.abs(c["final"] - flatten(p0, n).numpy())
    err = np.abs(final - c["final"])
    # most weights land within 5% of their total movement; none is off by more than a few lr quanta
    assert float(np.mean(err <= 0.05 * moved + 1e-6)) > 0.9, float(np.mean(err <= 0.05 * moved + 1e-6))
    assert float(err.max()) <= 4 * lr, float(err.max())
    assert float(np.abs(ema - c["ema"]).max()) <= 4 * lr * 1e-3 * c["steps"] + 1e-6


@pytest.mark.parametrize("tag", ["B_ragged", "A_numeric"])
def test_training_run_teacher_forced(tag):
    c = load_train(tag)
    s = orc.make_schedule(c["T"], "cosine")
    lr0, wd, n = float(c["lr"]), float(c["wd"]), len(c["d_layers"])
    names = orc.mlp_param_names(n)
    p = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    tr, d = build_trainer(c, p, lr0, wd, c["steps"])
    eng = tr._engine
    has_cat = c["K"][0] != 0
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    lr = lr0
    pack = lambda dct: [(dct[k + ".weight"].to(dev()), dct[k + ".bias"].to(dev())) for k in names]
    gen = torch.Generator().manual_seed(1)
    for step, (x0, t, eps, u) in enumerate(c["batches"]):
        for image, vals in (("weights", p), ("exp_avg", m), ("exp_avg_sq", v2)):
            eng.train_import(image, pack(vals))
        B = x0.shape[0]
        lm = torch.empty(B, device=dev()) if has_cat else None
        lg = torch.empty(B, device=dev())
        eng.train_step(x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()) if has_cat else None, 1.0 / B, lm, lg)
        got = export(tr, "grads")
        want, lm_ref, lg_ref = orc.loss_and_grads(s, c["K"], c["Nn"], p, x0, t, eps, u)
        # (the oracle equals the reference run bit for bit only on the CPU that generated the fixture:
        #  tests/test_oracle_train.py; on another host its fp32 libm/BLAS roundings differ, so no such assert here)
        assert abs(float(lg.mean()) - float(lg_ref)) <= 1e-5 * abs(float(lg_ref)) + 1e-6
        if has_cat:
            xt, idx = orc.q_sample_fused(s, c["K"], c["Nn"], x0, t, eps, u)
            out = orc.mlp_forward(p, orc.model_input(xt, idx, c["K"]), t)
            floor = float(orc.multinomial_loss_floor(s, c["K"], c["Nn"], out, x0, idx, t).mean())
            assert abs(float(lm.mean()) - float(lm_ref)) <= 1e-4 * abs(float(lm_ref)) + 3 * floor
        sens = {k: 0.0 for k in want}
        for trial in range(4):
            pp = {k: v * (1 + 1e-6 * torch.randn(v.shape, generator=gen)) for k, v in p.items()}
            alt, _, _ = orc.loss_and_grads(s, c["K"], c["Nn"], pp, x0, t, eps, u)
            for k in want:
                sens[k] = max(sens[k], float((alt[k] - want[k]).norm() / want[k].norm().clamp_min(1e-30)))
        for k in want:
            num, den = float((got[k].double() - want[k].double()).norm()), float(want[k].double().norm())
            # 24-row batches: a ReLU gate whose pre-activation is within rounding of zero flips a whole unit's
            # contribution.  The bar is 2e-4 or 3x the reference's OWN movement under a 1e-6 relative weight perturbation (the size of fp32 GEMM summation-order differences)
            assert num <= max(2e-4, 3 * sens[k]) * den + 1e-9, f"step {step} {k}: relative L2 error {num / max(den, 1e-30):.3e} (sensitivity {sens[k]:.1e})"
        # apply with the reference's gradient on both sides: the optimizer kernel stays on the trajectory
        eng.train_import("grads", pack(want))
        eng.train_apply(lr, wd, step + 1)
        for k in p:
            orc.adamw_update(p[k], want[k], m[k], v2[k], step + 1, lr, wd)
        lr = orc.annealed_lr(lr0, step, c["steps"])
        after = export(tr, "weights")
        for k in p:
            torch.testing.assert_close(after[k], p[k], rtol=2e-6, atol=1e-8)


def test_berka_run_200_steps():
    """BASELINE configs[0]: the reference trainer's 200-iteration run on the Berka 'trans' subset
    with the integration test's denoiser [512, 1024, 1024, 1024, 1024, 512], T = 1000, batch 24, lr 6e-4
    (tests/golden/train_berka.npz: the reference's draws, loss curve and a strided sample of its trained / EMA weights).
    The HIP trainer runs free on the same batches and draws.  The run is chaotic at this size (4.7 M weights, batch 24): the
    reference's own curve moves by up to 3.3e-2 relative (mean 6.4e-3) and its trained weights by up to 1.6e-2 (mean 1e-3) when
    its initial weights are perturbed by 1e-7 relative (measured with the oracle, which reproduces the fixture bit for bit on
    the CPU).  Bars: the first 10 losses within 1e-3, the whole curve within twice that floor, the mean of the last 20 losses
    within 2 %, the sampled weights within twice the floor."""
    from tests.test_oracle_train import load_berka

    c = load_berka()
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    lr, wd = float(c["lr"]), float(c["wd"])
    tr, d = build_trainer(c, p0, lr, wd, c["steps"])
    hist = []
    for step, (x0, t, eps, u) in enumerate(c["batches"]):
        lm, lg = tr._train_step(x0, None, injected=(t.to(dev()), eps.to(dev()), None))
        tr._anneal_learning_rate(step)
        hist.append(float(lg))
    hist, ref = np.asarray(hist), c["losses"][:, 1]
    rel = np.abs(hist - ref) / np.abs(ref)
    print(f"berka 200-step curve: max rel deviation {rel.max():.3e}, mean {rel.mean():.3e}, first 10 {rel[:10].max():.3e}; "
          f"loss {ref[0]:.4f} -> {ref[-20:].mean():.4f} (ours {hist[-20:].mean():.4f})")
    assert rel[:10].max() <= 1e-3, rel[:10]
    assert rel.max() <= 6.6e-2 and rel.mean() <= 1.3e-2, (rel.max(), rel.mean())
    assert abs(hist[-20:].mean() - ref[-20:].mean()) <= 2e-2 * ref[-20:].mean()
    tr.sync_modules()
    n = len(c["d_layers"])
    final = flatten({k: v.detach().cpu() for k, v in d._denoise_fn.state_dict().items()}, n).numpy()[:: c["stride"]]
    ema = flatten({k: v.detach().cpu() for k, v in tr.ema_model.state_dict().items()}, n).numpy()[:: c["stride"]]
    err = np.abs(final - c["final_sample"])
    print(f"berka weights: max err {err.max():.3e}, mean {err.mean():.3e}; ema max err {np.abs(ema - c['ema_sample']).max():.3e}")
    assert float(err.max()) <= 3.2e-2 and float(err.mean()) <= 2e-3, (float(err.max()), float(err.mean()))
    assert float(np.abs(ema - c["ema_sample"]).max()) <= 3.2e-2 * 0.2


def test_bucket_events_leave_the_gradients_unchanged():
    """mtk_train_set_bucket_events: with one event per gradient bucket the backward adds each bucket's split-batch partials as it
    goes and records the bucket's event; the gradient image is bit-identical to the unbucketed call, the layout tiles the image in
    completion order, and a side stream that waits on each event and touches only that bucket sees final values."""
    c = dict(K=[3, 7, 10, 2, 12], Nn=6, T=1000, D=40, d_layers=[256, 128, 128])
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    B = 3000
    x0, t, eps, u = [v.to(dev()) for v in big_batch(c, B)]
    tr, d = build_trainer(c, p0)
    eng = tr._engine
    eng.train_step(x0, t, eps, u, 1.0 / B, None, None)
    torch.cuda.synchronize()
    plain = tr.state[1].clone()
    layout = eng.train_bucket_layout()
    assert len(layout) == len(c["d_layers"]) + 3
    spans = sorted(layout)
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] == tr.state.shape[1]
    assert layout[0][0] == max(o for o, _ in layout)  # the head sits last in the image and completes first
    events = [torch.cuda.Event() for _ in layout]
    eng.train_set_bucket_events(events)
    side = torch.cuda.Stream()
    copies = []
    eng.train_step(x0, t, eps, u, 1.0 / B, None, None)
    with torch.cuda.stream(side):
        for ev, (off, ln) in zip(events, layout):
            side.wait_event(ev)
            copies.append((off, ln, tr.state[1][off:off + ln].clone()))
    torch.cuda.synchronize()
    assert torch.equal(tr.state[1], plain)
    for off, ln, cp in copies:
        assert torch.equal(cp, plain[off:off + ln]), (off, ln)
    eng.train_set_bucket_events(None)
    eng.train_step(x0, t, eps, u, 1.0 / B, None, None)
    torch.cuda.synchronize()
    assert torch.equal(tr.state[1], plain)


def _two_rank_body(rank, ws, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", RANK=str(rank), WORLD_SIZE=str(ws))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import midst_toolkit_amd.distributed as D

        c = dict(K=[3, 7, 10, 2, 12], Nn=6, T=1000, D=40, d_layers=[128, 128])
        p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
        B = 512
        x0, t, eps, u = [v.to(dev()) for v in big_batch(c, B)]
        tr, d = build_trainer(c, p0)
        assert (tr._rank, tr._world) == (rank, ws)
        eng = tr._engine
        eng.train_step(x0, t, eps, u, 1.0 / B, None, None)           # the whole batch on one device
        full = export(tr, "grads")
        lo, hi = D.shard_range(B, rank, ws)
        eng.train_step(x0[lo:hi], t[lo:hi].contiguous(), eps[lo:hi], u[lo:hi], 1.0 / B, None, None, row0=lo)  # this rank's block
        torch.cuda.synchronize()
        D.allreduce_gradients(tr.state[1])
        got = export(tr, "grads")
        worst = max(float((got[k] - full[k]).norm() / full[k].norm().clamp_min(1e-30)) for k in full)
        # the same exchange bucket by bucket (what the trainer does; over gloo its blocking form): identical sums
        eng.train_step(x0[lo:hi], t[lo:hi].contiguous(), eps[lo:hi], u[lo:hi], 1.0 / B, None, None, row0=lo)
        ex = D.BucketedGradientExchange(eng, tr.state[1])
        ex.launch()
        ex.wait()
        again = export(tr, "grads")
        if not all(bool(torch.equal(again[k], got[k])) for k in got):
            worst = float("inf")
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


def test_two_rank_training_gradient():
    """Two ranks (sharing the one GPU of the test box, gloo all-reduce on device tensors): block gradients scaled by
    1 / global batch and summed equal the single-device gradient of the whole batch."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import os

    os.environ["MASTER_PORT"] = str(port)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_two_rank_body, args=(r, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0, p.exitcode
    res = dict(q.get() for _ in range(2))
    assert res[0] < 2e-5 and res[1] < 2e-5, res


def _two_rank_sampling_body(rank, ws, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", RANK=str(rank), WORLD_SIZE=str(ws))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import midst_toolkit_amd as mtk
        import midst_toolkit_amd.distributed as D

        K, Nn, T, N = np.asarray([3, 7, 10, 2, 12]), 6, 12, 301
        torch.manual_seed(0)  # every process seeded alike, as the reference pipeline does
        net = mtk.MLPDiffusion(Nn + int(K.sum()), 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([64, 64], 0.0))
        d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev()).to(dev())
        d.rng = "philox"
        torch.manual_seed(1234)
        x, y = D.sample_all_sharded(d, N, 128, torch.ones(1))
        # the same seed on one device, all rows: what a single-GPU run of the sharded call draws
        torch.manual_seed(1234)
        seed = int(torch.randint(1, 2**62, (1,), dtype=torch.int64).item())
        full, _ = d.sample_all(N, 128, torch.ones(1), row0=0, seed=seed)
        q.put((rank, (tuple(x.shape), bool(torch.equal(x, full)), int(x.isnan().sum()))))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_sampling_equals_the_single_device_rows():
    """sample_all_sharded on two ranks (sharing the test box's one GPU, gloo gather): with rng = "philox" the gathered table is,
    row for row and bit for bit, what one device draws for the same seed - the samples do not depend on the number of GPUs."""
    import os
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_two_rank_sampling_body, args=(r, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0, p.exitcode
    res = dict(q.get() for _ in range(2))
    assert res[0] == ((301, 11), True, 0) and res[1] == ((301, 11), True, 0), res


def _rccl_one_rank_body(q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev())
    res = {}
    try:
        import midst_toolkit_amd as mtk
        import midst_toolkit_amd.distributed as D
        from midst_toolkit_amd import scoring

        # 1. training: the side-stream, per-bucket all-reduce under the backward, over consecutive iterations (the next
        #    iteration's memset of the gradient image and its kernels must order behind the exchange), against the single
        #    blocking all-reduce and against no exchange at all: at one rank every sum is the identity, so all three runs
        #    must agree bit for bit in gradients and in the weights after every AdamW step
        c = dict(K=[3, 7, 10, 2, 12], Nn=6, T=1000, D=40, d_layers=[256, 128, 128])
        p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
        batches = [[v.to(dev()) for v in big_batch(c, 1536, seed=10 + i)] for i in range(6)]

        def run(mode):
            tr, d = build_trainer(c, p0, steps=len(batches))
            if mode == "overlapped":
                tr.use_exchange(True)
                assert tr._exchange.overlapped and tr._exchange.forced and tr._exchange.stream is not None
            snaps = []
            for i, (x0, t, eps, u) in enumerate(batches):
                if mode == "blocking":
                    tr._engine.train_step(x0, t, eps, u, 1.0 / x0.shape[0], None, None)
                    dist.all_reduce(tr.state[1])
                    g = tr.optimizer.param_groups[0]
                    tr.optimizer.steps_done += 1
                    tr._engine.train_apply(g["lr"], g["weight_decay"], tr.optimizer.steps_done, g["betas"], g["eps"], tr.ema_rate)
                else:
                    tr._train_step(x0, injected=(t, eps, u))
                tr._anneal_learning_rate(i)
                snaps.append((tr.state[1].clone(), tr.state[0].clone()))
            torch.cuda.synchronize()
            if tr._exchange is not None:
                tr._exchange.close()
            return snaps

        plain, blocking, over = run("none"), run("blocking"), run("overlapped")
        res["train_blocking"] = all(bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])) for a, b in zip(plain, blocking))
        res["train_overlapped"] = all(bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])) for a, b in zip(plain, over))
        res["train_moved"] = not bool(torch.equal(plain[0][1], plain[-1][1]))

        # 2. gather_rows through all_gather_into_tensor (device tensors, RCCL), along both dims
        x = torch.randn((257, 5), device=dev())
        res["gather"] = bool(torch.equal(D.gather_rows(x, 257, force_collective=True), x)
                             and torch.equal(D.gather_rows(x.t().contiguous(), 257, dim=1, force_collective=True), x.t()))

        # 3. sharded sampling: seed and labels broadcast over RCCL, rows and labels gathered; equal to the plain call
        K, Nn, T, N = np.asarray([3, 7, 10, 2, 12]), 6, 12, 301
        torch.manual_seed(0)
        net = mtk.MLPDiffusion(Nn + int(K.sum()), 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([64, 64], 0.0))
        d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev()).to(dev())
        d.rng = "philox"
        td = torch.tensor([0.2, 0.5, 0.3])
        torch.manual_seed(1234)
        xs, ys = D.sample_all_sharded(d, N, 128, td, force_collective=True)
        torch.manual_seed(1234)
        seed = int(torch.randint(1, 2**62, (1,), dtype=torch.int64).item())
        full, yf = d.sample_all(N, 128, td, row0=0, seed=seed)
        res["sampling"] = (tuple(xs.shape), bool(torch.equal(xs, full)), bool(torch.equal(ys.cpu(), yf.cpu())), int(ys.max()) > 0)

        # 4. sharded scoring: blocks gathered along the record axis
        torch.manual_seed(0)
        netn = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([64, 64], 0.0))
        dn = mtk.GaussianMultinomialDiffusion(np.asarray([0]), Nn, netn, num_timesteps=50, device=dev()).to(dev())
        feats, noise = torch.randn((333, Nn)), torch.randn((3, Nn))
        got = D.score_records_sharded(dn, feats, noise, [1, 7, 30], force_collective=True)
        want = scoring.score_records(dn, feats.to(dev()), noise, [1, 7, 30])
        res["scoring"] = bool(torch.equal(got, want))
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_rccl_paths_on_a_single_rank_group():
    """Every RCCL call of the multi-GPU path, executed once on hardware - as far as a one-GPU box allows: an nccl
    process group of ONE rank.  The overlapped gradient exchange (side stream, per-bucket wait_event + all_reduce, wait_stream
    before the optimizer) is forced on and must leave gradients and weights bit-identical to the blocking all-reduce and to no
    exchange over six consecutive iterations; gather_rows, sample_all_sharded (seed + label broadcast, row gather) and
    score_records_sharded go through their collectives and must equal the single-process results."""
    import os
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_rccl_one_rank_body, args=(q,))
    p.start()
    p.join(240)
    assert p.exitcode == 0, p.exitcode
    res = q.get()
    assert res["train_blocking"] and res["train_overlapped"] and res["train_moved"], res
    assert res["gather"] and res["scoring"], res
    assert res["sampling"] == ((301, 11), True, True, True), res


def _rccl_two_device_body(rank, ws, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", RANK=str(rank), WORLD_SIZE=str(ws))
    torch.cuda.set_device(rank)
    mine = torch.device("cuda", rank)
    globals()["dev"] = lambda: mine  # the helpers of this module build on dev(): this process works on its own GPU
    dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=mine)
    try:
        import midst_toolkit_amd.distributed as D

        c = dict(K=[3, 7, 10, 2, 12], Nn=6, T=1000, D=40, d_layers=[256, 128, 128])
        p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
        B = 2048
        lo, hi = D.shard_range(B, rank, ws)
        batches = [[v[lo:hi].contiguous().to(mine) for v in big_batch(c, B, seed=10 + i)] for i in range(6)]

        def run(mode):
            tr, _ = build_trainer(c, p0, steps=len(batches))
            assert (tr._rank, tr._world) == (rank, ws)
            if mode == "overlapped":
                assert tr._exchange.overlapped and not tr._exchange.forced and tr._exchange.stream is not None
            elif mode == "buckets":
                tr.use_exchange(False)
            else:  # one blocking all-reduce of the whole gradient image (distributed.allreduce_gradients)
                tr._exchange.close()
                tr._exchange = None
            snaps = []
            for i, (x0, t, eps, u) in enumerate(batches):
                if mode == "whole":
                    tr._engine.train_step(x0, t, eps, u, 1.0 / B, None, None)
                    D.allreduce_gradients(tr.state[1])
                    g = tr.optimizer.param_groups[0]
                    tr.optimizer.steps_done += 1
                    tr._engine.train_apply(g["lr"], g["weight_decay"], tr.optimizer.steps_done, g["betas"], g["eps"], tr.ema_rate)
                else:
                    tr._train_step(x0, injected=(t, eps, u))
                tr._anneal_learning_rate(i)
                snaps.append((tr.state[1].clone(), tr.state[0].clone()))
            torch.cuda.synchronize()
            if tr._exchange is not None:
                tr._exchange.close()
            return snaps

        whole, buckets, over = run("whole"), run("buckets"), run("overlapped")
        same = lambda a, b: all(bool(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])) for x, y in zip(a, b))  # noqa: E731
        q.put((rank, dict(buckets=same(whole, buckets), overlapped=same(whole, over), moved=not bool(torch.equal(whole[0][1], whole[-1][1])),
                          weights=float(over[-1][1].double().sum()))))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between devices)")
def test_rccl_gradient_exchange_between_two_devices():
    """the overlapped exchange (side stream, per-bucket wait_event + all_reduce under the backward) between two GPUs
    over RCCL, six consecutive iterations: gradient image and weights after every AdamW step bit-identical to the blocking
    per-bucket form and to one all-reduce of the whole image, and identical on both ranks.  Skipped on a one-GPU box (there
    test_rccl_paths_on_a_single_rank_group runs the same calls at world size 1)."""
    import os
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_rccl_two_device_body, args=(r, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, p.exitcode
    res = dict(q.get() for _ in range(2))
    for r in (0, 1):
        assert res[r]["buckets"] and res[r]["overlapped"] and res[r]["moved"], res
    assert res[0]["weights"] == res[1]["weights"], res


def test_autograd_seam_leaves_a_trainers_image_alone():
    """A differentiable mixed_loss between trainer steps (a validation loss computed without torch.no_grad(), as loops written
    for the reference do) shares the trainer's context.  It must evaluate the LIVE weights of image 0 and never import the
    module's stale nn.Parameters over them (they are only refreshed by sync_modules)."""
    c = dict(K=[3, 7, 10, 2, 12], Nn=6, T=1000, D=40, d_layers=[128, 128])
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    tr, d = build_trainer(c, p0, steps=8)
    batches = [[v.to(dev()) for v in big_batch(c, 512, seed=30 + i)] for i in range(4)]
    for x0, t, eps, u in batches[:2]:
        tr._train_step(x0, injected=(t, eps, u))
    torch.cuda.synchronize()
    w_before, m_before = tr.state[0].clone(), tr.state[2].clone()
    initial = {k: v.detach().clone() for k, v in d._denoise_fn.state_dict().items()}
    assert not torch.equal(w_before, tr.state[4])  # the image has moved away from the initial weights (EMA lags it)
    x0 = batches[2][0]
    torch.manual_seed(5)
    lm, lg = d.mixed_loss(x0, {"y": torch.zeros(len(x0))})          # grad enabled: the autograd seam
    assert lm.requires_grad or lg.requires_grad
    (lm + lg).backward()
    torch.cuda.synchronize()
    assert torch.equal(tr.state[0], w_before) and torch.equal(tr.state[2], m_before)
    # ... and it is the live model that was evaluated, not the module's initial parameters
    torch.manual_seed(5)
    with torch.no_grad():
        lm2, lg2 = d.mixed_loss(x0, {"y": torch.zeros(len(x0))})
    assert abs(float(lg.detach()) - float(lg2)) <= 1e-5 * abs(float(lg2)) + 1e-7
    for k, v in d._denoise_fn.state_dict().items():
        assert torch.equal(v, initial[k])                            # parameters untouched until sync_modules
    x0, t, eps, u = batches[3]
    tr._train_step(x0, injected=(t, eps, u))                         # training carries on from the same state
    torch.cuda.synchronize()
    assert not torch.equal(tr.state[0], w_before) and bool(torch.isfinite(tr.state[0]).all())


def test_smaller_batch_after_the_capacity_batch_has_slab_room():
    """The split-batch slabs are sized when the activations are: splits_for is not monotonic in the batch (8320 rows can
    need fewer slices than 8192), so a smaller batch after the largest one must still fit."""
    c = dict(K=[10] * 10, Nn=20, T=1000, D=120, d_layers=[512, 512])
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    tr, d = build_trainer(c, p0)
    for B in (8320, 8192, 4224, 640):
        x0, t, eps, u = [v.to(dev()) for v in big_batch(c, B)]
        tr._engine.train_step(x0, t, eps, u, 1.0 / B, None, None)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(tr.state[1]).all())


@pytest.mark.parametrize("tag", ["B_ragged", "A_numeric"])
def test_reference_trainer_loop_runs_against_the_drop_in(tag):
    """The reference's own training loop, statement for statement (TRAINER:39-51 set-up, :70-94 _train_step, :58-68 anneal,
    :133 / :138-155 EMA), with torch.optim.AdamW and autograd: zero_grad -> mixed_loss -> (lm + lg).backward() -> step.
    The drop-in's mixed_loss builds the graph (HIP forward + backward behind a torch.autograd.Function).  The reference's
    draws of each iteration are injected where mixed_loss draws them; 12 iterations against tests/golden/train_*.npz with
    the bars of test_training_run_replays_reference, and the first iteration's gradients against the oracle's autograd."""
    from copy import deepcopy

    import midst_toolkit_amd as mtk
    from midst_toolkit_amd.trainer import update_ema

    c = load_train(tag)
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    lr0, wd, steps = float(c["lr"]), float(c["wd"]), c["steps"]
    net = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(c["d_layers"]), 0.0))
    net.load_state_dict(p0, strict=True)
    diffusion = mtk.GaussianMultinomialDiffusion(np.asarray(c["K"]), c["Nn"], net, num_timesteps=c["T"], device=dev()).to(dev())
    has_cat = c["K"][0] != 0
    draws = iter(c["batches"])
    current = {}

    def fixture_draws(features):  # stands where GMD:1025-1038 draws: randint, randn_like, per-column rand_like
        _, t, eps, u = current["batch"]
        return t.to(dev()), eps.to(dev()), (u.to(dev()) if has_cat else None)

    diffusion._draw_loss_noise = fixture_draws
    # ---- TRAINER:39-51
    ema_model = deepcopy(diffusion._denoise_fn)
    for param in ema_model.parameters():
        param.detach_()
    optimizer = torch.optim.AdamW(diffusion.parameters(), lr=lr0, weight_decay=wd)
    hist = []
    for step in range(steps):
        current["batch"] = next(draws)
        x = current["batch"][0].to(dev())
        target = torch.zeros(x.shape[0], dtype=torch.long, device=dev())
        # ---- TRAINER:70-94
        optimizer.zero_grad()
        loss_multi, loss_gauss = diffusion.mixed_loss(x, {"y": target})
        loss = loss_multi + loss_gauss
        loss.backward()
        if step == 0:
            s = orc.make_schedule(c["T"], "cosine")
            want, _, _ = orc.loss_and_grads(s, c["K"], c["Nn"], p0, *current["batch"])
            got = {k: v.grad.detach().cpu() for k, v in diffusion._denoise_fn.named_parameters()}
            assert set(got) == set(want)
            for k in want:
                num, den = float((got[k].double() - want[k].double()).norm()), float(want[k].double().norm())
                assert num <= 2e-4 * den + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e}"
        optimizer.step()
        # ---- TRAINER:58-68, :133
        for group in optimizer.param_groups:
            group["lr"] = lr0 * (1 - step / steps)
        update_ema(ema_model.parameters(), diffusion._denoise_fn.parameters())
        hist.append([float(loss_multi.detach()), float(loss_gauss.detach())])
    hist, ref = np.asarray(hist), c["losses"]
    assert np.all(np.abs(hist - ref) <= 1e-2 * np.abs(ref) + 1e-6), (hist, ref)
    n = len(c["d_layers"])
    final = flatten({k: v.detach().cpu() for k, v in diffusion._denoise_fn.state_dict().items()}, n).numpy()
    ema = flatten({k: v.detach().cpu() for k, v in ema_model.state_dict().items()}, n).numpy()
    moved = np.abs(c["final"] - flatten(p0, n).numpy())
    err = np.abs(final - c["final"])
    assert float(np.mean(err <= 0.05 * moved + 1e-6)) > 0.9, float(np.mean(err <= 0.05 * moved + 1e-6))
    assert float(err.max()) <= 4 * lr0, float(err.max())
    assert float(np.abs(ema - c["ema"]).max()) <= 4 * lr0 * 1e-3 * steps + 1e-6
    # the trained model is usable right away on the no-grad paths (the training image is detached and the weights re-packed)
    with torch.no_grad():
        lm, lg = diffusion.mixed_loss(current["batch"][0].to(dev()), {"y": None})
    assert bool(torch.isfinite(lm).all()) and bool(torch.isfinite(lg).all())
    diffusion.precision = diffusion._denoise_fn.precision = "bf16"
    xs, _ = diffusion.sample_all(16, 16, torch.ones(1))
    assert xs.shape[0] == 16 and bool(torch.isfinite(xs).all())
    # two differentiable forwards, backward through the first: refused (one set of activations per context)
    diffusion.precision = diffusion._denoise_fn.precision = "fp32"
    a1, b1 = diffusion.mixed_loss(x, {"y": target})
    a2, b2 = diffusion.mixed_loss(x, {"y": target})
    with pytest.raises(RuntimeError, match="another differentiable"):
        (a1 + b1).backward()
    (a2 + b2).backward()


def test_from_reference_state_round_trip():
    """Model ingestion from a state dict + constructor arguments: same parameters, same scores."""
    import midst_toolkit_amd as mtk
    from tests.helpers import case_params, load_case, tt

    c = load_case("M_mixed")
    sd = {"_denoise_fn." + k: v for k, v in case_params(c).items()}
    ref = mtk.GaussianMultinomialDiffusion(c["K"], c["Nn"], mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(c["d_layers"]), 0.0)),
                                           num_timesteps=c["T"], scheduler_type=mtk.SchedulerType(c["sched"]), device=torch.device("cpu"))
    sd.update({k: v for k, v in ref.state_dict().items() if not k.startswith("_denoise_fn.")})
    d = mtk.GaussianMultinomialDiffusion.from_reference_state(sd, {"num_classes": c["K"], "num_numerical_features": c["Nn"], "num_timesteps": c["T"],
                                                                   "scheduler_type": c["sched"]}, device=dev())
    assert [int(b.linear.out_features) for b in d._denoise_fn.mlp.blocks] == list(c["d_layers"])
    x, t = tt(c["ml_model_in"]).to(dev()), tt(c["ml_t"]).to(dev())
    with torch.no_grad():
        out = d._denoise_fn(x, t)
    want = tt(c["ml_model_out"])
    assert float((out.cpu() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    with pytest.raises(AssertionError):
        mtk.GaussianMultinomialDiffusion.from_reference_state(sd, {"num_classes": c["K"], "num_numerical_features": c["Nn"] + 1}, device=dev())
