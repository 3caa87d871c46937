"""CPU: the oracle (oracle/diffusion_oracle.py) against vectors produced by the reference itself.

Bit-exact where the reference's op sequence is restated op for op (tables, noising, categorical
draws, per-record losses, fp64 reverse state); this is what pins the oracle.
"""
import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as orc
from tests.helpers import CASES, GOLDEN, case_params, load_case, tt

SCHED_KEYS = [
    "alphas", "log_alpha", "log_1_min_alpha", "log_cumprod_alpha", "log_1_min_cumprod_alpha", "alphas_cumprod",
    "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
    "posterior_log_variance_clipped", "posterior_variance_f64", "model_variance_f64", "model_log_variance_f64",
]


@pytest.mark.parametrize("tag,T,kind", [("cosine_1000", 1000, "cosine"), ("linear_100", 100, "linear")])
def test_schedule_tables_bit_exact(tag, T, kind):
    s = orc.make_schedule(T, kind)
    with np.load(f"{GOLDEN}/sched_{tag}.npz") as z:
        for k in SCHED_KEYS:
            got = getattr(s, k).numpy()
            assert got.dtype == z[k].dtype, k
            assert np.array_equal(got, z[k]), k


def test_schedule_known_answers():
    # SURVEY.md section 8 a1 [probe] values (T=1000 cosine)
    s = orc.make_schedule(1000, "cosine")
    assert np.allclose(s.alphas_cumprod[[0, 500, 999]].numpy(), [0.999958694, 0.492285162, 2.42876697e-09], rtol=1e-7)
    assert abs(float(s.log_1_min_alpha[0]) - (-10.0950298)) < 1e-6
    assert abs(float(s.sqrt_recipm1_alphas_cumprod[999]) - 20291.1699) < 1e-2
    assert float(s.posterior_variance_f64[1]) == 2.178949614569182e-05


@pytest.mark.parametrize("tag", CASES)
def test_mixed_loss_rows_bit_exact(tag):
    c = load_case(tag)
    s = orc.make_schedule(c["T"], c["sched"])
    p = case_params(c)
    x0, t, eps, u = tt(c["x0"]), tt(c["ml_t"]), tt(c["ml_eps"]), tt(c["ml_u"])
    x_num_t, xt_idx = orc.q_sample_fused(s, c["K"], c["Nn"], x0, t, eps, u)
    assert torch.equal(x_num_t, tt(c["ml_x_num_t"]))
    assert torch.equal(xt_idx, tt(c["ml_xt_idx"]))
    x_in = orc.model_input(x_num_t, xt_idx, c["K"])
    assert torch.equal(x_in, tt(c["ml_model_in"]))
    out = orc.mlp_forward(p, x_in, t)
    assert torch.equal(out, tt(c["ml_model_out"]))
    lm, lg = orc.loss_rows(s, c["K"], c["Nn"], out, x0, xt_idx, eps, t)
    assert torch.equal(lm, tt(c["ml_lm_rows"]))
    assert torch.equal(lg, tt(c["ml_lg_rows"]))


@pytest.mark.parametrize("tag", CASES)
def test_mixed_loss_means_match_reference_call(tag):
    """The unmodified reference call (its own drawn t) vs the oracle with the captured draws."""
    c = load_case(tag)
    s = orc.make_schedule(c["T"], c["sched"])
    lm, lg, m_mean, g_mean = orc.mixed_loss(s, c["K"], c["Nn"], case_params(c), tt(c["x0"]), tt(c["ml_ref_call_t"]),
                                            tt(c["ml_eps"]), tt(c["ml_u"]))
    ref = c["ml_ref_call_means"]
    assert np.float32(m_mean) == ref[0] and np.float32(g_mean) == ref[1]


@pytest.mark.parametrize("tag", ["A_numeric", "N_tf"])
def test_tf_score_bit_exact(tag):
    c = load_case(tag)
    s = orc.make_schedule(c["T"], c["sched"])
    p = case_params(c)
    for k in range(4):
        t, add = (int(v) for v in c[f"tf{k}_t"])
        got = orc.tf_score(s, c["Nn"], p, tt(c["x0"]), tt(c["tf_noise"]), t, add)
        assert torch.equal(got, tt(c[f"tf{k}_score"])), (k, t, add)


@pytest.mark.parametrize("tag", CASES)
def test_reverse_steps_bit_exact(tag):
    c = load_case(tag)
    s = orc.make_schedule(c["T"], c["sched"])
    Nn = c["Nn"]
    for k in range(4):
        t = int(c[f"rs{k}_t"])
        mo = tt(c[f"rs{k}_model_out"])
        if Nn:
            z_new = orc.gaussian_p_step(s, mo[:, :Nn], tt(c[f"rs{k}_z"]), t, tt(c[f"rs{k}_n"]))
            assert z_new.dtype == torch.float64
            assert torch.equal(z_new, tt(c[f"rs{k}_z_new"])), (k, t)
        if c["C"]:
            idx = orc.multinomial_p_step(s, c["K"], mo[:, Nn:], tt(c[f"rs{k}_z_idx"]), t, tt(c[f"rs{k}_u"]))
            assert torch.equal(idx, tt(c[f"rs{k}_z_idx_new"])), (k, t)
            tv = torch.full((mo.shape[0],), t, dtype=torch.long)
            logp = orc.q_posterior(s, orc.block_log_softmax(mo[:, Nn:], c["K"]), orc.log_onehot(tt(c[f"rs{k}_z_idx"]), c["K"]), tv, c["K"])
            assert torch.equal(logp, tt(c[f"rs{k}_logp"]))


@pytest.mark.parametrize("tag", ["A_numeric", "B_ragged", "M_mixed", "L_catonly"])
def test_full_chain_bit_exact(tag):
    c = load_case(tag)
    s = orc.make_schedule(int(c["ch_T"]), c["sched"])
    first32 = bool(c["ch_first_noise_is_f32"])
    noises = [tt(n) for n in c["ch_noise"]]
    if first32 and len(noises):
        noises[0] = noises[0].float()
    us = [tt(v) for v in c["ch_u"]] if c["C"] else [None] * len(noises)
    u0 = tt(c["ch_u0"]) if c["C"] else None
    got = orc.sample_chain(s, c["K"], c["Nn"], case_params(c), tt(c["ch_z0"]), u0, noises, us)
    want = tt(c["ch_sample"])
    assert got.dtype == want.dtype == torch.float64
    assert torch.equal(got, want)


@pytest.mark.parametrize("which", ["cs", "sg"])
def test_guided_chains_bit_exact(which):
    """conditional_sample (GMD:1319) and sample(..., conditioning_function) (GMD:1370, condition_mean GMD:585) of the
    reference, every draw injected: the oracle reproduces both bit for bit."""
    from tests.helpers import guidance_function, load_guided

    d = load_guided()
    s = orc.make_schedule(d["T"], "cosine")
    cond = guidance_function(tt(d["centers"]), d["scale"], d["T"])
    noises = [tt(n) for n in d[f"{which}_noise"]]
    noises[0] = noises[0].float()
    got = orc.sample_chain(s, np.array([0]), d["Nn"], d["params"], tt(d[f"{which}_z0"]), None, noises, [None] * len(noises),
                           conditioning_function=cond, model_kwargs={"y": tt(d["targets"])})
    want = tt(d[f"{which}_sample"])
    assert got.dtype == want.dtype == torch.float64
    assert torch.equal(got, want)


@pytest.mark.parametrize("tag", ["A_numeric", "B_ragged"])
def test_ddim_chain_bit_exact(tag):
    """sample_ddim (GMD:1259-1316, eta = 0) of the reference, every draw injected: bit for bit, fp32 throughout."""
    from tests.helpers import load_ddim

    c = load_ddim(tag)
    s = orc.make_schedule(c["T"], c["sched"])
    us = [tt(v) for v in c["u"]] if c["C"] else [None] * c["T"]
    got = orc.sample_chain_ddim(s, c["K"], c["Nn"], case_params(c), tt(c["z0"]), tt(c["u0"]) if c["C"] else None,
                                [tt(n) for n in c["noise"]], us)
    want = tt(c["sample"])
    assert got.dtype == want.dtype == torch.float32
    assert torch.equal(got, want)


def test_scan_recurrence_equals_logcumsumexp():
    """The scalar fp64 recurrence the HIP kernel implements == torch's block LSE, bit for bit."""
    g = torch.Generator().manual_seed(7)
    K = np.array([3, 7, 10, 2, 12])
    off = orc.offsets_of(K)
    un = torch.randn((64, int(off[-1])), generator=g) * 6 - 20
    un[:, 5] = orc.LOG_TINY * 2
    want = orc.block_lse_cumulative(un, K)[:, off[:-1]]
    for r in range(un.shape[0]):
        got = orc.scan_lse_scalar(un[r].numpy(), off)
        assert np.array_equal(got, want[r].numpy()), r


@pytest.mark.parametrize("tag", ["A_numeric", "B_ragged", "M_mixed"])
def test_gaussian_kl_rows_bit_exact(tag):
    """GaussianLossType.KL (GMD:431-484): the restatement equals the reference's _gaussian_loss bit for bit (fp64) on the
    stored batch, with t = 0 everywhere (decoder NLL) and with t = 1 everywhere (tests/golden/kl_*.npz)."""
    import os

    c = load_case(tag)
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", f"kl_{tag}.npz"))
    Nn = c["Nn"]
    s = orc.make_schedule(c["T"], c["sched"])
    x0, mo, xt, t = tt(c["x0"])[:, :Nn], tt(c["ml_model_out"])[:, :Nn], tt(c["ml_x_num_t"]), tt(c["ml_t"])
    for name, tv in (("t", t), ("t0", torch.zeros_like(t)), ("t1", torch.ones_like(t))):
        got = orc.gaussian_kl_rows(s, mo, x0, xt, tv)
        assert got.dtype == torch.float64
        assert torch.equal(got, torch.from_numpy(k[f"kl_rows_{name}"])), name
    # the whole reference call: same draws -> same Gaussian mean (the denoiser runs on this host's BLAS: 1e-6)
    tc, eps = torch.from_numpy(k["call_t"]), torch.from_numpy(k["call_eps"])
    if c["C"] == 0:
        x_t = s.sqrt_alphas_cumprod[tc][:, None] * x0 + s.sqrt_one_minus_alphas_cumprod[tc][:, None] * eps
        out = orc.mlp_forward(case_params(c), x_t, tc)
        mean = float(orc.gaussian_kl_rows(s, out, x0, x_t, tc).mean())
        assert abs(mean - float(k["call_means"][1])) <= 1e-6 * abs(float(k["call_means"][1]))


def test_classifier_restatement_matches_the_reference_classifier():
    """The oracle's eval-mode Classifier and conditioning function against the reference's own (classifier_A_numeric.npz,
    generated by running MODEL:45-120 and SYNTH:429-451): logits, guidance gradients (plain and remove_first_col) and the
    conditional_sample chain they drive."""
    from tests.helpers import load_classifier

    c = load_classifier()
    x, t, y = tt(c["x"]), tt(c["t"]), tt(c["y"])
    torch.testing.assert_close(orc.classifier_logits(c["state"], x, t), tt(c["logits"]), rtol=1e-6, atol=1e-6)
    cond = orc.classifier_conditioning_function(c["state"], c["scale"])
    torch.testing.assert_close(cond(x.double(), t, y=y), tt(c["grad"]), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(cond(tt(c["x_rfc"]), t, y=y, remove_first_col=True), tt(c["grad_rfc"]), rtol=1e-5, atol=1e-7)
    s = orc.make_schedule(c["T"], "cosine")
    targets = tt(c["targets"])
    noise = [tt(n) for n in c["cs_noise"]]
    noise[0] = noise[0].float()
    got = orc.sample_chain(s, np.array([0]), c["Nn"], c["params"], tt(c["cs_z0"]), None, noise, [None] * c["T"], conditioning_function=cond,
                           model_kwargs={"y": targets})
    torch.testing.assert_close(got, tt(c["cs_sample"]), rtol=1e-6, atol=1e-6)


def test_ddim_family_bit_exact():
    """The rest of the DDIM family (tests/golden/ddimx_A_numeric.npz, generated from the reference): gaussian_ddim_sample with
    eta > 0 (GMD:1119), with condition_score (GMD:617-661), sample_ddim with a conditioning function (GMD:1259),
    gaussian_ddim_reverse_sample (GMD:1189) and multinomial_ddim_step with eta > 0 (GMD:1218): the oracle's restatements give
    the reference's outputs bit for bit."""
    from tests.helpers import guidance_function, load_ddimx

    c = load_ddimx()
    Nn, T, B = c["Nn"], c["T"], c["z0"].shape[0]
    s = orc.make_schedule(T, "cosine")
    params = case_params(c)
    cond = guidance_function(tt(c["centers"]), c["scale"], T)
    y = tt(c["targets"])

    def chain(z, noises, eta, guided, y_):
        for i, t in enumerate(reversed(range(T))):
            tv = torch.full((B,), t, dtype=torch.long)
            out = orc.mlp_forward(params, z.float(), tv).float()
            g = cond(z, tv, y=y_) if guided else None
            z = orc.gaussian_ddim_step(s, out, z, t, noises[i], eta=eta, cond_grad=g)
        return z

    assert torch.equal(chain(tt(c["z0"]), [tt(n) for n in c["eta_noise"]], c["eta"], False, y), tt(c["eta_sample"]))
    assert torch.equal(chain(tt(c["z0"]), [tt(n) for n in c["etacs_noise"]], c["eta_cs"], True, y), tt(c["etacs_sample"]))
    zeros = [torch.zeros((B, Nn))] * T
    assert torch.equal(chain(tt(c["sd_z0"]), zeros, 0.0, True, y), tt(c["sd_sample"]))
    z = tt(c["x_data"])
    for t in range(T):
        z = orc.gaussian_ddim_reverse_step(s, orc.mlp_forward(params, z.float(), torch.full((B,), t, dtype=torch.long)).float(), z, t)
    assert torch.equal(z, tt(c["rev_sample"]))
    sm = orc.make_schedule(c["md_T"], "cosine")
    for t in (0, 1, 7, 15):
        got = orc.multinomial_ddim_step(sm, c["md_K"], tt(c["md_out_cat"]), tt(c["md_z_idx"]), t, tt(c[f"md_u_{t}"]), eta=c["md_eta"])
        assert torch.equal(got, tt(c[f"md_idx_{t}"])), t


def test_edm_restatement_bit_exact():
    """TabSyn's EDM denoiser (models/tabsyn/model/modules.py:23-185, utils.py:20-87; tests/golden/edm_latent24.npz, generated by
    running the reference): the oracle's MLPDiffusion.forward, Preconditioner.forward, EDMLoss and Euler-Heun sampler give the
    reference's outputs bit for bit on the CPU that generated the fixture (1e-6 elsewhere: fp32 BLAS summation order)."""
    from tests.helpers import load_edm

    c = load_edm()
    p, dt = c["params"], c["dim_t"]
    x = tt(c["x"])

    def same(got, want):
        assert got.dtype == torch.float32 and got.shape == want.shape
        assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))

    same(orc.edm_forward(p, x, tt(c["labels"]), dt), tt(c["forward"]))
    same(orc.edm_denoise(p, x, tt(c["sigma"]), dt), tt(c["denoise"]))
    same(orc.edm_denoise(p, x, torch.tensor(0.37), dt), tt(c["denoise_scalar"]))
    loss = orc.edm_loss(p, x, tt(c["loss_rnd"]), tt(c["loss_noise"]), dt)
    same(loss, tt(c["loss"]))
    assert abs(float(loss.mean(-1).mean()) - float(c["loss_mean"])) <= 1e-6 * float(c["loss_mean"])
    got = orc.edm_sample(p, tt(c["s_latents"]), [tt(n) for n in c["s_noise"]], dt, c["num_steps"])
    assert float((got - tt(c["s_sample"])).abs().max()) <= 1e-4 * float(tt(c["s_sample"]).abs().max())


def test_label_conditioned_denoiser_restatement():
    """IsTargetConditioned.EMBEDDING with classes (MODEL:583-585, 610-612; tests/golden/label_label3.npz, generated by running the
    reference): the oracle's denoiser with and without ``y``, the mixed_loss rows with ``y`` (GMD:1042) and the ``sample`` chain
    with the drawn labels (GMD:1401-1406) reproduce the reference's outputs."""
    from tests.helpers import load_label

    c = load_label()
    p, K, Nn, T = c["params"], c["K"], c["Nn"], c["T"]
    x, t, y = tt(c["fw_x"]), tt(c["fw_t"]), tt(c["fw_y"])

    def same(got, want, tol=1e-6):
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))

    same(orc.mlp_forward(p, x, t, y=y), tt(c["fw_out_y"]))
    same(orc.mlp_forward(p, x, t), tt(c["fw_out_none"]))
    assert float((tt(c["fw_out_y"]) - tt(c["fw_out_none"])).abs().max()) > 1e-2  # the labels matter in this fixture
    s = orc.make_schedule(T, "cosine")
    lm, lg, m_mean, g_mean = orc.mixed_loss(s, K, Nn, p, tt(c["ml_x0"]), tt(c["ml_t"]), tt(c["ml_eps"]), tt(c["ml_u"]), y=tt(c["ml_y"]))
    same(lg, tt(c["ml_lg_rows"]), 1e-5)
    assert abs(float(lm.mean()) - float(c["ml_means"][0])) <= 1e-4 * abs(float(c["ml_means"][0]))
    s2 = orc.make_schedule(c["ch_T"], "cosine")
    noise = [tt(n) for n in c["ch_noise"]]
    noise[0] = noise[0].float()
    got = orc.sample_chain(s2, K, Nn, p, tt(c["ch_z0"]), tt(c["ch_u0"]), noise, [tt(u) for u in c["ch_u"]], y=tt(c["ch_y"]))
    want = tt(c["ch_sample"])
    assert torch.equal(got[:, Nn:], want[:, Nn:])
    same(got[:, :Nn], want[:, :Nn], 1e-4)


def test_x0_and_direct_parametrisations_bit_exact():
    """GaussianParametrization.X0 + Parametrization.DIRECT (GMD:399-408, 822-832): the oracle's branches against the reference
    constructed with both (tools/make_golden.py --param-only) - mixed_loss rows and the whole call's means, single reverse steps
    (numeric state, the reference's p_pred rows, the drawn codes), a whole ancestral chain, the KL Gaussian loss, a DDIM chain and
    reverse-DDIM steps: bit for bit.  The same checks with the default flags must FAIL on these fixtures (the flags do something)."""
    import os

    from tests.helpers import load_ddim

    c = load_case("P_x0direct")
    K, Nn = c["K"], c["Nn"]
    s = orc.make_schedule(c["T"], c["sched"])
    p = case_params(c)
    x0, t, eps, u = tt(c["x0"]), tt(c["ml_t"]), tt(c["ml_eps"]), tt(c["ml_u"])
    out, xt_idx = tt(c["ml_model_out"]), tt(c["ml_xt_idx"])
    lm, lg = orc.loss_rows(s, K, Nn, out, x0, xt_idx, eps, t, direct=True)
    assert torch.equal(lm, tt(c["ml_lm_rows"])) and torch.equal(lg, tt(c["ml_lg_rows"]))
    assert not torch.equal(orc.loss_rows(s, K, Nn, out, x0, xt_idx, eps, t)[0], tt(c["ml_lm_rows"]))
    _, _, m_mean, g_mean = orc.mixed_loss(s, K, Nn, p, x0, tt(c["ml_ref_call_t"]), eps, u, x0_param=True, direct=True)
    assert np.float32(m_mean) == c["ml_ref_call_means"][0] and np.float32(g_mean) == c["ml_ref_call_means"][1]
    for k in range(4):
        tk = int(c[f"rs{k}_t"])
        mo = tt(c[f"rs{k}_model_out"])
        z_new = orc.gaussian_p_step(s, mo[:, :Nn], tt(c[f"rs{k}_z"]), tk, tt(c[f"rs{k}_n"]), x0_param=True)
        assert z_new.dtype == torch.float64 and torch.equal(z_new, tt(c[f"rs{k}_z_new"])), (k, tk)
        assert not torch.equal(orc.gaussian_p_step(s, mo[:, :Nn], tt(c[f"rs{k}_z"]), tk, tt(c[f"rs{k}_n"])), z_new)
        idx = orc.multinomial_p_step(s, K, mo[:, Nn:], tt(c[f"rs{k}_z_idx"]), tk, tt(c[f"rs{k}_u"]), direct=True)
        assert torch.equal(idx, tt(c[f"rs{k}_z_idx_new"])), (k, tk)
        assert torch.equal(orc.block_log_softmax(mo[:, Nn:], K), tt(c[f"rs{k}_logp"]))
    s2 = orc.make_schedule(int(c["ch_T"]), c["sched"])
    noises = [tt(n) for n in c["ch_noise"]]
    noises[0] = noises[0].float()
    got = orc.sample_chain(s2, K, Nn, p, tt(c["ch_z0"]), tt(c["ch_u0"]), noises, [tt(v) for v in c["ch_u"]], x0_param=True, direct=True)
    assert got.dtype == torch.float64 and torch.equal(got, tt(c["ch_sample"]))
    # KL Gaussian loss
    kf = np.load(os.path.join(os.path.dirname(__file__), "golden", "kl_P_x0direct.npz"))
    mo, xt = out[:, :Nn], tt(c["ml_x_num_t"])
    for name, tv in (("t", t), ("t0", torch.zeros_like(t)), ("t1", torch.ones_like(t))):
        gk = orc.gaussian_kl_rows(s, mo, x0[:, :Nn], xt, tv, x0_param=True)
        assert gk.dtype == torch.float64 and torch.equal(gk, torch.from_numpy(kf[f"kl_rows_{name}"])), name
    assert not torch.equal(orc.gaussian_kl_rows(s, mo, x0[:, :Nn], xt, t), torch.from_numpy(kf["kl_rows_t"]))
    # DDIM chain (the multinomial DDIM step has no parametrisation: GMD:1211-1257 reads the log-softmax directly) and reverse DDIM
    d = load_ddim("P_x0direct")
    sd = orc.make_schedule(d["T"], d["sched"])
    gd = orc.sample_chain_ddim(sd, d["K"], d["Nn"], case_params(d), tt(d["z0"]), tt(d["u0"]), [tt(n) for n in d["noise"]], [tt(v) for v in d["u"]],
                               x0_param=True)
    assert gd.dtype == torch.float32 and torch.equal(gd, tt(d["sample"]))
    r = np.load(os.path.join(os.path.dirname(__file__), "golden", "ddimrev_P_x0direct.npz"))
    sr = orc.make_schedule(int(r["T"]), "cosine")
    for k in range(3):
        zn = orc.gaussian_ddim_reverse_step(sr, tt(r[f"r{k}_model_out"]), tt(r[f"r{k}_z"]), int(r[f"r{k}_t"]), x0_param=True)
        assert torch.equal(zn, tt(r[f"r{k}_z_new"])), k
