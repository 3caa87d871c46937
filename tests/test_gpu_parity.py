"""GPU: the HIP path (through the C ABI) against the oracle and the reference's golden vectors.

Bars (DESIGN.md "Parity"):
  * numeric noising, Gaussian loss, TF scores, denoiser (fp32 mode), fp64 reverse state: <= 1e-4 relative
    (measured ~1e-6);
  * categorical draws: identical to the reference given the same uniforms, except Gumbel near-ties
    (the reference's own top-2 margin below TIE_MARGIN), which must be rarer than 1e-3 of decisions;
  * multinomial loss: batch mean <= 1e-4 relative; per record |d| <= 1e-4 |ref| + 2 * conditioning floor
    (tests/test_conditioning.py shows why a plain per-record 1e-4 is not well-posed).
"""
import sys

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as orc
from tests.helpers import CASES, case_params, load_case, oracle_flags, tt

pytestmark = pytest.mark.gpu
TIE_MARGIN = 2e-5
RTOL = 1e-4


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def build(c, precision="fp32", T=None):
    import midst_toolkit_amd as mtk

    dp = mtk.DiffusionParameters(list(c["d_layers"]), 0.0)
    net = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, dp)
    net.load_state_dict(case_params(c), strict=True)
    net.precision = precision
    fl = oracle_flags(c)  # fixtures of the X0 / DIRECT parametrisations build the drop-in the way the reference was built
    d = mtk.GaussianMultinomialDiffusion(c["K"], c["Nn"], net, num_timesteps=T or c["T"],
                                         scheduler_type=mtk.SchedulerType(c["sched"]), device=dev(),
                                         gaussian_parametrization=mtk.GaussianParametrization.X0 if fl.get("x0_param") else mtk.GaussianParametrization.EPS,
                                         parametrization=mtk.Parametrization.DIRECT if fl.get("direct") else mtk.Parametrization.X0)
    d = d.to(dev())
    d.precision = precision
    return d


def close(got, want, rtol=RTOL, atol=1e-6):
    got, want = got.double().cpu(), want.double().cpu()
    err = (got - want).abs()
    ok = err <= rtol * want.abs() + atol
    assert bool(ok.all()), f"max abs err {float(err.max()):.3e}, max rel {float((err / want.abs().clamp_min(1e-9)).max()):.3e}"


def check_categorical(got_idx, want_idx, logits, u, K, exact=False):
    """``exact`` (the reference fixtures, reference math): every draw equals the reference's.  Otherwise (fast math, big
    random batches): equal, or a Gumbel near-tie in the reference's own arithmetic, rarer than 1e-3 of the decisions."""
    got_idx, want_idx = got_idx.cpu().long(), want_idx.cpu().long()
    diff = got_idx != want_idx
    n_bad = int(diff.sum())
    if exact:
        assert n_bad == 0, f"{n_bad} of {diff.numel()} categorical draws differ from the reference's (bit-exact bar)"
        return
    assert n_bad <= max(1, int(1e-3 * diff.numel())), f"{n_bad} of {diff.numel()} categorical draws differ"
    if n_bad:
        off = orc.offsets_of(K)
        g = -torch.log(-torch.log(u + 1e-30) + 1e-30) + logits
        for r, k in zip(*np.nonzero(diff.numpy())):
            blk = g[r, off[k]:off[k + 1]]
            top2 = torch.topk(blk, 2).values
            assert float(top2[0] - top2[1]) < TIE_MARGIN, f"row {r} block {k}: differs without a near-tie ({blk.tolist()})"


def ranking_report(got, want):
    """How the ordering of the records by ``got`` differs from their ordering by ``want`` ([B, P], one ranking per noise
    draw): positions that differ, the largest relative gap between two reference scores that come out swapped, Spearman
    rho, Kendall tau (by sampling pairs on big batches) and the overlap of the top 10 % sets."""
    got, want = got.double().cpu(), want.double().cpu()
    B, P = got.shape
    order_g = torch.argsort(got, dim=0, stable=True)
    order_w = torch.argsort(want, dim=0, stable=True)
    moved = int((order_g != order_w).sum())
    ref_sorted = torch.gather(want, 0, order_g)
    drop = (ref_sorted[:-1] - ref_sorted[1:]) / ref_sorted[:-1].abs().clamp_min(1e-30)
    worst_swap = float(drop.clamp_min(0).max()) if B > 1 else 0.0
    rank_g = torch.argsort(order_g, dim=0).double()
    rank_w = torch.argsort(order_w, dim=0).double()
    rg, rw = rank_g - rank_g.mean(0), rank_w - rank_w.mean(0)
    spearman = float(((rg * rw).sum(0) / (rg.norm(dim=0) * rw.norm(dim=0)).clamp_min(1e-30)).min()) if B > 1 else 1.0
    gen = torch.Generator().manual_seed(0)
    n_pairs = min(200_000, B * (B - 1) // 2) if B > 1 else 0
    if n_pairs and B * (B - 1) // 2 <= 200_000:
        i, j = torch.triu_indices(B, B, 1)
    elif n_pairs:
        i = torch.randint(0, B, (n_pairs,), generator=gen)
        j = torch.randint(0, B, (n_pairs,), generator=gen)
    if n_pairs:
        sg = torch.sign(got[i] - got[j])
        sw = torch.sign(want[i] - want[j])
        valid = (sw != 0)
        kendall = float(((sg * sw)[valid]).mean())
        discordant = float(((sg * sw)[valid] < 0).double().mean())
    else:
        kendall, discordant = 1.0, 0.0
    k = max(1, B // 10)
    top = min(len(set(order_g[-k:, p].tolist()) & set(order_w[-k:, p].tolist())) / k for p in range(P))
    return {"positions_moved": moved, "positions": B * P, "worst_swapped_gap": worst_swap, "spearman": spearman, "kendall": kendall,
            "discordant_pairs": discordant, "top10_overlap": top}


def check_ranking(got, want, tie=4e-6):
    """Membership-score ranking (north_star: bit-exact given the same injected noise): ordering the records by the GPU
    score must be the reference's ordering, except between records whose REFERENCE scores are closer than `tie` relative
    - the documented ties: fp32 summation-order noise of the denoiser GEMMs, ~1e-6, decides those in the reference too.
    Every position that differs must belong to such a tie group."""
    got, want = got.cpu(), want.cpu()
    order = torch.argsort(got, dim=0, stable=True)
    ref_sorted = torch.gather(want, 0, order)
    drop = ref_sorted[:-1] - ref_sorted[1:]
    assert bool((drop <= tie * ref_sorted[:-1].abs()).all()), float((drop / ref_sorted[:-1].abs()).max())
    want_order = torch.argsort(want, dim=0, stable=True)
    moved = order != want_order
    if bool(moved.any()):  # a moved position holds a record whose reference score ties (within `tie`) with a neighbour's
        ws = torch.gather(want, 0, want_order)
        gap_lo = torch.cat([torch.full_like(ws[:1], float("inf")), (ws[1:] - ws[:-1]) / ws[1:].abs()], 0)
        gap_hi = torch.cat([(ws[1:] - ws[:-1]) / ws[:-1].abs(), torch.full_like(ws[:1], float("inf"))], 0)
        in_tie = (gap_lo <= tie) | (gap_hi <= tie)
        assert bool(in_tie[moved].all()), f"{int((moved & ~in_tie).sum())} ranking positions differ outside documented ties"
    return int(moved.sum())


def colmajor(x):
    """[B, D] host tensor → device [D][B] buffer."""
    return x.t().contiguous().to(dev())


@pytest.mark.parametrize("tag", CASES)
def test_q_sample(tag):
    c = load_case(tag)
    d = build(c)
    eng = d.engine()
    B, D, C = c["x0"].shape[0], c["D"], c["C"]
    x0, t = tt(c["x0"]).to(dev()), tt(c["ml_t"]).to(dev())
    eps = tt(c["ml_eps"]).to(dev()) if c["Nn"] else None
    u = tt(c["ml_u"]).to(dev()) if C else None
    for strides, shape in (((1, B), (D, B)), ((D, 1), (B, D))):
        x_in = torch.full(shape, float("nan"), device=dev())
        idx = torch.zeros((max(C, 1), B), dtype=torch.int32, device=dev())
        eng.q_sample(x0, t, eps, u, x_in, strides, idx)
        x_in_rm = x_in.t() if strides == (1, B) else x_in
        close(x_in_rm[:, : c["Nn"]], tt(c["ml_x_num_t"]), rtol=1e-6)
        if C:
            s = orc.make_schedule(c["T"], c["sched"])
            logits = orc.q_pred(s, orc.log_onehot(tt(c["x0"])[:, c["Nn"]:].long(), c["K"]), tt(c["ml_t"]), orc.log_k_expanded(c["K"]))
            check_categorical(idx[:C].t(), tt(c["ml_xt_idx"]), logits, tt(c["ml_u"]), c["K"], exact=True)
            same = (idx[:C].t().cpu().long() == tt(c["ml_xt_idx"])).all(dim=1)
            assert torch.equal(x_in_rm.cpu()[same][:, c["Nn"]:], tt(c["ml_model_in"])[same][:, c["Nn"]:])


@pytest.mark.parametrize("tag", CASES)
def test_mlp_forward_fp32(tag):
    c = load_case(tag)
    d = build(c)
    x, t = tt(c["ml_model_in"]).to(dev()), tt(c["ml_t"]).to(dev())
    with torch.no_grad():
        out = d._denoise_fn(x, t)
    want = tt(c["ml_model_out"])
    scale = float(want.abs().max())
    close(out, want, rtol=RTOL, atol=1e-5 * scale)
    # column-major in / out and a scalar timestep
    B, D = x.shape
    eng = d.engine()
    o2 = torch.empty((D, B), device=dev())
    eng.mlp_forward(x.t().contiguous(), t, out=o2, x_strides=(1, B), out_strides=(1, B), B=B)
    assert torch.equal(o2.t().contiguous(), out)
    tv = torch.full((B,), 7, dtype=torch.long)
    o3 = eng.mlp_forward(x, 7)
    close(o3, orc.mlp_forward(case_params(c), tt(c["ml_model_in"]), tv), rtol=RTOL, atol=1e-5 * scale)


@pytest.mark.parametrize("tag", CASES + ["P_x0direct"])
def test_loss_rows_with_injected_model_out(tag):
    c = load_case(tag)
    d = build(c)
    eng = d.engine()
    B, C, Nn = c["x0"].shape[0], c["C"], c["Nn"]
    x0, t = tt(c["x0"]).to(dev()), tt(c["ml_t"]).to(dev())
    mo = colmajor(tt(c["ml_model_out"]))
    idx = tt(c["ml_xt_idx"]).t().contiguous().int().to(dev()) if C else None
    eps = tt(c["ml_eps"]).to(dev()) if Nn else None
    lm = torch.empty(B, device=dev()) if C else None
    lg = torch.empty(B, device=dev()) if Nn else None
    eng.loss_rows(mo, (1, B), x0, idx, eps, t, lm, lg)
    if Nn:
        close(lg, tt(c["ml_lg_rows"]), rtol=1e-5)
    if C:
        want = tt(c["ml_lm_rows"])
        s = orc.make_schedule(c["T"], c["sched"])
        floor = orc.multinomial_loss_floor(s, c["K"], Nn, tt(c["ml_model_out"]), tt(c["x0"]), tt(c["ml_xt_idx"]), tt(c["ml_t"]))
        err = (lm.cpu().double() - want.double()).abs()
        assert bool((err <= RTOL * want.abs().double() + 2 * floor).all()), float((err / (RTOL * want.abs().double() + 2 * floor)).max())
        assert abs(float(lm.mean().cpu()) - float(want.mean())) <= RTOL * abs(float(want.mean()))
        # the t == 0 rows take the decoder-NLL branch (rows 0 and 3 of every fixture)
        assert int(tt(c["ml_t"])[0]) == 0 and bool(torch.isfinite(lm[0].cpu()))


@pytest.mark.parametrize("tag", CASES + ["P_x0direct"])
def test_mixed_loss_end_to_end(tag):
    c = load_case(tag)
    d = build(c)
    x0 = tt(c["x0"]).to(dev())
    t = tt(c["ml_t"]).to(dev())
    eps = tt(c["ml_eps"]).to(dev()) if c["Nn"] else None
    u = tt(c["ml_u"]).to(dev()) if c["C"] else None
    lm, lg = d.mixed_loss_rows(x0, timestep=t, noise=eps, uniform=u)
    if c["Nn"]:
        close(lg, tt(c["ml_lg_rows"]), rtol=RTOL)
    else:
        assert lg.shape == (1,) and float(lg) == 0.0 and not lg.is_cuda
    if c["C"]:
        want = tt(c["ml_lm_rows"])
        assert abs(float(lm.mean().cpu()) - float(want.mean())) <= RTOL * abs(float(want.mean()))
    else:
        assert lm.shape == (1,) and float(lm) == 0.0 and not lm.is_cuda
    # the unmodified entry point, drawing like the reference does
    torch.manual_seed(3)
    with torch.no_grad():
        m, g = d.mixed_loss(x0, {"y": torch.zeros(x0.shape[0], dtype=torch.long, device=dev())})
    assert m.ndim == 0 and g.ndim == 0 and bool(torch.isfinite(m)) and bool(torch.isfinite(g))


@pytest.mark.parametrize("tag", ["A_numeric", "N_tf"])
def test_tf_score_fp32(tag):
    from midst_toolkit_amd import scoring

    c = load_case(tag)
    d = build(c)
    x0 = tt(c["x0"]).to(dev())
    noise = tt(c["tf_noise"])
    P = noise.shape[0]
    for k in range(4):
        t, add = (int(v) for v in c[f"tf{k}_t"])
        got = scoring.mixed_loss(d, x0, {"y": None}, noise.tolist(), P, add, t)
        want = tt(c[f"tf{k}_score"])
        assert got.shape == want.shape
        close(got, want, rtol=RTOL)
        if add == 0:  # membership-score ranking over the records, per noise draw
            check_ranking(got.cpu(), want)
    ts = [int(c[f"tf{k}_t"][0]) for k in (0, 1, 3)]
    many = scoring.score_records(d, x0, noise.to(dev()), ts, 0)
    for j, k in enumerate((0, 1, 3)):
        close(many[j], tt(c[f"tf{k}_score"]), rtol=RTOL)


@pytest.mark.parametrize("tag", CASES + ["P_x0direct"])
def test_reverse_steps(tag):
    c = load_case(tag)
    d = build(c)
    eng = d.engine()
    B, Nn, C, D, S = c["x0"].shape[0], c["Nn"], c["C"], c["D"], c["S"]
    s = orc.make_schedule(c["T"], c["sched"])
    for k in range(4):
        t = int(c[f"rs{k}_t"])
        mo = colmajor(tt(c[f"rs{k}_model_out"]))
        z = tt(c[f"rs{k}_z"]).double().t().contiguous().to(dev()) if Nn else None
        z_idx = tt(c[f"rs{k}_z_idx"]).t().contiguous().int().to(dev()) if C else None
        n = tt(c[f"rs{k}_n"]).to(dev()) if Nn else None
        u = tt(c[f"rs{k}_u"]).to(dev()) if C else None
        x_next = torch.full((D, B), float("nan"), device=dev())
        eng.p_sample_step(B, t, k == 0, mo, (1, B), z, z_idx, n, u, x_next, (1, B))
        if Nn:
            close(z.t(), tt(c[f"rs{k}_z_new"]), rtol=1e-12 if k else 1e-6, atol=1e-12 if k else 1e-7)
            assert torch.equal(x_next[:Nn].t().cpu(), z.t().float().cpu())
        if C:
            check_categorical(z_idx.t(), tt(c[f"rs{k}_z_idx_new"]), tt(c[f"rs{k}_logp"]), tt(c[f"rs{k}_u"]), c["K"], exact=True)
            want_oh = orc.log_onehot(z_idx.t().cpu().long(), c["K"])
            assert torch.equal(x_next[Nn:].t().cpu(), want_oh)


@pytest.mark.parametrize("tag", ["A_numeric", "B_ragged", "M_mixed", "L_catonly", "K1_single", "P_x0direct"])
def test_full_chain(tag):
    c = load_case(tag)
    Tc = int(c["ch_T"])
    d = build(c, T=Tc)
    Nn, C = c["Nn"], c["C"]
    noise = [tt(n).to(dev()) for n in c["ch_noise"]]
    if bool(c["ch_first_noise_is_f32"]) and noise:
        noise[0] = noise[0].float()
    inj = {"z0": tt(c["ch_z0"]).to(dev()), "noise": noise, "y": torch.zeros(c["ch_z0"].shape[0], dtype=torch.long)}
    if C:
        inj["u0"] = tt(c["ch_u0"]).to(dev())
        inj["u"] = [tt(v).to(dev()) for v in c["ch_u"]]
    got, outs = d.sample(c["ch_z0"].shape[0], torch.ones(1), _injected=inj)
    want = tt(c["ch_sample"])
    assert got.dtype == torch.float64 and got.shape == want.shape and not got.is_cuda
    same = (got[:, Nn:] == want[:, Nn:]).all(dim=1) if C else torch.ones(got.shape[0], dtype=torch.bool)
    # bit-exact bar on the reference's own chain: every category of every row (a flipped Gumbel near-tie would change
    # that row's whole trajectory; none flips on these fixtures)
    assert bool(same.all()), f"{int((~same).sum())} of {same.numel()} rows differ in a category from the reference's chain"
    if Nn:
        # The chain multiplies eps_hat by sqrt_recipm1_alphas_cumprod (up to 2e4) at every step, so the
        # reference's own fp32 denoiser rounding already moves the result: measure that (fp32 vs fp64
        # denoiser inside the oracle) and allow 3x it on top of the 1e-4 relative bar.
        s = orc.make_schedule(Tc, c["sched"])
        cpu = lambda v: v.cpu() if isinstance(v, torch.Tensor) else v  # noqa: E731
        args = (s, c["K"], Nn, case_params(c), cpu(inj["z0"]), cpu(inj.get("u0")), [cpu(n) for n in noise],
                [cpu(v) for v in inj["u"]] if C else [None] * len(noise))
        ref64 = orc.sample_chain(*args, mlp_dtype=torch.float64, **oracle_flags(c))
        agree = same & ((ref64[:, Nn:] == want[:, Nn:]).all(dim=1) if C else same)
        own = float((ref64[agree][:, :Nn] - want[agree][:, :Nn]).abs().max())
        close(got[same][:, :Nn], want[same][:, :Nn], rtol=RTOL, atol=max(1e-5, 3 * own))


def test_reverse_ddim_step_x0_parametrisation():
    """gaussian_ddim_reverse_step (GMD:1160-1186) with GaussianParametrization.X0 (the denoiser output is pred_xstart, GMD:404-405):
    single steps with injected denoiser output against the reference's (tests/golden/ddimrev_P_x0direct.npz); with the default
    parametrisation the same inputs give another state."""
    import os

    import midst_toolkit_amd as mtk

    r = np.load(os.path.join(os.path.dirname(__file__), "golden", "ddimrev_P_x0direct.npz"))
    Nn, T = int(r["Nn"]), int(r["T"])
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([int(v) for v in r["d_layers"]], 0.0))
    for x0_param in (True, False):
        d = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev(),
                                             gaussian_parametrization=mtk.GaussianParametrization.X0 if x0_param else mtk.GaussianParametrization.EPS).to(dev())
        eng = d.engine()
        for k in range(3):
            z = tt(r[f"r{k}_z"]).double().t().contiguous().to(dev())
            B = z.shape[1]
            eng.ddim_reverse_step(B, int(r[f"r{k}_t"]), colmajor(tt(r[f"r{k}_model_out"])), (1, B), z, None, (1, B))
            if x0_param:
                close(z.t(), tt(r[f"r{k}_z_new"]), rtol=1e-6, atol=1e-6)
            else:
                assert float((z.t().cpu() - tt(r[f"r{k}_z_new"]).double()).abs().max()) > 1e-2


@pytest.mark.parametrize("which", ["cs", "sg"])
def test_guided_chains(which):
    """conditional_sample (GMD:1319) and sample(..., conditioning_function) with the reference's own draws injected,
    against the reference's outputs (tests/golden/guided_A_numeric.npz) — same bar as test_full_chain.  The
    conditioning function runs on the device tensors the chain hands it, like the classifier gradient of SYNTH:429."""
    import midst_toolkit_amd as mtk
    from tests.helpers import guidance_function, load_guided

    g = load_guided()
    Nn, T = g["Nn"], g["T"]
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(g["d_layers"], 0.0))
    net.load_state_dict(g["params"], strict=True)
    d = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev()).to(dev())
    cond = guidance_function(tt(g["centers"]), g["scale"], T)
    targets = tt(g["targets"])
    noise = [tt(n).to(dev()) for n in g[f"{which}_noise"]]
    noise[0] = noise[0].float()
    inj = {"z0": tt(g[f"{which}_z0"]).to(dev()), "noise": noise}
    seen = []

    def spy(features, timestep, **kw):
        seen.append((features.dtype, tuple(features.shape), features.is_cuda, int(timestep[0])))
        return cond(features, timestep, **kw)

    if which == "cs":
        got, outs = d.conditional_sample(targets, model_kwargs={"y": targets}, conditioning_function=spy, _injected=inj)
        assert torch.equal(outs["y"].cpu(), targets.long())
    else:
        inj["y"] = tt(g["sg_y"])
        got, outs = d.sample(targets.shape[0], torch.ones(3), model_kwargs={"y": targets}, conditioning_function=spy, _injected=inj)
        assert torch.equal(outs["y"].cpu(), tt(g["sg_y"]).long())
    # the callback saw what the reference's would: [B, Nn] on the device, fp32 at t = T-1 and fp64 after, t descending
    assert [s_[3] for s_ in seen] == list(reversed(range(T)))
    assert seen[0][0] == torch.float32 and all(s_[0] == torch.float64 for s_ in seen[1:])
    assert all(s_[1] == (targets.shape[0], Nn) and s_[2] for s_ in seen)
    want = tt(g[f"{which}_sample"])
    assert got.dtype == torch.float64 and got.shape == want.shape and not got.is_cuda
    s = orc.make_schedule(T, "cosine")
    cpu_noise = [n.cpu() for n in noise]
    ref64 = orc.sample_chain(s, np.array([0]), Nn, g["params"], inj["z0"].cpu(), None, cpu_noise, [None] * T, mlp_dtype=torch.float64,
                             conditioning_function=cond, model_kwargs={"y": targets})
    own = float((ref64 - want).abs().max())
    close(got, want, rtol=RTOL, atol=max(1e-5, 3 * own))
    # and guidance does something: the unguided chain from the same draws lands elsewhere
    plain, _ = d.sample(targets.shape[0], torch.ones(3), _injected=dict(inj, y=torch.zeros(targets.shape[0], dtype=torch.long)))
    assert float((plain - want).abs().max()) > 1e-2


@pytest.mark.parametrize("tag", ["A_numeric", "B_ragged", "P_x0direct"])
def test_ddim_chain(tag):
    """sample_ddim (GMD:1259, eta = 0) with the reference's draws injected against the reference's output
    (tests/golden/ddim_*.npz; the oracle reproduces it bit for bit): the whole-chain bar of test_full_chain."""
    from tests.helpers import load_ddim

    c = load_ddim(tag)
    Nn, C, T, B = c["Nn"], c["C"], c["T"], c["z0"].shape[0]
    d = build(c, T=T)
    inj = {"z0": tt(c["z0"]).to(dev()), "noise": [None] * T, "y": torch.zeros(B, dtype=torch.long)}
    if C:
        inj["u0"] = tt(c["u0"]).to(dev())
        inj["u"] = [tt(v).to(dev()) for v in c["u"]]
    got, _ = d.sample_ddim(B, torch.ones(1), _injected=inj)
    want = tt(c["sample"])
    assert got.dtype == torch.float32 and got.shape == want.shape and not got.is_cuda
    same = (got[:, Nn:] == want[:, Nn:]).all(dim=1) if C else torch.ones(B, dtype=torch.bool)
    assert bool(same.all()), f"{int((~same).sum())} of {B} rows differ in a category from the reference's DDIM chain"
    s = orc.make_schedule(T, c["sched"])
    us = [tt(v) for v in c["u"]] if C else [None] * T
    ref64 = orc.sample_chain_ddim(s, c["K"], Nn, case_params(c), tt(c["z0"]), tt(c["u0"]) if C else None, [tt(n) for n in c["noise"]], us,
                                  mlp_dtype=torch.float64, x0_param=oracle_flags(c).get("x0_param", False))
    agree = same & ((ref64[:, Nn:] == want[:, Nn:]).all(dim=1) if C else same)
    own = float((ref64[agree][:, :Nn] - want[agree][:, :Nn]).abs().max())
    close(got[same][:, :Nn], want[same][:, :Nn], rtol=RTOL, atol=max(1e-5, 3 * own))
    # sample_all(ddim=True) routes here (GMD:1452)
    torch.manual_seed(1)
    x, y = d.sample_all(40, 32, torch.ones(1), ddim=True)
    assert x.shape == (40, Nn + C) and x.dtype == torch.float32 and y.shape == (40,) and not bool(x.isnan().any())
    if C:
        codes = x[:, Nn:]
        assert bool((codes == codes.round()).all()) and bool((codes >= 0).all()) and bool((codes < torch.tensor(c["K"]).float()).all())


def test_ddim_family():
    """f4: the rest of the DDIM family against the reference's outputs (tests/golden/ddimx_A_numeric.npz): gaussian_ddim_sample with
    eta > 0 (GMD:1119) without and with score-based guidance (condition_score, GMD:617-661), sample_ddim with a conditioning
    function, gaussian_ddim_reverse_sample (GMD:1189), and one multinomial_ddim_step with eta > 0 (GMD:1218) through the C ABI."""
    from tests.helpers import guidance_function, load_ddimx

    c = load_ddimx()
    Nn, T, B = c["Nn"], c["T"], c["z0"].shape[0]
    d = build(c, T=T)
    cond = guidance_function(tt(c["centers"]), c["scale"], T)
    y = tt(c["targets"]).to(dev())
    z0 = tt(c["z0"]).to(dev())
    got = d.gaussian_ddim_sample(z0, T, {}, eta=c["eta"], _injected=[tt(n) for n in c["eta_noise"]])
    assert got.dtype == torch.float32
    close(got, tt(c["eta_sample"]), rtol=RTOL, atol=1e-5)
    got = d.gaussian_ddim_sample(z0, T, {}, eta=c["eta_cs"], model_kwargs={"y": y}, conditioning_function=cond, _injected=[tt(n) for n in c["etacs_noise"]])
    close(got, tt(c["etacs_sample"]), rtol=RTOL, atol=1e-5)
    assert float((got.cpu() - tt(c["eta_sample"])).abs().max()) > 1e-2
    inj = {"z0": tt(c["sd_z0"]).to(dev()), "noise": [None] * T, "y": tt(c["sd_y"])}
    got, outs = d.sample_ddim(B, torch.ones(3), model_kwargs={"y": y}, conditioning_function=cond, _injected=inj)
    close(got, tt(c["sd_sample"]), rtol=RTOL, atol=1e-5)
    got = d.gaussian_ddim_reverse_sample(tt(c["x_data"]).to(dev()), T, {})
    close(got, tt(c["rev_sample"]), rtol=RTOL, atol=1e-5)
    # philox mode runs, and eta > 0 adds noise
    d.rng = "philox"
    a = d.gaussian_ddim_sample(z0, T, {}, eta=0.0)
    b = d.gaussian_ddim_sample(z0, T, {}, eta=0.0)
    e = d.gaussian_ddim_sample(z0, T, {}, eta=0.7)
    assert torch.equal(a, b) and float((a - e).abs().max()) > 1e-3 and not bool(e.isnan().any())
    # multinomial_ddim_step, eta > 0: the categorical draw of one step, straight through the engine
    import midst_toolkit_amd as mtk

    K = c["md_K"]
    net = mtk.MLPDiffusion(int(K.sum()) + 1, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([32], 0.0))
    dc = mtk.GaussianMultinomialDiffusion(K, 1, net, num_timesteps=c["md_T"], device=dev()).to(dev())
    eng = dc.engine()
    Bc, S, Cc = c["md_out_cat"].shape[0], int(K.sum()), len(K)
    mo = torch.cat([torch.zeros((Bc, 1)), tt(c["md_out_cat"])], 1).to(dev()).contiguous()
    for t in (0, 1, 7, 15):
        z = torch.zeros((1, Bc), dtype=torch.float64, device=dev())
        z_idx = tt(c["md_z_idx"]).t().contiguous().to(device=dev(), dtype=torch.int32)
        eng.ddim_step(Bc, t, mo, (S + 1, 1), z, z_idx, tt(c[f"md_u_{t}"]).to(dev()).contiguous(), None, (0, 0), eta=c["md_eta"],
                      noise=torch.zeros((Bc, 1), device=dev()))
        want = tt(c[f"md_idx_{t}"]).t().to(torch.int32)
        assert torch.equal(z_idx.cpu(), want), (t, int((z_idx.cpu() != want).sum()))


def test_tabsyn_edm_denoiser():
    """f4: TabSyn's EDM denoiser on HIP (mtk_edm_forward / mtk_edm_denoise / mtk_edm_loss) against the reference's own outputs
    (tests/golden/edm_latent24.npz): MLPDiffusion.forward, Preconditioner.forward with per-row and scalar sigma, EDMLoss and
    Model.forward on the reference's draws (<= 1e-4 relative to the row scale), and the Euler-Heun sampler with its draws injected
    (a 10-step chain whose first sigma is 80: the bar is relative to the sample's scale)."""
    from midst_toolkit_amd import tabsyn
    from tests.helpers import load_edm

    c = load_edm()
    net = tabsyn.MLPDiffusion(c["d_in"], c["dim_t"])
    net.load_state_dict(c["params"], strict=True)
    net = net.to(dev())
    mdl = tabsyn.Model(net)
    x = tt(c["x"]).to(dev())

    def near(got, want, rel=RTOL):
        got, want = got.cpu().double(), want.double()
        assert got.shape == want.shape
        scale = want.abs().amax(dim=-1, keepdim=True).clamp_min(1e-3)
        assert float(((got - want).abs() / scale).max()) <= rel, float(((got - want).abs() / scale).max())

    with pytest.raises(NotImplementedError):
        net(x, tt(c["labels"]).to(dev()))  # parameters require grad and autograd is on: the HIP path is forward only
    with torch.no_grad():
        near(net(x, tt(c["labels"]).to(dev())), tt(c["forward"]))
        near(mdl.preconditioner(x, tt(c["sigma"]).to(dev())), tt(c["denoise"]))
        near(mdl.preconditioner(x, torch.tensor(0.37)), tt(c["denoise_scalar"]))
        inj = (tt(c["loss_rnd"]), tt(c["loss_noise"]))
        near(mdl.loss_fn(mdl.preconditioner, x, inj), tt(c["loss"]))
        assert abs(float(mdl(x, inj)) - float(c["loss_mean"])) <= RTOL * float(c["loss_mean"])
        got = tabsyn.sample(mdl.preconditioner, x.shape[0], c["d_in"], num_steps=c["num_steps"], device=dev(),
                            _injected={"latents": tt(c["s_latents"]), "noise": [tt(n) for n in c["s_noise"]]})
        want = tt(c["s_sample"])
        assert float((got.cpu() - want).abs().max()) <= 1e-3 * float(want.abs().max())
        torch.manual_seed(5)
        free = tabsyn.sample(mdl.preconditioner, 300, c["d_in"], num_steps=c["num_steps"], device=dev())  # not a tile multiple
        assert free.shape == (300, c["d_in"]) and not bool(free.isnan().any())
        # TabSyn's own shape (pipeline.py:141: dim_t = 1024) runs
        big = tabsyn.MLPDiffusion(40, 1024).to(dev())
        out = tabsyn.Preconditioner(big)(torch.randn((513, 40), device=dev()), torch.full((513,), 0.5, device=dev()))
        assert out.shape == (513, 40) and not bool(out.isnan().any())


def test_tabsyn_edm_loss_trains():
    """models/tabsyn/model/modules.py:165-219 under autograd (VERDICT r2 missing 1): Model.forward with grad enabled builds a graph
    whose backward is the library's chain (mtk_edm_loss_keep / mtk_edm_backward).  On the reference fixture's weights and draws
    the loss equals the reference's and all fourteen gradient tensors equal the oracle's autograd through the restated modules
    (<= 3e-4 relative L2 each); gradients are identical run to run; torch.optim.Adam steps on them reduce the loss and the handle
    takes the new weights in place; a second differentiable forward invalidates the first one's backward."""
    from midst_toolkit_amd import tabsyn
    from tests.helpers import load_edm

    c = load_edm()
    net = tabsyn.MLPDiffusion(c["d_in"], c["dim_t"])
    net.load_state_dict(c["params"], strict=True)
    net = net.to(dev())
    mdl = tabsyn.Model(net)
    x = tt(c["x"]).to(dev())
    inj = (tt(c["loss_rnd"]), tt(c["loss_noise"]))
    p_ref = {k: tt(v).clone().requires_grad_(True) for k, v in c["params"].items()}
    want = orc.edm_loss(p_ref, tt(c["x"]), inj[0], inj[1], c["dim_t"]).mean(-1).mean()
    want.backward()
    loss = mdl(x, inj)
    assert loss.requires_grad
    loss.backward()
    assert abs(float(loss.detach()) - float(c["loss_mean"])) <= RTOL * float(c["loss_mean"])
    got = {k: v.grad.detach().cpu().clone() for k, v in net.named_parameters()}
    assert set(got) == set(p_ref)
    for k, v in p_ref.items():
        num, den = float((got[k].double() - v.grad.double()).norm()), float(v.grad.double().norm())
        assert num <= 3e-4 * den + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e}"
    net.zero_grad()
    mdl(x, inj).backward()
    assert all(torch.equal(v.grad.cpu(), got[k]) for k, v in net.named_parameters())
    # the reference's loop: zero_grad -> loss -> backward -> step; same draws every step, so the loss must fall
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    handle = net._handle
    losses = []
    for _ in range(8):
        opt.zero_grad()
        step_loss = mdl(x, inj)
        step_loss.backward()
        opt.step()
        losses.append(float(step_loss.detach()))
    assert all(b < a for a, b in zip(losses, losses[1:])) and losses[-1] < 0.95 * losses[0], losses
    assert net._handle is handle  # new weights went into the same handle
    with torch.no_grad():
        assert abs(float(mdl(x, inj)) - losses[-1]) > 0  # evaluation after the last step sees the stepped weights
    a = mdl(x, inj)
    b = mdl(x, inj)
    with pytest.raises(RuntimeError, match="another differentiable"):
        a.backward()
    b.backward()



def test_label_conditioned_denoiser():
    """IsTargetConditioned.EMBEDDING with classes (MODEL:583-585, 610-612) against the reference's own outputs
    (tests/golden/label_label3.npz): the label embedding is extra rows of the time-embedding table
    (mtk_mlp_set_label_embedding); forward with / without y, mixed_loss rows with y, the sample chain with drawn labels."""
    import midst_toolkit_amd as mtk
    from helpers import load_label

    c = load_label()
    K, Nn, T, D = c["K"], c["Nn"], c["T"], c["Nn"] + int(c["K"].sum())

    def make(T_):
        net = mtk.MLPDiffusion(D, c["n_labels"], mtk.IsTargetConditioned.EMBEDDING, mtk.DiffusionParameters(list(c["d_layers"]), 0.0))
        net.load_state_dict(c["params"], strict=True)
        d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T_, scheduler_type=mtk.SchedulerType("cosine"), device=dev()).to(dev())
        return d

    d = make(T)
    with torch.no_grad():  # used stand-alone first (1000 timesteps by default): the context is rebuilt when the diffusion hands over its T
        d._denoise_fn.to(dev())(tt(c["fw_x"]).to(dev()), tt(c["fw_t"]).to(dev()), tt(c["fw_y"]).to(dev()))
    d.engine()  # hands the diffusion's T to the denoiser (its table has (labels + 1) x T rows)
    x, t, y = tt(c["fw_x"]).to(dev()), tt(c["fw_t"]).to(dev()), tt(c["fw_y"]).to(dev())
    with torch.no_grad():
        close(d._denoise_fn(x, t, y), tt(c["fw_out_y"]), rtol=RTOL, atol=1e-5)
        close(d._denoise_fn(x, t), tt(c["fw_out_none"]), rtol=RTOL, atol=1e-5)
    lm, lg = d.mixed_loss_rows(tt(c["ml_x0"]).to(dev()), timestep=tt(c["ml_t"]).to(dev()), noise=tt(c["ml_eps"]).to(dev()),
                               uniform=tt(c["ml_u"]).to(dev()), y=tt(c["ml_y"]).to(dev()))
    close(lg, tt(c["ml_lg_rows"]), rtol=RTOL)
    assert abs(float(lm.mean().cpu()) - float(c["ml_means"][0])) <= RTOL * abs(float(c["ml_means"][0]))
    torch.manual_seed(3)
    with torch.no_grad():  # the unmodified entry point: y travels in ``outputs``
        m, g = d.mixed_loss(tt(c["ml_x0"]).to(dev()), {"y": tt(c["ml_y"]).to(dev())})
    assert bool(torch.isfinite(m)) and bool(torch.isfinite(g))
    # the chain: labels as the reference drew them from target_dist
    d2 = make(c["ch_T"])
    noise = [tt(n).to(dev()) for n in c["ch_noise"]]
    noise[0] = noise[0].float()
    inj = {"z0": tt(c["ch_z0"]).to(dev()), "noise": noise, "y": tt(c["ch_y"]).long(), "u0": tt(c["ch_u0"]).to(dev()),
           "u": [tt(v).to(dev()) for v in c["ch_u"]]}
    got, outs = d2.sample(c["ch_z0"].shape[0], torch.tensor([0.2, 0.5, 0.3]), _injected=inj)
    want = tt(c["ch_sample"])
    assert torch.equal(outs["y"].cpu().long(), tt(c["ch_y"]).long())
    assert torch.equal(got[:, Nn:], want[:, Nn:])
    close(got[:, :Nn], want[:, :Nn], rtol=RTOL, atol=1e-4)
    # a label-conditioned net trains too (tests/test_gpu_train.py::test_label_conditioned_denoiser_trains); what stays off the path says so
    mtk.ClavaDDPMTrainer(d, None, 1e-3, 0.0, 1, device=dev())
    reg =mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.EMBEDDING, mtk.DiffusionParameters([64, 64], 0.0)).to(dev())
    with pytest.raises(NotImplementedError), torch.no_grad():
        reg(x, t, torch.zeros((x.shape[0], 1), device=dev()))


def test_dropout_configured_denoiser_runs_in_eval_mode():
    """MODEL:162-180: a denoiser configured with dropout > 0 loads and, in eval mode (Dropout = identity: how scoring and sampling
    call it), gives the dropout-free network's outputs; in training mode a plain forward() refuses instead of silently dropping
    nothing (the inference kernels carry no mask; the training iteration applies its own: tests/test_gpu_train.py)."""
    import midst_toolkit_amd as mtk

    c = load_case("A_numeric")
    net = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.CONCAT, mtk.DiffusionParameters(list(c["d_layers"]), 0.1))
    net.load_state_dict(case_params(c), strict=True)
    d = mtk.GaussianMultinomialDiffusion(c["K"], c["Nn"], net, num_timesteps=c["T"], device=dev()).to(dev())
    x, t = tt(c["ml_model_in"]).to(dev()), tt(c["ml_t"]).to(dev())
    with pytest.raises(NotImplementedError), torch.no_grad():
        net(x, t)
    d.eval()
    with torch.no_grad():
        close(net(x, t), tt(c["ml_model_out"]))


@pytest.mark.parametrize("math", ["reference", "fast"])
def test_wide_categorical_blocks_stress_shape(math):
    """SURVEY section 8's stress variant (blocks of 100 levels; here K = [100, 37, 100, 3], S = 240, D = 243 > 128: the layered
    denoiser, block caches that stream): noising, per-record losses and one reverse step against the oracle on injected draws -
    the oracle is the restatement pinned bit for bit on the reference's fixtures.  Categorical draws identical (reference
    math; the fast kernels within their near-tie allowance), losses within the conditioning floor, batch means <= 1e-4."""
    import midst_toolkit_amd as mtk

    K, Nn, T, B = np.asarray([100, 37, 100, 3]), 3, 1000, 600
    C, S, D = len(K), int(K.sum()), Nn + int(K.sum())
    params = orc.init_mlp_params(D, [128, 128], seed=0)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([128, 128], 0.0))
    net.load_state_dict(params, strict=True)
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev()).to(dev())
    d.math = math
    g = torch.Generator().manual_seed(77)
    x0 = torch.cat([torch.randn((B, Nn), generator=g)] + [torch.randint(0, int(k), (B, 1), generator=g).float() for k in K], 1)
    t = torch.randint(0, T, (B,), generator=g)
    t[:4] = torch.tensor([0, 1, T - 1, 0])
    eps, u = torch.randn((B, Nn), generator=g), torch.rand((B, S), generator=g)
    s = orc.make_schedule(T, "cosine")
    xt, idx = orc.q_sample_fused(s, K, Nn, x0, t, eps, u)
    out = orc.mlp_forward(params, orc.model_input(xt, idx, K), t)
    lm_ref, lg_ref = orc.loss_rows(s, K, Nn, out, x0, idx, eps, t)
    eng = d.engine()
    x_in = torch.empty((D, B), device=dev())
    got_idx = torch.zeros((C, B), dtype=torch.int32, device=dev())
    eng.q_sample(x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()), x_in, (1, B), got_idx)
    close(x_in[:Nn].t(), xt, rtol=1e-6)
    logits = orc.q_pred(s, orc.log_onehot(x0[:, Nn:].long(), K), t, orc.log_k_expanded(K))
    global TIE_MARGIN
    saved = TIE_MARGIN
    try:
        TIE_MARGIN = 2e-5 if math == "reference" else 2e-4
        check_categorical(got_idx.t(), idx, logits, u, K, exact=(math == "reference"))
    finally:
        TIE_MARGIN = saved
    # losses on the ORACLE's categorical state and model output (so a near-tie flip above cannot leak in)
    lm = torch.empty(B, device=dev())
    lg = torch.empty(B, device=dev())
    eng.loss_rows(colmajor(out), (1, B), x0.to(dev()), idx.t().contiguous().int().to(dev()), eps.to(dev()), t.to(dev()), lm, lg)
    close(lg, lg_ref, rtol=1e-5)
    floor = orc.multinomial_loss_floor(s, K, Nn, out, x0, idx, t)
    err = (lm.cpu().double() - lm_ref.double()).abs()
    # reference math: the fixtures' bar (2 floors + slack for the longer scan); the fast kernels take the block log-sum-exp
    # exactly instead of through the reference's cumulative scan, whose own rounding grows with the row's 240 columns
    n_floor = 3 if math == "reference" else 12
    worst = float((err / (RTOL * lm_ref.abs().double() + n_floor * floor)).max())
    print(f"wide blocks, math={math}: worst per-record error / bar = {worst:.3f}, mean rel err {abs(float(lm.mean().cpu()) - float(lm_ref.mean())) / abs(float(lm_ref.mean())):.2e}")
    assert worst <= 1.0, worst
    assert abs(float(lm.mean().cpu()) - float(lm_ref.mean())) <= RTOL * abs(float(lm_ref.mean()))
    # the denoiser at D = 243 (layered bf16 path; fp32 exact path) and the drop-in's mixed_loss_rows end to end
    got = torch.empty((D, B), device=dev())
    eng.mlp_forward(x_in, t.to(dev()), out=got, precision="fp32", x_strides=(1, B), out_strides=(1, B), B=B)
    same = (got_idx.t().cpu().long() == idx).all(dim=1)
    close(got.t()[same.to(dev())], out[same], rtol=RTOL, atol=1e-5 * float(out.abs().max()))
    d.precision = net.precision = "bf16"
    eng = d.engine()
    eng.mlp_forward(x_in, t.to(dev()), out=got, precision="bf16", x_strides=(1, B), out_strides=(1, B), B=B)
    assert float((got.t()[same.to(dev())].cpu() - out[same]).abs().max()) <= BF16_TOL * float(out.abs().max())


def test_wide_block_reverse_step_draws_the_posterior():
    """The fast reverse step on a 100-level block (streamed in chunks, closed-form normaliser, inverse-CDF draw from one Philox
    word): 400k rows with the same state and logits must draw categories with the frequencies of the reference's posterior
    (oracle.q_posterior on the same logits), within 5 binomial standard deviations per level."""
    import midst_toolkit_amd as mtk

    K, Nn, T, B, t = np.asarray([100, 5]), 1, 1000, 400_000, 400
    D = Nn + int(K.sum())
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([32], 0.0))
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev()).to(dev())
    d.rng, d.math = "philox", "fast"
    eng = d.engine()
    g = torch.Generator().manual_seed(9)
    logits = torch.randn((1, int(K.sum())), generator=g) * 2.0
    cur = torch.tensor([[17, 3]])
    s = orc.make_schedule(T, "cosine")
    tv = torch.full((1,), t, dtype=torch.long)
    logp = orc.q_posterior(s, orc.block_log_softmax(logits, K), orc.log_onehot(cur, K), tv, K)[0]
    want = torch.exp(logp.double())
    out = torch.cat([torch.zeros((1, Nn)), logits], 1).t().contiguous().to(dev()).expand(D, B).contiguous()
    z = torch.zeros((Nn, B), dtype=torch.float64, device=dev())
    zi = cur.t().contiguous().to(device=dev(), dtype=torch.int32).expand(2, B).contiguous()
    eng.p_sample_step(B, t, False, out, (1, B), z, zi, None, None, None, (0, 0), seed=12345)
    off = 0
    for k, Kk in enumerate(K):
        freq = torch.bincount(zi[k].cpu().long(), minlength=int(Kk)).double() / B
        p = want[off:off + int(Kk)]
        assert abs(float(p.sum()) - 1.0) < 1e-5
        sd = torch.sqrt(p * (1 - p) / B)
        assert bool(((freq - p).abs() <= 5 * sd + 1e-6).all()), (k, float(((freq - p).abs() / (5 * sd + 1e-6)).max()))
        off += int(Kk)


def test_conditional_sample_rejects_categorical_tables():
    c = load_case("B_ragged")
    d = build(c, T=4)
    with pytest.raises(ValueError):
        d.conditional_sample(torch.zeros((4, 1), dtype=torch.long))


def test_sample_all_and_philox_mode():
    c = load_case("B_ragged")
    d = build(c, T=12)
    torch.manual_seed(0)
    x, y = d.sample_all(50, 32, torch.ones(1))
    assert x.shape == (50, c["Nn"] + c["C"]) and x.dtype == torch.float64 and y.shape == (50,)
    cats = x[:, c["Nn"]:]
    assert bool((cats >= 0).all()) and bool((cats < tt(c["K"]).double()).all()) and bool((cats == cats.round()).all())
    d.rng = "philox"
    x2, _ = d.sample_all(40, 40, torch.ones(1))
    assert x2.shape == (40, c["Nn"] + c["C"]) and bool(torch.isfinite(x2).all())
    lm, lg = d.mixed_loss_rows(tt(c["x0"]).to(dev()))
    assert bool(torch.isfinite(lm).all()) and bool(torch.isfinite(lg).all())
    # in-kernel noise has the right moments: Gaussian loss of an untrained net ~ E[(eps - out)^2]
    big = torch.randn((8192, c["Nn"])).clamp(-5.2, 5.2)
    rows = torch.cat([big] + [torch.randint(0, int(k), (8192, 1)).float() for k in c["K"]], 1).to(dev())
    eng = d.engine()
    x_in = torch.empty((c["D"], 8192), device=dev())
    idx = torch.empty((c["C"], 8192), dtype=torch.int32, device=dev())
    t0 = torch.full((8192,), d.num_timesteps - 1, dtype=torch.long, device=dev())
    eng.q_sample(rows, t0, None, None, x_in, (1, 8192), idx, seed=99)
    xn = x_in[: c["Nn"]]
    assert abs(float(xn.mean())) < 0.05 and abs(float(xn.std()) - 1.0) < 0.05  # x_T ~ N(0, 1)
    for k, K in enumerate(c["K"]):  # x_T categories ~ uniform
        freq = torch.bincount(idx[k].long().cpu(), minlength=int(K)).float() / 8192
        assert float((freq - 1.0 / int(K)).abs().max()) < 0.03


def test_empty_batches():
    """Zero records in, zero rows out: every entry point of the path returns without a launch (the reference's ops on
    empty tensors give empty tensors too)."""
    from midst_toolkit_amd import scoring

    c = load_case("B_ragged")
    d = build(c)
    D, Nn = c["D"], c["Nn"]
    x0 = tt(c["x0"]).to(dev())[:0]
    with torch.no_grad():
        out = d._denoise_fn(torch.empty((0, D), device=dev()), torch.empty((0,), dtype=torch.int64, device=dev()))
    assert out.shape == (0, D)
    lm, lg = d.mixed_loss_rows(x0, timestep=torch.empty((0,), dtype=torch.int64, device=dev()), noise=torch.empty((0, Nn), device=dev()),
                               uniform=torch.empty((0, c["S"]), device=dev()))
    assert lm.shape == (0,) and lg.shape == (0,)
    a = load_case("A_numeric")
    da = build(a)
    sc = scoring.score_records(da, tt(a["x0"]).to(dev())[:0], tt(a["tf_noise"]).to(dev()), [3, 7], 0)
    assert sc.shape[1] == 0 and sc.numel() == 0
    with pytest.raises(RuntimeError):  # like the reference: torch.multinomial (GMD:1401) refuses to draw zero labels
        d.sample(0, torch.ones(1))
    eng = d.engine()  # the kernels of a reverse step on zero rows
    z, zi = torch.empty((Nn, 0), dtype=torch.float64, device=dev()), torch.empty((c["C"], 0), dtype=torch.int32, device=dev())
    e0 = torch.empty((D, 0), device=dev())
    eng.sample_init(0, None, None, z, zi, e0, (1, 0), 5, 0)
    eng.p_sample_step(0, 3, False, e0, (1, 0), z, zi, None, None, e0, (1, 0), 5, 0)
    smp = torch.empty((0, Nn + c["C"]), dtype=torch.float64, device=dev())
    eng.sample_decode(0, z, zi, smp)


def test_batches_beyond_32_bit_element_indices():
    """2^24 + 5 records of shape M: the column-major intermediates hold 2.0e9 elements (past 2^31) and 8 GB each.  In-kernel
    Philox noise is keyed by (seed, row0 + row), so the last rows of the big batch must equal, bit for bit, a small batch of
    those rows submitted with row0 = their offset - through noising, the fused bf16 denoiser and the loss kernel."""
    c = load_case("M_mixed")
    d = build(c, precision="bf16")
    d.rng, d.math = "philox", "fast"
    eng = d.engine()
    Nn, C, D, K = c["Nn"], c["C"], c["D"], c["K"]
    B, n = (1 << 24) + 5, 300
    g = torch.Generator(device=dev()).manual_seed(8)
    x0 = torch.cat([torch.randn((B, Nn), device=dev(), generator=g)] +
                   [torch.randint(0, int(k), (B, 1), device=dev(), generator=g).float() for k in K], 1)
    t = torch.randint(0, c["T"], (B,), device=dev(), generator=g)

    def run(x0_, t_, rows, row0):
        x_in = torch.empty((D, rows), device=dev())
        idx = torch.empty((C, rows), dtype=torch.int32, device=dev())
        out = torch.empty((D, rows), device=dev())
        lm, lg = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
        eng.q_sample(x0_, t_, None, None, x_in, (1, rows), idx, 77, row0)
        eng.mlp_forward(x_in, t_, out=out, precision="bf16", x_strides=(1, rows), out_strides=(1, rows), B=rows)
        eng.loss_rows(out, (1, rows), x0_, idx, None, t_, lm, lg, 77, row0)
        return x_in, idx, out, lm, lg

    big = run(x0, t, B, 0)
    assert bool(torch.isfinite(big[3]).all()) and bool(torch.isfinite(big[4]).all())
    for lo in (B - n, (1 << 23) + 11):  # the tail (past element 2^31 of every column block) and the middle
        small = run(x0[lo:lo + n].contiguous(), t[lo:lo + n].contiguous(), n, lo)
        for a_, b_ in zip(big[:3], small[:3]):
            assert torch.equal(a_[:, lo:lo + n], b_), lo
        assert torch.equal(big[3][lo:lo + n], small[3]) and torch.equal(big[4][lo:lo + n], small[4]), lo
    # one reverse step on the same batch (fp64 state, Philox draws keyed by the global row)
    def step(rows, row0, out_cm, z, zi):
        x_next = torch.empty((D, rows), device=dev())
        eng.p_sample_step(rows, 400, False, out_cm, (1, rows), z, zi, None, None, x_next, (1, rows), 91, row0)
        return x_next

    z = torch.randn((Nn, B), device=dev(), generator=g, dtype=torch.float64)
    zi = big[1].clone()
    lo = B - n
    zs, zis = z[:, lo:].clone().contiguous(), zi[:, lo:].clone().contiguous()
    xn_big = step(B, 0, big[2], z, zi)
    xn_small = step(n, lo, big[2][:, lo:].contiguous(), zs, zis)
    assert torch.equal(xn_big[:, lo:], xn_small) and torch.equal(z[:, lo:], zs) and torch.equal(zi[:, lo:], zis)
    del big, z, zi, xn_big
    torch.cuda.empty_cache()


def test_bad_arguments_raise():
    c = load_case("A_numeric")
    d = build(c)
    eng = d.engine()
    x = torch.zeros((4, c["D"]), device=dev())
    with pytest.raises(AssertionError):
        eng.mlp_forward(x, 10_000)  # t outside the schedule
    with pytest.raises(ValueError):
        d2 = build(load_case("B_ragged"))
        d2.engine().tf_score(torch.zeros((4, 6), device=dev()), torch.zeros((2, 6), device=dev()), [1])


def test_out_of_range_timesteps_poison_their_rows():
    """Per-row timesteps live in device memory the host never reads.  The reference's sinusoid embedding accepts any t;
    here emb is a [T] table and the schedule rows are [T] too, so a timestep outside [0, T) must be loud and safe: the
    kernels clamp it for addressing and the row's outputs come back NaN - every other row is untouched."""
    c = load_case("M_mixed")
    B, Nn, C, D, T = 300, c["Nn"], c["C"], c["D"], c["T"]
    x0, t, eps, u = _big_mixed_batch(c, B)
    bad_rows = [5, 77, 299]
    t_bad = t.clone()
    t_bad[5], t_bad[77], t_bad[299] = T, -1, 10**12
    ok = torch.ones(B, dtype=torch.bool)
    ok[bad_rows] = False
    for precision in ("fp32", "bf16"):
        d = build(c, precision=precision)
        eng = d.engine()
        xin = orc.model_input(*orc.q_sample_fused(orc.make_schedule(T, c["sched"]), c["K"], Nn, x0, t, eps, u), c["K"]).to(dev())
        good = eng.mlp_forward(xin, t.to(dev()), precision=precision)
        got = eng.mlp_forward(xin, t_bad.to(dev()), precision=precision)
        assert bool(got[~ok].isnan().all()) and torch.equal(got[ok], good[ok]), precision
        lm, lg = d.mixed_loss_rows(x0.to(dev()), timestep=t_bad.to(dev()), noise=eps.to(dev()), uniform=u.to(dev()))
        lm0, lg0 = d.mixed_loss_rows(x0.to(dev()), timestep=t.to(dev()), noise=eps.to(dev()), uniform=u.to(dev()))
        assert bool(lm[~ok].isnan().all()) and bool(lg[~ok].isnan().all())
        assert torch.equal(lm[ok], lm0[ok]) and torch.equal(lg[ok], lg0[ok])
        # in-kernel noise, math "fast": those kernels poison in their own stores (no separate pass behind them)
        d.rng = "philox"
        lm, lg = d.mixed_loss_rows(x0.to(dev()), timestep=t_bad.to(dev()), seed=9)
        lm0, lg0 = d.mixed_loss_rows(x0.to(dev()), timestep=t.to(dev()), seed=9)
        assert bool(lm[~ok].isnan().all()) and bool(lg[~ok].isnan().all())
        assert torch.equal(lm[ok], lm0[ok]) and torch.equal(lg[ok], lg0[ok]) and bool(torch.isfinite(lm0).all())
        eng.set_math("fast")
        xi = torch.empty((D, B), device=dev())
        idx = torch.empty((C, B), dtype=torch.int32, device=dev())
        eng.q_sample(x0.to(dev()), t_bad.to(dev()), None, None, xi, (1, B), idx, seed=9)
        assert bool(xi[:, ~ok].isnan().all()) and bool(torch.isfinite(xi[:, ok]).all())
        d.rng = "torch"
    # layered bf16 path (a shape the fused kernel does not cover)
    import midst_toolkit_amd as mtk

    net = mtk.MLPDiffusion(24, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([256, 256], 0.0)).to(dev())
    net.precision = "bf16"
    xs = torch.randn((B, 24), device=dev())
    with torch.no_grad():
        good, got = net(xs, t.to(dev())), net(xs, t_bad.to(dev()))
    assert bool(got[~ok].isnan().all()) and torch.equal(got[ok], good[ok])
    # the stand-alone Gaussian helpers: q_sample poisons, the MSE (which does not depend on t) does not
    xt = d.gaussian_q_sample(x0[:, :Nn].to(dev()), t_bad.to(dev()), eps.to(dev()))
    assert bool(xt[~ok].isnan().all()) and bool(torch.isfinite(xt[ok]).all())
    mse = d._gaussian_loss(xt[ok], x0[ok][:, :Nn].to(dev()), xt[ok], t_bad[ok].to(dev()), eps[ok].to(dev()))
    assert bool(torch.isfinite(mse).all())


# ------------------------------------------------------------------------------------------------
# bf16 mode (fused register-resident MFMA kernel).  Stated tolerance: operands are rounded to bf16
# (8 significant bits) at every layer, accumulation is fp32; outputs agree with the fp32 reference to
# BF16_TOL of the output scale, Tartan-Federer scores to BF16_TOL relative.
# ------------------------------------------------------------------------------------------------
BF16_TOL = 2e-2


@pytest.mark.parametrize("tag", ["M_mixed", "N_tf"])
def test_mlp_forward_bf16(tag):
    c = load_case(tag)
    d = build(c, precision="bf16")
    x, t = tt(c["ml_model_in"]).to(dev()), tt(c["ml_t"]).to(dev())
    with torch.no_grad():
        out = d._denoise_fn(x, t)
    want = tt(c["ml_model_out"])
    scale = float(want.abs().max())
    err = float((out.cpu() - want).abs().max())
    assert err <= BF16_TOL * scale, (err, scale)
    # against the same network evaluated with bf16-rounded operands in fp32 (isolates the kernel's
    # indexing from the rounding): must be much closer
    B, D = x.shape
    eng = d.engine()
    o2 = torch.empty((D, B), device=dev())
    eng.mlp_forward(x.t().contiguous(), t, out=o2, precision="bf16", x_strides=(1, B), out_strides=(1, B), B=B)
    assert torch.equal(o2.t().contiguous(), out)
    ref = bf16_emulated_forward(case_params(c), tt(c["ml_model_in"]), tt(c["ml_t"]))
    assert float((out.cpu() - ref).abs().max()) <= 2e-3 * scale, float((out.cpu() - ref).abs().max()) / scale
    # ragged batch sizes around the 128-row tile
    for n in (1, 31, 33, 129):
        o = eng.mlp_forward(x[:n].contiguous(), t[:n].contiguous(), precision="bf16")
        assert torch.equal(o, out[:n]), n


@pytest.mark.parametrize("shape", [(120, [1024] * 4), (142, [512, 1024, 1024, 1024, 1024, 512]), (40, [128, 256]), (200, [1536, 768, 2048])])
def test_mlp_forward_bf16_layered(shape):
    """Denoiser shapes outside the fused kernel (hidden width != 512, unequal widths, D > 128) run layer by layer
    on bf16 MFMA: same bars as the fused kernel."""
    import midst_toolkit_amd as mtk

    D, layers = shape
    params = orc.init_mlp_params(D, layers, seed=3)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0)).to(dev())
    net.load_state_dict(params, strict=True)
    net.precision = "bf16"
    g = torch.Generator().manual_seed(1)
    x = torch.randn((300, D), generator=g)
    t = torch.randint(0, 1000, (300,), generator=g)
    with torch.no_grad():
        out = net(x.to(dev()), t.to(dev()))
    want = orc.mlp_forward(params, x, t)
    scale = float(want.abs().max())
    assert float((out.cpu() - want).abs().max()) <= BF16_TOL * scale
    ref = bf16_emulated_forward(params, x, t)
    assert float((out.cpu() - ref).abs().max()) <= 2e-3 * scale, float((out.cpu() - ref).abs().max()) / scale
    eng = net.engine(dev())
    for n in (1, 127, 129):
        o = eng.mlp_forward(x[:n].to(dev()).contiguous(), t[:n].to(dev()).contiguous(), precision="bf16")
        assert torch.equal(o, out[:n]), n
    o2 = torch.empty((D, 300), device=dev())  # column-major in and out
    eng.mlp_forward(x.t().contiguous().to(dev()), t.to(dev()), out=o2, precision="bf16", x_strides=(1, 300), out_strides=(1, 300), B=300)
    assert torch.equal(o2.t().contiguous(), out)


@pytest.mark.parametrize("K,Nn", [([10] * 10, 20), ([3, 12, 16, 2], 5), ([4, 30, 7], 3), ([0], 24), ([4, 9, 16], 0)])
def test_caller_sized_reverse_step_equals_the_general_kernel(monkeypatch, K, Nn):
    """Round 3: for at most 16,384 rows a reverse step with in-kernel noise runs one lane per (record, item)
    (k_p_sample_step_items) instead of one lane per record.  Same expressions and draws (the block draw is one shared function):
    under the A/B switch MTK_STEP_ROWS the new state - fp64 numeric columns, category codes, the next input matrix - is the same
    bit for bit, at the first step (fp32 mean, GMD first-step semantics), in the middle, at t = 0, with classifier guidance and
    with injected Gaussian noise.  Blocks of 12 / 16 levels sit on the register variants' edges, 30 is streamed."""
    import midst_toolkit_amd as mtk

    K = np.asarray(K)
    C = 0 if int(K[0]) == 0 else len(K)
    D, T, B = Nn + int(K.sum()), 50, 1337
    torch.manual_seed(5)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([128, 128], 0.0))
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev()).to(dev())
    eng = d.engine()
    eng.set_math("fast")
    g = torch.Generator().manual_seed(9)
    out = (torch.randn((D, B), generator=g) * 2.0).to(dev())
    z0 = torch.randn((max(Nn, 1), B), generator=g, dtype=torch.float64).to(dev())
    zi0 = torch.stack([torch.randint(0, int(k), (B,), generator=g) for k in (K if C else [1])]).to(torch.int32).to(dev())
    grad = torch.randn((B, Nn), generator=g).to(dev()) if Nn else None
    noise = torch.randn((B, Nn), generator=g, dtype=torch.float64).to(dev()) if Nn else None

    def run():
        res = []
        for t, first, gr, nz in ((T - 1, 1, None, None), (17, 0, None, None), (0, 0, None, None), (23, 0, grad, None), (31, 0, None, noise)):
            z, zi, x = z0.clone(), zi0.clone(), torch.zeros((D, B), device=dev())
            eng.p_sample_step(B, t, first, out, (1, B), z, zi, nz, None, x, (1, B), seed=77, row0=4096, grad=gr)
            res.append((z, zi, x))
        return res

    new = run()
    monkeypatch.setenv("MTK_STEP_ROWS", "1")
    old = run()
    for a, b in zip(new, old):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    if C:
        assert not torch.equal(new[1][1], zi0)  # the step did draw


@pytest.mark.parametrize("shape,B", [((120, [1024] * 4), 1000), ((120, [1024] * 4), 4096), ((142, [512, 1024, 1024, 1024, 1024, 512]), 3900),
                                     ((40, [128, 256]), 77),
                                     # 2,048 < rows <= 12,288: the persistent kernel on 128 x 128 tiles (the 4,096- and 3,900-row cases above too)
                                     ((120, [1024] * 4), 8192), ((142, [512, 1024, 1024, 1024, 1024, 512]), 12001), ((40, [128, 256]), 5000),
                                     # 284 tiles of 256 x 256: a second round that would be a ninth full -> 128 x 128 tiles as well
                                     ((120, [1024] * 4), 18000), ((142, [512, 1024, 1024, 1024, 1024, 512]), 20000)])
def test_small_pass_layered_kernel_equals_the_big_tile(monkeypatch, shape, B):
    """Round 3: passes of at most 2,048 rows run the layered denoiser on 64 x 64 tiles (k_gemm_bf16_s: every CU busy from 1 k rows
    on - the batch sizes the reference's callers sample, SYNTH:79), passes of up to 12,288 rows on 128 x 128 tiles of the persistent
    kernel, instead of 256 x 256.  A re-tiling, not other arithmetic: same
    MFMA, same k order, accumulators started from the bias, the same epilogue - under the A/B switch MTK_LAYERED_BIG the outputs
    are the same bits, with per-record timesteps (time embedding gathered in the first block's epilogue) and with one timestep
    per launch (proj folded into the first block)."""
    import midst_toolkit_amd as mtk

    D, layers = shape
    params = orc.init_mlp_params(D, layers, seed=3)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0)).to(dev())
    net.load_state_dict(params, strict=True)
    net.precision = "bf16"
    g = torch.Generator().manual_seed(4)
    x = torch.randn((B, D), generator=g).to(dev())
    t = torch.randint(0, 1000, (B,), generator=g).to(dev())
    eng = net.engine(dev())
    small = [eng.mlp_forward(x, t, precision="bf16"), eng.mlp_forward(x, 417, precision="bf16")]
    monkeypatch.setenv("MTK_LAYERED_BIG", "1")
    big = [eng.mlp_forward(x, t, precision="bf16"), eng.mlp_forward(x, 417, precision="bf16")]
    for a, b in zip(small, big):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    want = orc.mlp_forward(params, x.cpu()[:300], t.cpu()[:300])
    assert float((small[0].cpu()[:300] - want).abs().max()) <= BF16_TOL * float(want.abs().max())


@pytest.mark.parametrize("shape", [(20, [512] * 4), (120, [512] * 4), (120, [1024] * 4), (142, [512, 1024, 1024, 1024, 1024, 512]), (40, [128, 256])])
def test_mlp_forward_bf16_one_timestep_launches(shape):
    """Launches that carry ONE timestep (every step of the reverse chain passes an integer, not a vector) run proj folded
    into the first hidden layer - the fused kernel's mode 2 and the layered path's folded first block: W1 Wp and
    W1 (emb[t] + b_proj) + b1 are formed in fp32 and rounded once.  Same function as the per-row path: both within the
    bf16 bar of the fp32 oracle, and within a few bf16 roundings of each other."""
    import midst_toolkit_amd as mtk

    D, layers = shape
    params = orc.init_mlp_params(D, layers, seed=5)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0)).to(dev())
    net.load_state_dict(params, strict=True)
    net.precision = "bf16"
    eng = net.engine(dev())
    g = torch.Generator().manual_seed(4)
    B = 1000
    x = torch.randn((B, D), generator=g)
    xd = x.to(dev())
    for ts in (0, 437, 999):
        tv = torch.full((B,), ts, dtype=torch.int64)
        want = orc.mlp_forward(params, x, tv)
        scale = float(want.abs().max())
        one = eng.mlp_forward(xd, ts, precision="bf16")
        per_row = eng.mlp_forward(xd, tv.to(dev()), precision="bf16")
        assert float((one.cpu() - want).abs().max()) <= BF16_TOL * scale, ts
        assert float((one - per_row).abs().max()) <= 6e-3 * scale, (ts, float((one - per_row).abs().max()) / scale)
        # a row's result does not depend on the batch it sits in; column-major in and out
        assert torch.equal(eng.mlp_forward(xd[130:391].contiguous(), ts, precision="bf16"), one[130:391]), ts
        o2 = torch.empty((D, B), device=dev())
        eng.mlp_forward(xd.t().contiguous(), ts, out=o2, precision="bf16", x_strides=(1, B), out_strides=(1, B), B=B)
        assert torch.equal(o2.t().contiguous(), one), ts


@pytest.mark.parametrize("shape", [(120, [1024] * 4), (142, [512, 1024, 1024, 1024, 1024, 512])])
def test_mlp_forward_bf16_layered_many_tiles(shape):
    """The layered GEMM is persistent: a workgroup streams the k-blocks of several output tiles through one LDS double
    buffer.  70,001 rows give every workgroup 4-5 tiles (and, with D = 142, an odd number of k-blocks per tile, so the
    buffer parity flips from tile to tile).  A row's result does not depend on the batch it sits in: windows of the big
    batch must equal separate small calls bit for bit (those are checked against the oracle above), and a sample of
    rows is checked against the oracle directly."""
    import midst_toolkit_amd as mtk

    D, layers = shape
    params = orc.init_mlp_params(D, layers, seed=3)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0)).to(dev())
    net.load_state_dict(params, strict=True)
    net.precision = "bf16"
    g = torch.Generator().manual_seed(2)
    B = 70_001
    x = torch.randn((B, D), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    xd, td = x.to(dev()), t.to(dev())
    with torch.no_grad():
        out = net(xd, td)
    assert bool(torch.isfinite(out).all())
    eng = net.engine(dev())
    for lo in (0, 256 * 97 + 13, 65_536 - 150, B - 300):
        o = eng.mlp_forward(xd[lo:lo + 300].contiguous(), td[lo:lo + 300].contiguous(), precision="bf16")
        assert torch.equal(o, out[lo:lo + 300]), lo
    pick = torch.randperm(B, generator=g)[:1500]
    want = orc.mlp_forward(params, x[pick], t[pick])
    scale = float(want.abs().max())
    assert float((out.cpu()[pick] - want).abs().max()) <= BF16_TOL * scale


# ------------------------------------------------------------------------------------------------
# bf16x3 mode (MTK_PREC_BF16X3): split-bf16 operands (hi + lo), three bf16 MFMAs per product, fp32 accumulate, layer by
# layer.  This is the fast mode that carries the fp32 bars: denoiser, losses and scores <= 1e-4 relative, score ranking
# identical to the reference's on the fixtures.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", CASES)
def test_mlp_forward_bf16x3(tag):
    c = load_case(tag)
    d = build(c, precision="bf16x3")
    x, t = tt(c["ml_model_in"]).to(dev()), tt(c["ml_t"]).to(dev())
    with torch.no_grad():
        out = d._denoise_fn(x, t)
    want = tt(c["ml_model_out"])
    scale = float(want.abs().max())
    close(out, want, rtol=RTOL, atol=2e-5 * scale)
    B, D = x.shape
    eng = d.engine()
    o2 = torch.empty((D, B), device=dev())
    eng.mlp_forward(x.t().contiguous(), t, out=o2, precision="bf16x3", x_strides=(1, B), out_strides=(1, B), B=B)
    assert torch.equal(o2.t().contiguous(), out)
    for n in (1, 31, 129):
        o = eng.mlp_forward(x[:n].contiguous(), t[:n].contiguous(), precision="bf16x3")
        assert torch.equal(o, out[:n]), n


@pytest.mark.parametrize("shape", [(120, [1024] * 4), (142, [512, 1024, 1024, 1024, 1024, 512]), (40, [128, 256]), (200, [1536, 768, 2048])])
def test_mlp_forward_bf16x3_shapes(shape):
    """The shapes of BASELINE configs 1, 4 and 5 and two odd ones, 70,001 rows (several tiles per workgroup, a second pass
    through the layer chain): a sample of rows against the oracle at the fp32 bar; windows equal separate small calls."""
    import midst_toolkit_amd as mtk

    D, layers = shape
    params = orc.init_mlp_params(D, layers, seed=3)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0)).to(dev())
    net.load_state_dict(params, strict=True)
    net.precision = "bf16x3"
    g = torch.Generator().manual_seed(2)
    B = 70_001
    x = torch.randn((B, D), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    xd, td = x.to(dev()), t.to(dev())
    with torch.no_grad():
        out = net(xd, td)
    pick = torch.randperm(B, generator=g)[:1500]
    want = orc.mlp_forward(params, x[pick], t[pick])
    scale = float(want.abs().max())
    err = float((out.cpu()[pick] - want).abs().max())
    assert err <= 1e-4 * scale, (err / scale)
    eng = net.engine(dev())
    for lo in (0, 256 * 97 + 13, 65_536 - 150, B - 300):
        o = eng.mlp_forward(xd[lo:lo + 300].contiguous(), td[lo:lo + 300].contiguous(), precision="bf16x3")
        assert torch.equal(o, out[lo:lo + 300]), lo


@pytest.mark.parametrize("tag", ["A_numeric", "N_tf"])
def test_tf_score_bf16x3(tag):
    from midst_toolkit_amd import scoring

    c = load_case(tag)
    d = build(c, precision="bf16x3")
    x0 = tt(c["x0"]).to(dev())
    noise = tt(c["tf_noise"])
    P = noise.shape[0]
    for k in range(4):
        t, add = (int(v) for v in c[f"tf{k}_t"])
        got = scoring.mixed_loss(d, x0, {"y": None}, noise.tolist(), P, add, t)
        want = tt(c[f"tf{k}_score"])
        close(got, want, rtol=RTOL)
        if add == 0:  # ranking: the reference's, except inside ties of the reference's own scores (4e-5: ten times the fp32 path's)
            check_ranking(got.cpu(), want, tie=4e-5)


@pytest.mark.parametrize("tag", ["A_numeric", "B_ragged", "M_mixed", "L_catonly"])
def test_mixed_loss_bf16x3(tag):
    """Whole mixed_loss in the split-bf16 mode against the reference's per-record values: Gaussian rows <= 1e-4; the
    multinomial rows carry the conditioning floor of tests/test_conditioning.py, per record as in the fp32 test and
    averaged for the batch mean.  (Why the mean needs it too: on the 96-record B_ragged fixture the REFERENCE's own batch
    mean moves by up to 4.2e-4 relative when its denoiser output is perturbed by 3e-7 relative - less than fp32 GEMM
    summation-order noise - because single records move by percents; measured with the oracle, 20 draws.)"""
    c = load_case(tag)
    d = build(c, precision="bf16x3")
    x0 = tt(c["x0"]).to(dev())
    t = tt(c["ml_t"]).to(dev())
    eps = tt(c["ml_eps"]).to(dev()) if c["Nn"] else None
    u = tt(c["ml_u"]).to(dev()) if c["C"] else None
    lm, lg = d.mixed_loss_rows(x0, timestep=t, noise=eps, uniform=u)
    if c["Nn"]:
        close(lg, tt(c["ml_lg_rows"]), rtol=RTOL)
    if c["C"]:
        want = tt(c["ml_lm_rows"])
        s = orc.make_schedule(c["T"], c["sched"])
        floor = orc.multinomial_loss_floor(s, c["K"], c["Nn"], tt(c["ml_model_out"]), tt(c["x0"]), tt(c["ml_xt_idx"]), tt(c["ml_t"]))
        err = (lm.cpu().double() - want.double()).abs()
        assert bool((err <= RTOL * want.abs().double() + 2 * floor).all()), float((err / (RTOL * want.abs().double() + 2 * floor)).max())
        assert abs(float(lm.mean().cpu()) - float(want.mean())) <= RTOL * abs(float(want.mean())) + 2 * float(floor.mean())


def bf16_emulated_forward(params, x, t):
    """fp32 math on bf16-rounded weights and activations (what the fused kernel computes, up to
    summation order)."""
    import torch.nn.functional as F

    r = lambda v: v.to(torch.bfloat16).float()  # noqa: E731
    emb = orc.time_table(params, int(t.max()) + 1)[t]
    h = F.linear(r(x), r(params["proj.weight"]), params["proj.bias"]) + emb
    i = 0
    while f"mlp.blocks.{i}.linear.weight" in params:
        h = F.relu(F.linear(r(h), r(params[f"mlp.blocks.{i}.linear.weight"]), params[f"mlp.blocks.{i}.linear.bias"]))
        i += 1
    return F.linear(r(h), r(params["mlp.head.weight"]), params["mlp.head.bias"])


def test_tf_score_bf16():
    from midst_toolkit_amd import scoring

    c = load_case("N_tf")
    d = build(c, precision="bf16")
    x0 = tt(c["x0"]).to(dev())
    noise = tt(c["tf_noise"]).to(dev())
    ts = [int(c[f"tf{k}_t"][0]) for k in (0, 1, 3)]
    many = scoring.score_records(d, x0, noise, ts, 0, precision="bf16")
    for j, k in enumerate((0, 1, 3)):
        want = tt(c[f"tf{k}_score"])
        rel = float(((many[j].cpu() - want).abs() / want.abs()).max())
        assert rel <= BF16_TOL, (k, rel)
    t2, add2 = (int(v) for v in c["tf2_t"])
    got = scoring.score_records(d, x0, noise, [t2], add2, precision="bf16")[0]
    assert float(((got.cpu() - tt(c["tf2_score"])).abs() / tt(c["tf2_score"]).abs()).max()) <= BF16_TOL
    # a batch that is not a multiple of the tile, many timesteps (> one launch group)
    ts = list(range(0, 1000, 13))
    big = scoring.score_records(d, x0[:37], noise, ts, 0, precision="bf16")
    ref = scoring.score_records(d, x0[:37], noise, ts, 0, precision="fp32")
    assert big.shape == (len(ts), 37, noise.shape[0])
    assert float(((big - ref).abs() / ref.abs()).max()) <= BF16_TOL


def _trained_shape_n(steps=600, batch=4096, n_members=2_000, n_total=100_000):
    """A briefly trained 4x512 / 20-feature denoiser (BASELINE configs[1]'s shape) on 2,000 draws of a two-component Gaussian
    mixture (~1,200 epochs, enough to start memorising them), and 100k records to score: the 2,000 training members first,
    then fresh draws of the same mixture."""
    import midst_toolkit_amd as mtk

    Nn, T = 20, 1000
    g = torch.Generator().manual_seed(20250821)

    def draw(n):
        comp = (torch.rand((n, 1), generator=g) < 0.35).float()
        return (torch.randn((n, Nn), generator=g) * (0.4 + 0.5 * comp) + (2.0 * comp - 0.7)).clamp(-5.2, 5.2)

    members, others = draw(n_members), draw(n_total - n_members)
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([512] * 4, 0.0))
    diff = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev()).to(dev())
    tr = mtk.ClavaDDPMTrainer(diff, iter(()), 1e-3, 0.0, steps, device=dev())
    xm = members.to(dev())
    for step in range(steps):
        idx = torch.randint(0, n_members, (batch,), generator=g).to(dev())
        tr._train_step(xm[idx], None)
        tr._anneal_learning_rate(step)
    tr.sync_modules()
    tr.finish()
    return diff, torch.cat([members, others]).to(dev()), torch.randn((4, Nn), generator=g).to(dev()), n_members


def test_bf16_ranking_fidelity():
    """What the bf16 tolerance means for the attack (north_star: membership-score rankings).  The headline
    precision does not reproduce the fp32 ranking position by position - no 8-bit-mantissa product can - so the stated
    tolerance is on what the attack consumes: rank correlation with the exact path, the top-decile membership set, and the
    guarantee that two records only come out swapped when their fp32 scores are closer than the two score errors combined.
    Measured on (a) the reference's own scores of the N_tf fixture, (b) 100k records under a random-init 4x512 denoiser and
    (c) 100k records (half of them training members) under a briefly trained one; DESIGN section 4 quotes the figures
    printed here.  The bf16x3 mode on the same inputs must give the fp32 ranking except inside documented ties."""
    import midst_toolkit_amd as mtk
    from midst_toolkit_amd import scoring

    def bars(rep, rel, kendall, spearman, top, rel_bar=BF16_TOL):
        assert rel <= rel_bar, rel
        assert rep["worst_swapped_gap"] <= 2.2 * rel, rep  # a swap needs |want_i - want_j| <= err_i + err_j
        assert rep["kendall"] >= kendall and rep["spearman"] >= spearman and rep["top10_overlap"] >= top, rep

    c = load_case("N_tf")
    d = build(c, precision="bf16")
    x0, noise = tt(c["x0"]).to(dev()), tt(c["tf_noise"]).to(dev())
    for k in (0, 1, 3):
        t = int(c[f"tf{k}_t"][0])
        got, want = scoring.score_records(d, x0, noise, [t], 0, precision="bf16")[0], tt(c[f"tf{k}_score"])
        rep, rel = ranking_report(got, want), float(((got.cpu() - want).abs() / want.abs()).max())
        print(f"bf16 ranking, N_tf fixture vs reference, t={t}: max rel err {rel:.2e}", rep)
        bars(rep, rel, 0.97, 0.995, 0.9)

    Nn, T, B, P = 20, 1000, 100_000, 4
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([512] * 4, 0.0))
    rnd = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev()).to(dev())
    g = torch.Generator().manual_seed(20250821)
    x, nz = torch.randn((B, Nn), generator=g).clamp(-5.2, 5.2).to(dev()), torch.randn((P, Nn), generator=g).to(dev())
    trained, xt, nzt, n_mem = _trained_shape_n()
    ts = [5, 50, 200, 500, 900]
    # a trained denoiser's scores are small residuals (eps - prediction)^2, so the same bf16 error of the prediction is a larger
    # fraction of them (measured: up to 1.2e-1 of a score at t = 900, where the residual is smallest): no per-score bar is
    # claimed for bf16 there, only the ranking bars and the member / non-member separation; bf16x3 keeps <= 1e-4 per score
    for name, diff, rec, eps, (kendall, spearman, top, rel_bar) in (("random-init", rnd, x, nz, (0.97, 0.998, 0.95, BF16_TOL)),
                                                                     ("trained", trained, xt, nzt, (0.96, 0.995, 0.95, float("inf")))):
        ref = scoring.score_records(diff, rec, eps, ts, 0, precision="fp32")
        got = scoring.score_records(diff, rec, eps, ts, 0, precision="bf16")
        x3 = scoring.score_records(diff, rec, eps, ts, 0, precision="bf16x3")
        for j, t in enumerate(ts):
            rep, rel = ranking_report(got[j], ref[j]), float(((got[j] - ref[j]).abs() / ref[j].abs()).max())
            print(f"bf16 ranking, 100k records, {name} 4x512, t={t}: median score {float(ref[j].median()):.3e}, max rel err {rel:.2e}", rep)
            bars(rep, rel, kendall, spearman, top, rel_bar)
            rep3, rel3 = ranking_report(x3[j], ref[j]), float(((x3[j] - ref[j]).abs() / ref[j].abs()).max())
            print(f"bf16x3 ranking, 100k records, {name} 4x512, t={t}: max rel err {rel3:.2e}", rep3)
            # trained: a score is a squared residual ~1e-2 of the eps scale, so fp32 summation-order noise of the prediction
            # (~5e-7, present between any two fp32 implementations) is already up to ~2e-4 of the smallest scores
            assert rel3 <= (RTOL if name == "random-init" else 1e-3) and rep3["worst_swapped_gap"] <= 2.2 * rel3 and rep3["kendall"] >= 0.9999, (rel3, rep3)
        if name == "trained":  # the attack's signal: members score lower than non-members; bf16 must keep that separation
            im = torch.randint(0, n_mem, (400_000,), generator=g).to(dev())
            io = torch.randint(n_mem, rec.shape[0], (400_000,), generator=g).to(dev())
            for j, t in enumerate(ts):
                auc = [float((s_.mean(1)[im] < s_.mean(1)[io]).float().mean()) for s_ in (ref[j], got[j])]
                print(f"member-vs-other pair AUC at t={t}: fp32 {auc[0]:.4f}, bf16 {auc[1]:.4f}")
                assert abs(auc[0] - auc[1]) <= 2e-3, auc


def test_mixed_loss_bf16_mean():
    c = load_case("M_mixed")
    d = build(c, precision="bf16")
    x0, t = tt(c["x0"]).to(dev()), tt(c["ml_t"]).to(dev())
    lm, lg = d.mixed_loss_rows(x0, timestep=t, noise=tt(c["ml_eps"]).to(dev()), uniform=tt(c["ml_u"]).to(dev()))
    wm, wg = tt(c["ml_lm_rows"]), tt(c["ml_lg_rows"])
    assert abs(float(lg.mean().cpu()) - float(wg.mean())) <= BF16_TOL * float(wg.mean())
    assert abs(float(lm.mean().cpu()) - float(wm.mean())) <= 5 * BF16_TOL * float(wm.mean())


# ------------------------------------------------------------------------------------------------
# MTK_MATH_FAST (hardware exp/log, exact block log-sum-exp): same formulas, checked against the
# oracle on injected noise.  Batch means <= 1e-4; per record within the conditioning floor; categorical
# draws equal except near-ties (margin 2e-4 of the hardware log) rarer than 1e-3.
# ------------------------------------------------------------------------------------------------
def _big_mixed_batch(c, B, seed=11):
    g = torch.Generator().manual_seed(seed)
    K, Nn, S = c["K"], c["Nn"], c["S"]
    x0 = torch.cat([torch.randn((B, Nn), generator=g).clamp(-5.2, 5.2)] + [torch.randint(0, int(k), (B, 1), generator=g).float() for k in K], 1)
    t = torch.randint(0, c["T"], (B,), generator=g)
    t[:3] = torch.tensor([0, 1, c["T"] - 1])
    return x0, t, torch.randn((B, Nn), generator=g), torch.rand((B, S), generator=g)


@pytest.mark.parametrize("tag", ["M_mixed", "B_ragged", "K1_single", "P_x0direct"])
def test_fast_math_against_oracle(tag, monkeypatch):
    global TIE_MARGIN
    c = load_case(tag)
    B, Nn, C, D = 4096, c["Nn"], c["C"], c["D"]
    d = build(c)
    d.math = "fast"
    eng = d.engine()
    s = orc.make_schedule(c["T"], c["sched"])
    x0, t, eps, u = _big_mixed_batch(c, B)
    want_xt, want_idx = orc.q_sample_fused(s, c["K"], Nn, x0, t, eps, u)
    x_in = torch.empty((D, B), device=dev())
    idx = torch.empty((C, B), dtype=torch.int32, device=dev())
    eng.q_sample(x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()), x_in, (1, B), idx)
    close(x_in[:Nn].t(), want_xt, rtol=1e-6)
    logits = orc.q_pred(s, orc.log_onehot(x0[:, Nn:].long(), c["K"]), t, orc.log_k_expanded(c["K"]))
    monkeypatch.setattr(sys.modules[__name__], "TIE_MARGIN", 2e-4)
    check_categorical(idx.t(), want_idx, logits, u, c["K"])
    # loss with the oracle's denoiser output and categories injected
    out = orc.mlp_forward(case_params(c), orc.model_input(want_xt, want_idx, c["K"]), t)
    fl = oracle_flags(c)
    wm, wg = orc.loss_rows(s, c["K"], Nn, out, x0, want_idx, eps, t, direct=fl.get("direct", False))
    lm, lg = torch.empty(B, device=dev()), torch.empty(B, device=dev())
    eng.loss_rows(colmajor(out), (1, B), x0.to(dev()), want_idx.t().contiguous().int().to(dev()), eps.to(dev()), t.to(dev()), lm, lg)
    close(lg, wg, rtol=1e-5)
    assert abs(float(lm.mean().cpu()) - float(wm.mean())) <= RTOL * abs(float(wm.mean()))
    floor = orc.multinomial_loss_floor(s, c["K"], Nn, out, x0, want_idx, t)
    err = (lm.cpu().double() - wm.double()).abs()
    bar = RTOL * wm.abs().double() + 3 * floor + 1e-5  # 3x: the reference and the fast path each sit within ~1.5 floor of the exact value
    assert bool((err <= bar).all()), float((err / bar).max())
    # reverse step, t in the middle and t = 0
    for tt_ in (c["T"] // 2, 0):
        g = torch.Generator().manual_seed(tt_ + 5)
        mo = torch.randn((B, D), generator=g) * 2
        z = torch.randn((B, Nn), generator=g, dtype=torch.float64)
        zi = torch.stack([torch.randint(0, int(k), (B,), generator=g) for k in c["K"]], 1)
        n = torch.randn((B, Nn), generator=g, dtype=torch.float64)
        uu = torch.rand((B, c["S"]), generator=g)
        want_z = orc.gaussian_p_step(s, mo[:, :Nn], z, tt_, n, x0_param=fl.get("x0_param", False))
        tv = torch.full((B,), tt_, dtype=torch.long)
        logp = orc.q_posterior(s, orc.block_log_softmax(mo[:, Nn:], c["K"]), orc.log_onehot(zi, c["K"]), tv, c["K"])
        if fl.get("direct"):
            logp = orc.block_log_softmax(mo[:, Nn:], c["K"])
        want_zi = orc.gumbel_argmax(logp, uu, c["K"])
        zd = z.t().contiguous().to(dev())
        zid = zi.t().contiguous().int().to(dev())
        eng.p_sample_step(B, tt_, False, colmajor(mo), (1, B), zd, zid, n.to(dev()), uu.to(dev()), None, (1, B))
        close(zd.t(), want_z, rtol=1e-12, atol=1e-12)
        check_categorical(zid.t(), want_zi, logp, uu, c["K"])


@pytest.mark.parametrize("tag", ["A_numeric", "B_ragged", "M_mixed", "P_x0direct"])
def test_gaussian_kl_loss(tag):
    """GaussianLossType.KL through the C ABI (mtk_gauss_kl_rows) against the reference's own per-record values
    (tests/golden/kl_*.npz; the oracle equals them bit for bit): fp64 arithmetic on both sides, 1e-4 relative is the
    north-star bar, measured ~1e-9.  Then the drop-in mixed_loss with the loss type set, the reference's draws injected."""
    import os

    import midst_toolkit_amd as mtk

    c = load_case(tag)
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", f"kl_{tag}.npz"))
    Nn, B = c["Nn"], c["x0"].shape[0]
    d = build(c)
    d.gaussian_loss_type = mtk.GaussianLossType.KL
    x0, mo, xt, t = tt(c["x0"])[:, :Nn], tt(c["ml_model_out"])[:, :Nn], tt(c["ml_x_num_t"]), tt(c["ml_t"])
    for name, tv in (("t", t), ("t0", torch.zeros_like(t)), ("t1", torch.ones_like(t))):
        got = d._gaussian_loss(mo.to(dev()), x0.to(dev()), xt.to(dev()), tv.to(dev()), torch.zeros_like(mo).to(dev()))
        assert got.dtype == torch.float64 and got.shape == (B,)
        close(got, torch.from_numpy(k[f"kl_rows_{name}"]), rtol=1e-7, atol=1e-9)
    # whole path: q_sample -> denoiser (fp32 MFMA) -> KL rows, same timesteps and noise as the reference's call
    uni = tt(c["ml_u"]).to(dev()) if c["C"] else None  # categorical draws do not enter the Gaussian term's inputs' numeric block
    lm, lg = d.mixed_loss_rows(tt(c["x0"]).to(dev()), timestep=torch.from_numpy(k["call_t"]).to(dev()),
                               noise=torch.from_numpy(k["call_eps"]).to(dev()), uniform=uni)
    assert lg.dtype == torch.float64
    if c["C"] == 0:  # with categorical columns the denoiser input depends on the reference's uniforms of that call, which the fixture does not hold
        want = float(k["call_means"][1])
        assert abs(float(lg.mean().cpu()) - want) <= RTOL * abs(want)
    mtk.ClavaDDPMTrainer(d, None, 1e-3, 0.0, 1, device=dev()).finish()  # the trainer takes a KL-configured model (gradients: test_gpu_train.py)
    # strided (column-major) operands through the engine: identical values
    eng = d._aux_engine(Nn)
    out2 = torch.empty((B,), dtype=torch.float64, device=dev())
    eng.gauss_kl_rows(colmajor(mo), (1, B), x0.contiguous().to(dev()), colmajor(xt), (1, B), t.to(dev()), out2)
    assert torch.equal(out2, d._gaussian_loss(mo.to(dev()), x0.to(dev()), xt.to(dev()), t.to(dev()), torch.zeros_like(mo).to(dev())))


def test_fast_math_wide_blocks(monkeypatch):
    """Blocks wider than the small register cache (12): 14 and 16 levels take the 16-column cached kernels, 20 levels the
    streaming pass.  Same checks as above with a random denoiser output, plus the in-kernel-noise draws stay in range."""
    import midst_toolkit_amd as mtk

    K, Nn, T, B = [14, 3, 16, 20, 5], 4, 1000, 4096
    c = {"K": K, "Nn": Nn, "S": sum(K), "T": T}
    C, S, D = len(K), sum(K), Nn + sum(K)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([64, 64], 0.0))
    d = mtk.GaussianMultinomialDiffusion(np.array(K), Nn, net, num_timesteps=T, device=dev()).to(dev())
    d.math = "fast"
    eng = d.engine()
    s = orc.make_schedule(T, "cosine")
    x0, t, eps, u = _big_mixed_batch(c, B, seed=21)
    want_xt, want_idx = orc.q_sample_fused(s, K, Nn, x0, t, eps, u)
    x_in = torch.empty((D, B), device=dev())
    idx = torch.empty((C, B), dtype=torch.int32, device=dev())
    eng.q_sample(x0.to(dev()), t.to(dev()), eps.to(dev()), u.to(dev()), x_in, (1, B), idx)
    logits = orc.q_pred(s, orc.log_onehot(x0[:, Nn:].long(), K), t, orc.log_k_expanded(K))
    monkeypatch.setattr(sys.modules[__name__], "TIE_MARGIN", 2e-4)
    check_categorical(idx.t(), want_idx, logits, u, K)
    g = torch.Generator().manual_seed(4)
    out = torch.randn((B, D), generator=g) * 2
    wm, wg = orc.loss_rows(s, K, Nn, out, x0, want_idx, eps, t)
    lm, lg = torch.empty(B, device=dev()), torch.empty(B, device=dev())
    eng.loss_rows(colmajor(out), (1, B), x0.to(dev()), want_idx.t().contiguous().int().to(dev()), eps.to(dev()), t.to(dev()), lm, lg)
    close(lg, wg, rtol=1e-5)
    assert abs(float(lm.mean().cpu()) - float(wm.mean())) <= RTOL * abs(float(wm.mean()))
    floor = orc.multinomial_loss_floor(s, K, Nn, out, x0, want_idx, t)
    err = (lm.cpu().double() - wm.double()).abs()
    bar = RTOL * wm.abs().double() + 3 * floor + 1e-5
    assert bool((err <= bar).all()), float((err / bar).max())
    for tt_ in (T // 2, 0):
        g = torch.Generator().manual_seed(tt_ + 6)
        mo = torch.randn((B, D), generator=g) * 2
        z = torch.randn((B, Nn), generator=g, dtype=torch.float64)
        zi = torch.stack([torch.randint(0, int(k), (B,), generator=g) for k in K], 1)
        n = torch.randn((B, Nn), generator=g, dtype=torch.float64)
        uu = torch.rand((B, S), generator=g)
        tv = torch.full((B,), tt_, dtype=torch.long)
        logp = orc.q_posterior(s, orc.block_log_softmax(mo[:, Nn:], K), orc.log_onehot(zi, K), tv, K)
        want_zi = orc.gumbel_argmax(logp, uu, K)
        zd = z.t().contiguous().to(dev())
        zid = zi.t().contiguous().int().to(dev())
        eng.p_sample_step(B, tt_, False, colmajor(mo), (1, B), zd, zid, n.to(dev()), uu.to(dev()), None, (1, B))
        close(zd.t(), orc.gaussian_p_step(s, mo[:, :Nn], z, tt_, n), rtol=1e-12, atol=1e-12)
        check_categorical(zid.t(), want_zi, logp, uu, K)
        # in-kernel noise (inverse-CDF draws from the register-cached / streamed blocks): the empirical distribution of
        # the widest cached block (16 levels) and the streamed one (20) follows the oracle's posterior
        zid2 = zi.t().contiguous().int().to(dev())
        zd2 = z.t().contiguous().to(dev())
        eng.p_sample_step(B, tt_, False, colmajor(mo), (1, B), zd2, zid2, None, None, None, (1, B), seed=77)
        got = zid2.t().cpu().long()
        off = orc.offsets_of(K)
        for k in (2, 3):
            assert int(got[:, k].min()) >= 0 and int(got[:, k].max()) < K[k]
            p = torch.exp(logp[:, off[k]:off[k + 1]].double())
            p = p / p.sum(1, keepdim=True)
            expect = p.sum(0)                                   # expected count per level
            count = torch.bincount(got[:, k], minlength=K[k]).double()
            var = (p * (1 - p)).sum(0)
            assert bool(((count - expect).abs() <= 5 * var.sqrt() + 1).all()), (k, count.tolist(), expect.tolist())


def test_fast_math_staged_rows_equal_direct_rows():
    """The fast kernels copy a wave's 64 row-major records into LDS when x0 is 16-byte aligned and narrow, and read
    them through strides otherwise: same arithmetic, so outputs are identical — full waves and a ragged last wave."""
    c = load_case("M_mixed")
    B, Nn, C, D = 64 * 15 + 40, c["Nn"], c["C"], c["D"]
    d = build(c)
    d.math = "fast"
    eng = d.engine()
    x0, t, eps, u = _big_mixed_batch(c, B, seed=5)
    ld = x0.shape[1]
    aligned = x0.to(dev())
    buf = torch.empty(B * ld + 1, device=dev())
    direct = buf[1:].view(B, ld)
    direct.copy_(aligned)
    assert aligned.data_ptr() % 16 == 0 and direct.data_ptr() % 16 != 0
    g = torch.Generator().manual_seed(9)
    out = colmajor(torch.randn((B, D), generator=g) * 2)
    res = []
    for x in (aligned, direct):
        x_in = torch.empty((D, B), device=dev())
        idx = torch.empty((C, B), dtype=torch.int32, device=dev())
        eng.q_sample(x, t.to(dev()), eps.to(dev()), u.to(dev()), x_in, (1, B), idx)
        x_in2 = torch.empty((D, B), device=dev())
        idx2 = torch.empty((C, B), dtype=torch.int32, device=dev())
        eng.q_sample(x, t.to(dev()), None, None, x_in2, (1, B), idx2, seed=31)
        lm, lg = torch.empty(B, device=dev()), torch.empty(B, device=dev())
        eng.loss_rows(out, (1, B), x, idx, eps.to(dev()), t.to(dev()), lm, lg)
        res.append((x_in, idx, x_in2, idx2, lm, lg))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_fast_math_philox_statistics():
    """In-kernel noise + fast math: the loss of a fixed batch agrees with the injected-noise reference
    value in distribution (different random stream, so only the mean over many records is compared)."""
    c = load_case("M_mixed")
    B = 16384
    d = build(c)
    x0, t, eps, u = _big_mixed_batch(c, B)
    lm_ref, lg_ref = d.mixed_loss_rows(x0.to(dev()), timestep=t.to(dev()), noise=eps.to(dev()), uniform=u.to(dev()))
    d.rng = "philox"
    lm, lg = d.mixed_loss_rows(x0.to(dev()), timestep=t.to(dev()), seed=123)
    se_g = float(lg_ref.std()) / B ** 0.5
    se_m = float(lm_ref.std()) / B ** 0.5
    assert abs(float(lg.mean()) - float(lg_ref.mean())) < 6 * se_g
    assert abs(float(lm.mean()) - float(lm_ref.mean())) < 6 * se_m


def test_philox_categorical_distributions():
    """In-kernel noise draws categories by inverse CDF from one Philox word per block (noising: closed form).
    Same distributions as the reference's Gumbel arg-max: empirical frequencies over 2^18 identical rows
    must sit within 5 sigma of the oracle's probabilities, for noising, the reverse step and the initial draw."""
    c = load_case("B_ragged")
    B, Nn, C, D, K = 1 << 18, c["Nn"], c["C"], c["D"], c["K"]
    d = build(c)
    d.rng = "philox"
    eng = d.engine()
    s = orc.make_schedule(c["T"], c["sched"])
    off = orc.offsets_of(K)
    g = torch.Generator().manual_seed(3)

    def check(idx, probs, what):
        for k in range(C):
            p = probs[off[k]:off[k + 1]].double()
            p = p / p.sum()
            f = torch.bincount(idx[k].cpu().long(), minlength=int(K[k])).double() / B
            sig = (p * (1 - p) / B).sqrt() + 1e-9
            assert bool(((f - p).abs() <= 5 * sig + 2e-6).all()), f"{what} block {k}: {f.tolist()} vs {p.tolist()}"

    for tval in (0, 40, 500, 999):
        row = torch.cat([torch.zeros(Nn)] + [torch.tensor([float(int(k) // 2)]) for k in K])
        x0 = row.repeat(B, 1).to(dev())
        t = torch.full((B,), tval, dtype=torch.long, device=dev())
        x_in = torch.empty((D, B), device=dev())
        idx = torch.empty((C, B), dtype=torch.int32, device=dev())
        eng.q_sample(x0, t, None, None, x_in, (1, B), idx, seed=77 + tval)
        logq = orc.q_pred(s, orc.log_onehot(row[None, Nn:].long(), K), torch.tensor([tval]), orc.log_k_expanded(K))[0]
        check(idx, logq.exp(), f"q_sample t={tval}")
        # the model input image is the log-one-hot of the drawn categories
        want_img = orc.log_onehot(idx.t().cpu().long()[:64], K)
        assert torch.equal(x_in[Nn:, :64].t().cpu(), want_img)
        # reverse step from that state with one fixed logit row
        mo = (torch.randn((1, D), generator=g) * 2).repeat(B, 1)
        zi = torch.stack([torch.full((B,), int(k) - 1) for k in K], 1)
        logp = orc.q_posterior(s, orc.block_log_softmax(mo[:1, Nn:], K), orc.log_onehot(zi[:1], K), torch.tensor([tval]), K)[0]
        z = torch.zeros((max(Nn, 1), B), dtype=torch.float64, device=dev())
        zid = zi.t().contiguous().int().to(dev())
        eng.p_sample_step(B, tval, False, colmajor(mo), (1, B), z, zid, None, None, x_in, (1, B), seed=5 + tval)
        check(zid, logp.exp(), f"p_sample_step t={tval}")
        zs = z[:Nn].cpu()
        want_mean = orc.gaussian_p_step(s, mo[:1, :Nn], torch.zeros((1, Nn), dtype=torch.float64), tval, torch.zeros((1, Nn), dtype=torch.float64))[0]
        assert bool(((zs.mean(1) - want_mean).abs() <= 6 * zs.std(1) / B ** 0.5 + 1e-6).all())
    z = torch.empty((max(Nn, 1), B), dtype=torch.float64, device=dev())
    zid = torch.empty((C, B), dtype=torch.int32, device=dev())
    eng.sample_init(B, None, None, z, zid, x_in, (1, B), seed=9)
    check(zid, torch.ones(int(np.sum(K))), "sample_init")
    assert abs(float(z[:Nn].mean())) < 6 / (B * Nn) ** 0.5 and abs(float(z[:Nn].std()) - 1) < 0.01


@pytest.mark.parametrize("tag,precision", [("M_mixed", "bf16"), ("B_ragged", "fp32"), ("A_numeric", "bf16x3")])
def test_graph_replayed_chain_equals_eager_chain(tag, precision):
    """sample() with use_graph: the T-loop (GMD:1403-1417) replayed as one captured HIP graph gives bit-for-bit the rows of
    the eager loop (same kernels, same arguments; seed and first row come from the 16-byte device pair), a captured graph
    is reused across seeds and row offsets, and mtk_ctx_reserve made the entry points capture-safe."""
    c = load_case(tag)
    d = build(c, precision=precision, T=60)
    d.rng = "philox"
    B = 300
    eager1, _ = d.sample(B, torch.ones(1), seed=11, row0=0)
    eager2, _ = d.sample(B, torch.ones(1), seed=12, row0=1000)
    d.use_graph = True
    graph1, _ = d.sample(B, torch.ones(1), seed=11, row0=0)
    assert len(d._graphs) == 1
    graph2, _ = d.sample(B, torch.ones(1), seed=12, row0=1000)   # same graph, new 16 bytes
    assert len(d._graphs) == 1
    assert torch.equal(graph1, eager1) and torch.equal(graph2, eager2)
    assert not torch.equal(graph1, graph2)
    # rows are a function of (seed, global row): a shifted window of the same stream
    shifted, _ = d.sample(B, torch.ones(1), seed=11, row0=100)
    assert torch.equal(shifted[: B - 100], graph1[100:])
    # sample_all drives it batch after batch, continuing the row range
    x, y = d.sample_all(2 * B, B, torch.ones(1), seed=11)
    assert torch.equal(x[:B], graph1) and x.shape[0] == 2 * B
    d.use_graph = False
    x2, _ = d.sample_all(2 * B, B, torch.ones(1), seed=11)
    assert torch.equal(x2, x)


@pytest.mark.parametrize("tag,precision", [("B_ragged", "fp32"), ("M_mixed", "bf16x3"), ("B_ragged", "bf16")])  # bf16 at 2x128: the layered path
def test_cached_chain_graph_survives_workspace_growth(tag, precision):
    """A captured chain holds raw addresses of the context's activation workspaces.  A later call with more rows than were
    reserved (an eager forward, or a capture at a second batch size) frees and reallocates them: the cached graph must be
    re-captured (mtk_ctx_workspace_generation), not replayed into freed memory."""
    c = load_case(tag)
    d = build(c, precision=precision, T=40)
    d.rng = "philox"
    B1, B2 = 300, 5000
    eager1, _ = d.sample(B1, torch.ones(1), seed=21, row0=0)
    eager2, _ = d.sample(B2, torch.ones(1), seed=22, row0=0)
    d2 = build(c, precision=precision, T=40)   # fresh context: workspaces start small
    d2.rng, d2.use_graph = "philox", True
    g1, _ = d2.sample(B1, torch.ones(1), seed=21, row0=0)
    gen1 = d2.engine().workspace_generation()
    assert torch.equal(g1, eager1)
    # an eager denoiser call with more rows than the capture reserved moves the workspaces
    eng = d2.engine()
    x = torch.randn((B2, eng.D), device=dev())
    eng.mlp_forward(x, torch.zeros((B2,), dtype=torch.int64, device=dev()), precision=precision)
    assert d2.engine().workspace_generation() != gen1
    # whatever reuses the freed blocks must not be touched by the replay either
    junk = [torch.full((1 << 20,), float("nan"), device=dev()) for _ in range(8)]
    g1b, _ = d2.sample(B1, torch.ones(1), seed=21, row0=0)
    assert torch.equal(g1b, eager1)
    assert all(bool(torch.isnan(j).all()) for j in junk)
    # a second cached batch size grows the reservation; the first entry is refreshed on its next use, both stay cached
    g2, _ = d2.sample(B2, torch.ones(1), seed=22, row0=0)
    g1c, _ = d2.sample(B1, torch.ones(1), seed=21, row0=0)
    g2b, _ = d2.sample(B2, torch.ones(1), seed=22, row0=0)
    assert torch.equal(g2, eager2) and torch.equal(g1c, eager1) and torch.equal(g2b, eager2)
    assert len(d2._graphs) == 2
    gen = d2.engine().workspace_generation()
    d2.sample(B1, torch.ones(1), seed=23, row0=0)
    d2.sample(B2, torch.ones(1), seed=23, row0=0)
    assert d2.engine().workspace_generation() == gen   # steady state: no reallocation, no re-capture


@pytest.mark.parametrize("shape", ["M_4x1024", "ragged_2x256", "numeric_2x256", "catonly_2x256"])
def test_denoiser_reads_the_chain_state(shape):
    """mtk_mlp_forward_state (GMD:1406: the reference concatenates float(z) and the log-one-hot rows into a [B, D] matrix
    every reverse step): the first block of the layered bf16 denoiser builds that operand from z / z_idx itself.  Same
    rounding, same weights, same MFMA order as the route through the materialised matrix: the outputs must be identical
    bit for bit - on a batch that is not a multiple of the tile, on one that spans two passes, and over whole chains
    (eager and graph-replayed) whose step kernels no longer write any input matrix."""
    import midst_toolkit_amd as mtk

    K, Nn, layers = {"M_4x1024": ([10] * 10, 20, [1024] * 4), "ragged_2x256": ([3, 7, 10, 2, 12], 6, [256, 256]),
                     "numeric_2x256": ([0], 24, [256, 256]), "catonly_2x256": ([4, 9, 16], 0, [256, 256])}[shape]
    K = np.asarray(K)
    C = 0 if int(K[0]) == 0 else len(K)
    S = int(K.sum())
    D, T = Nn + S, 30
    torch.manual_seed(3)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(layers, 0.0))
    net.precision = "bf16"
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev()).to(dev())
    d.precision, d.rng = "bf16", "philox"
    eng = d.engine()
    assert eng.state_input_ok("bf16") and not eng.state_input_ok("fp32") and not eng.state_input_ok("bf16x3")
    g = torch.Generator().manual_seed(1)
    for B in (1000, 70000 if shape == "M_4x1024" else 3000):
        z = (torch.randn((max(Nn, 1), B), generator=g, dtype=torch.float64) * 1.7).to(dev())
        zi = torch.stack([torch.randint(0, int(k), (B,), generator=g) for k in (K if C else [1])]).to(torch.int32).to(dev())
        x = torch.full((D, B), orc.LOG_TINY if hasattr(orc, "LOG_TINY") else float(np.log(np.float32(1e-30))), dtype=torch.float32, device=dev())
        x[:Nn] = z[:Nn].float()
        off = 0
        for k in range(C):
            x[Nn + off: Nn + off + int(K[k])].scatter_(0, zi[k].long().unsqueeze(0), 0.0)
            off += int(K[k])
        want = torch.empty((D, B), device=dev())
        got = torch.full((D, B), float("nan"), device=dev())
        eng.mlp_forward(x, 17, out=want, precision="bf16", x_strides=(1, B), out_strides=(1, B), B=B)
        eng.mlp_forward_state(z, zi, 17, got, B, "bf16", out_strides=(1, B))
        assert torch.equal(got, want), float((got - want).abs().max())
    # whole chains: state route against the materialised route, eager and graph-replayed
    B = 700
    d.state_input = False
    ref, _ = d.sample(B, torch.ones(1), seed=5, row0=0)
    d.state_input = True
    got, _ = d.sample(B, torch.ones(1), seed=5, row0=0)
    assert torch.equal(got, ref) and bool(torch.isfinite(ref).all())
    d.use_graph = True
    rep, _ = d.sample(B, torch.ones(1), seed=5, row0=0)
    assert torch.equal(rep, ref)


def test_attack_feature_driver_reproduces_the_reference_loop():
    """AttackFeatureDriver against the loop it replaces: for each model, for each timestep, for each
    additional timestep, mixed_loss(...) -> x[:, count * P:(count + 1) * P] (TF:531-549).  Same kernels: identical features;
    and against the reference's own scores of the N_tf fixture at the 1e-4 bar.  Models are ingested from state dicts."""
    import midst_toolkit_amd as mtk
    from midst_toolkit_amd import scoring

    c = load_case("N_tf")
    noise = tt(c["tf_noise"])
    P = noise.shape[0]
    pairs = [tuple(int(v) for v in c[f"tf{k}_t"]) for k in range(4)]       # (t, add) of the fixture's four calls
    timesteps = sorted({t for t, a in pairs})
    adds = sorted({a for t, a in pairs})
    drv = mtk.AttackFeatureDriver(noise.tolist(), timesteps, adds, device=dev())
    assert drv.feature_dimension == P * len(timesteps) * len(adds)
    d0 = build(c)
    sd = {k: v.detach().cpu() for k, v in d0.state_dict().items()}
    ctor = {"num_classes": c["K"], "num_numerical_features": c["Nn"], "num_timesteps": c["T"], "scheduler_type": c["sched"]}
    drv.add_model("m0", state_dict=sd, ctor_args=ctor)                      # ingested without any reference class
    other = {k: (v * 1.01 if k.endswith("head.weight") else v) for k, v in sd.items()}
    drv.add_model("m1", state_dict=other, ctor_args=ctor)
    x0 = tt(c["x0"]).to(dev())
    feats = drv.features("m0", x0)
    assert feats.shape == (x0.shape[0], drv.feature_dimension) and feats.is_cuda
    count = 0
    for t in timesteps:                                                     # the reference's nest
        for a in adds:
            if t + a < c["T"]:
                want = scoring.mixed_loss(d0, x0, {"y": None}, noise.tolist(), P, a, t)
                assert torch.equal(feats[:, count * P:(count + 1) * P], want), (t, a)
            count += 1
    for k, (t, a) in enumerate(pairs):                                      # and the reference's own values
        col = (timesteps.index(t) * len(adds) + adds.index(a)) * P
        close(feats[:, col:col + P], tt(c[f"tf{k}_score"]), rtol=RTOL)
    labels = torch.cat([torch.zeros(x0.shape[0] // 2), torch.ones(x0.shape[0] - x0.shape[0] // 2)])
    x, y = drv.matrix([("m0", x0, labels), ("m1", x0, labels)])
    assert x.shape == (2 * x0.shape[0], drv.feature_dimension) and y.shape == (2 * x0.shape[0],)
    assert torch.equal(x[: x0.shape[0]], feats) and not torch.equal(x[x0.shape[0]:], feats)
    # the matrix feeds a classifier of the reference's shape (classification.py: 3-layer tanh MLP) as is; challenge scoring
    clf = torch.nn.Sequential(torch.nn.Linear(drv.feature_dimension, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1), torch.nn.Sigmoid())
    pred = drv.challenge_predictions("m1", x0, clf)
    assert pred.shape == (x0.shape[0], 1) and float(pred.min()) == 0.0 and float(pred.max()) == 1.0


def test_classifier_guidance_on_hip():
    """the reference's guidance classifier (MODEL:45-120, eval mode) and its conditioning function
    (SYNTH:429-451) through mtk_clf_guidance, against the reference's own outputs (tests/golden/classifier_A_numeric.npz):
    logits and gradients <= 1e-4 of their scale; then conditional_sample (GMD:1319) driven by the HIP conditioning function
    against the reference's chain at the whole-chain bar."""
    import midst_toolkit_amd as mtk
    from tests.helpers import load_classifier

    c = load_classifier()
    clf = mtk.Classifier(c["Nn"], c["n_classes"], 128, c["hidden"], dropout_prob=0.1).to(dev())
    clf.load_state_dict(c["state"], strict=True)
    clf.train()
    with pytest.raises(NotImplementedError):  # the input gradient is the eval-mode network's (training mode: test_classifier_trains_on_hip)
        clf.guidance_gradient(torch.zeros((2, c["Nn"]), device=dev()), torch.zeros(2, dtype=torch.long, device=dev()), torch.zeros(2, dtype=torch.long, device=dev()))
    clf.eval()
    x, t, y = tt(c["x"]).to(dev()), tt(c["t"]).to(dev()), tt(c["y"]).to(dev())
    with torch.no_grad():
        logits = clf(x, t)
    want = tt(c["logits"])
    assert float((logits.cpu() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    cond = mtk.get_conditioning_function(clf, c["scale"])
    g = cond(x.double(), t, y=y)
    wg = tt(c["grad"])
    assert g.shape == wg.shape and g.dtype == torch.float32
    assert float((g.cpu() - wg).abs().max()) <= 1e-4 * float(wg.abs().max()), float((g.cpu() - wg).abs().max()) / float(wg.abs().max())
    g2 = cond(tt(c["x_rfc"]).to(dev()), t, y=y, remove_first_col=True)
    assert float((g2.cpu() - tt(c["grad_rfc"])).abs().max()) <= 1e-4 * float(wg.abs().max())
    # out-of-range timesteps poison their rows, ragged batch sizes work
    tb = t.clone()
    tb[7] = 5000
    gb = cond(x, tb, y=y)
    assert bool(gb[7].isnan().all()) and torch.equal(torch.cat([gb[:7], gb[8:]]), torch.cat([g[:7], g[8:]]))
    assert torch.equal(cond(x[:37], t[:37], y=y[:37]), g[:37])
    # the chain: conditional_sample with the reference's draws injected
    Nn, T = c["Nn"], c["T"]
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(c["d_layers"], 0.0))
    net.load_state_dict(c["params"], strict=True)
    d = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev()).to(dev())
    targets = tt(c["targets"])
    noise = [tt(n).to(dev()) for n in c["cs_noise"]]
    noise[0] = noise[0].float()
    inj = {"z0": tt(c["cs_z0"]).to(dev()), "noise": noise}
    got, outs = d.conditional_sample(targets, model_kwargs={"y": targets.to(dev())}, conditioning_function=cond, _injected=inj)
    want = tt(c["cs_sample"])
    s = orc.make_schedule(T, "cosine")
    ocond = orc.classifier_conditioning_function(c["state"], c["scale"])
    ref64 = orc.sample_chain(s, np.array([0]), Nn, c["params"], inj["z0"].cpu(), None, [n.cpu() for n in noise], [None] * T, mlp_dtype=torch.float64,
                             conditioning_function=ocond, model_kwargs={"y": targets})
    own = float((ref64 - want).abs().max())
    close(got, want, rtol=RTOL, atol=max(1e-5, 3 * own))
    plain, _ = d.sample(targets.shape[0], torch.ones(3), _injected=dict(inj, y=torch.zeros(targets.shape[0], dtype=torch.long)))
    assert float((plain - want).abs().max()) > 1e-2  # the guidance moved the chain
    # the chain runs this package's conditioning function on a side stream beside the denoiser (it needs the state, not the
    # denoiser's output): the same samples as with everything on one stream
    d.guidance_stream = False
    one_stream, _ = d.conditional_sample(targets, model_kwargs={"y": targets.to(dev())}, conditioning_function=cond, _injected=inj)
    assert torch.equal(one_stream, got)


def test_classifier_guidance_split_bf16_mode():
    """Classifier.precision = "bf16x3" (mtk_clf_set_precision; not in the reference): every GEMM of the guidance as a split-bf16
    product of its fp32 operands (hi + lo bf16 numbers, three bf16 MFMAs per product, fp32 accumulation and storage: k_gemm2x).
    Bar against the reference's own logits and input gradients: 1e-3 of their scale (measured ~1e-5; the margin is for a LeakyReLU
    whose pre-activation sits at zero and flips).  Switching back restores the fp32 bits; out-of-range timesteps still poison their
    rows; plain bf16 operands are refused (they do not carry the input gradient: DESIGN section 5)."""
    import midst_toolkit_amd as mtk
    from tests.helpers import load_classifier

    c = load_classifier()
    clf = mtk.Classifier(c["Nn"], c["n_classes"], 128, c["hidden"], dropout_prob=0.1).to(dev())
    clf.load_state_dict(c["state"], strict=True)
    clf.eval()
    x, t, y = tt(c["x"]).to(dev()), tt(c["t"]).to(dev()), tt(c["y"]).to(dev())
    cond = mtk.get_conditioning_function(clf, c["scale"])
    with torch.no_grad():
        exact_logits, exact_grad = clf(x, t), cond(x.double(), t, y=y)
        clf.precision = "bf16x3"
        logits, grad = clf(x, t), cond(x.double(), t, y=y)
    want_l, want_g = tt(c["logits"]), tt(c["grad"])
    assert not torch.equal(logits, exact_logits)  # the mode did switch
    assert float((logits.cpu() - want_l).abs().max()) <= 1e-3 * float(want_l.abs().max())
    assert float((grad.cpu() - want_g).abs().max()) <= 1e-3 * float(want_g.abs().max()), float((grad.cpu() - want_g).abs().max()) / float(want_g.abs().max())
    # a larger, ragged batch: the typical row is at the fp32 path's own level
    g = torch.Generator().manual_seed(3)
    xb = torch.randn((3001, c["Nn"]), generator=g).to(dev())
    tb = torch.randint(0, c["T"], (3001,), generator=g).to(dev())
    yb = torch.randint(0, c["n_classes"], (3001,), generator=g).to(dev())
    with torch.no_grad():
        gs = cond(xb, tb, y=yb)
        clf.precision = "fp32"
        gf = cond(xb, tb, y=yb)
    rel = ((gs - gf).norm(dim=1) / gf.norm(dim=1).clamp_min(1e-30)).cpu()
    assert float(rel.median()) <= 1e-4 and float(rel.quantile(0.99)) <= 1e-3, (float(rel.median()), float(rel.quantile(0.99)))
    clf.precision = "bf16x3"
    tb2 = t.clone()
    tb2[3] = -1
    assert bool(cond(x, tb2, y=y)[3].isnan().all())
    clf.precision = "fp32"
    with torch.no_grad():
        assert torch.equal(clf(x, t), exact_logits) and torch.equal(cond(x.double(), t, y=y), exact_grad)
    clf.precision = "bf16"
    with pytest.raises(ValueError):
        clf(x, t)


@pytest.mark.parametrize("tag,precision", [("M_mixed", "bf16"), ("B_ragged", "bf16x3"), ("B_ragged", "fp32")])
def test_weight_update_with_unchanged_shapes_allocates_nothing(tag, precision):
    """mtk_mlp_set_weights: a second call with the same layer shapes takes every block of the old images back
    from the context's pool - the allocation counter does not move (no hipMalloc; nothing freed, nothing synchronised) - and the
    denoiser then runs on the NEW weights, equal to a fresh context's output.  A later call with other shapes allocates again."""
    import midst_toolkit_amd as mtk

    c = load_case(tag)
    d = build(c, precision=precision)
    net = d.denoise_fn if hasattr(d, "denoise_fn") else d._denoise_fn
    B = 700
    x = torch.randn((B, c["D"]), device=dev())
    t = torch.randint(0, c["T"], (B,), device=dev())
    net.eval()
    ng = torch.no_grad()
    ng.__enter__()
    first = net(x, t).clone()
    eng = net._engine
    n0 = eng.alloc_count()
    for p in net.parameters():
        p.mul_(1.01)
    second = net(x, t).clone()          # sync_weights -> mtk_mlp_set_weights, same shapes
    assert eng.alloc_count() == n0, "a same-shape weight update allocated"
    assert not torch.equal(first, second)
    fresh = mtk.MLPDiffusion(c["D"], 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(c["d_layers"]), 0.0)).to(dev())
    fresh.load_state_dict(net.state_dict(), strict=True)
    fresh.precision = precision
    fresh.eval()
    assert torch.equal(fresh(x, t), second)
    for p in net.parameters():          # and again: the scratch of the call is pooled too
        p.mul_(0.99)
    net(x, t)
    assert eng.alloc_count() == n0
    ng.__exit__(None, None, None)


@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_classifier_trains_on_hip(p_drop):
    """MODEL:45-120 with the module in train() (VERDICT r2 missing 1; TRAIN:409-425, 560-627: logits = classifier(x_t, t);
    cross_entropy(logits, y).backward(); AdamW.step()).  The oracle is torch's own modules on the CPU - the reference's
    Classifier IS this stack of nn.Linear / ReLU / LeakyReLU / BatchNorm1d / Dropout - run in training mode with the library's
    dropout masks injected (torch's Bernoulli stream cannot be reproduced off-device).  Logits <= 1e-4 of the row scale; every
    gradient tensor (14 + 2 per BatchNorm) <= 3e-4 relative L2; running_mean / running_var / num_batches_tracked as torch
    updates them; identical run to run; an AdamW loop on it lowers the loss; eval mode afterwards uses the trained statistics."""
    import copy

    import torch.nn.functional as F
    from torch import nn

    import midst_toolkit_amd as mtk
    from tests.helpers import load_classifier

    c = load_classifier()
    torch.manual_seed(3)
    clf = mtk.Classifier(c["Nn"], c["n_classes"], 128, c["hidden"], dropout_prob=p_drop)
    clf.load_state_dict(c["state"], strict=True)
    ref = copy.deepcopy(clf)            # stays on the CPU: plain torch modules
    clf = clf.to(dev())
    B = 777                             # not a tile multiple
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B, c["Nn"]), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    y = torch.randint(0, c["n_classes"], (B,), generator=g)

    def ref_forward(m, masks):
        emb = m.timestep_embedding(orc.timestep_embedding(t, 128))
        h = m.proj(x) + emb
        k = 0
        for layer in m.model:
            if isinstance(layer, nn.Dropout):
                if m.training and p_drop > 0:
                    h = h * masks[k].float() / (1 - p_drop)
                k += 1
            else:
                h = layer(h)
        return h

    clf.train()
    ref.train()
    clf.dropout_seed = 99
    logits = clf(x.to(dev()), t.to(dev()))
    assert logits.requires_grad
    masks = [m.cpu() for m in clf.dropout_masks(B)] if p_drop > 0 else None
    if p_drop > 0:
        assert all(abs(float(m.float().mean()) - (1 - p_drop)) < 0.02 for m in masks)
    F.cross_entropy(logits, y.to(dev())).backward()
    want_logits = ref_forward(ref, masks)
    F.cross_entropy(want_logits, y).backward()
    scale = want_logits.detach().abs().amax(dim=1, keepdim=True).clamp_min(1e-3)
    assert float(((logits.detach().cpu() - want_logits.detach()).abs() / scale).max()) <= 1e-4
    got = {k: v.grad.detach().cpu() for k, v in clf.named_parameters()}
    for k, v in ref.named_parameters():
        if k.startswith("transformer_layer"):
            continue
        num, den = float((got[k].double() - v.grad.double()).norm()), float(v.grad.double().norm())
        assert num <= 3e-4 * den + 1e-9, f"{k}: relative L2 error {num / max(den, 1e-30):.3e}"
    for (k, b_got), (_, b_ref) in zip(clf.named_buffers(), ref.named_buffers()):
        assert torch.allclose(b_got.cpu().double(), b_ref.double(), rtol=1e-5, atol=1e-6), k
    # same seed, same call count -> same masks, same gradients
    clf2 = mtk.Classifier(c["Nn"], c["n_classes"], 128, c["hidden"], dropout_prob=p_drop)
    clf2.load_state_dict(c["state"], strict=True)
    clf2 = clf2.to(dev()).train()
    clf2.dropout_seed = 99
    F.cross_entropy(clf2(x.to(dev()), t.to(dev())), y.to(dev())).backward()
    assert all(torch.equal(v.grad.cpu(), got[k]) for k, v in clf2.named_parameters())
    # the reference's loop
    opt = torch.optim.AdamW(clf.parameters(), lr=1e-3)
    losses = []
    for _ in range(10):
        opt.zero_grad()
        loss = F.cross_entropy(clf(x.to(dev()), t.to(dev())), y.to(dev()))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    # eval mode: the trained parameters and running statistics, against torch's eval-mode modules holding the same state
    clf.eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in clf.state_dict().items()}, strict=True)
    ref.eval()
    with torch.no_grad():
        e_got = clf(x.to(dev()), t.to(dev())).cpu()
        e_want = ref_forward(ref, None)
    es = e_want.abs().amax(dim=1, keepdim=True).clamp_min(1e-3)
    assert float(((e_got - e_want).abs() / es).max()) <= 1e-4
    a = clf.train()(x.to(dev()), t.to(dev()))
    b = clf(x.to(dev()), t.to(dev()))
    with pytest.raises(RuntimeError, match="another training-mode forward"):
        a.sum().backward()
    b.sum().backward()
