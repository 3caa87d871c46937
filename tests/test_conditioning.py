"""CPU: how well-posed is per-record parity of the multinomial loss?

The reference forms each block normaliser as log(exp(c_end - m) - exp(c_start - m)) + m from a
whole-row cumulative log-sum-exp (DU:250-289).  Replacing torch-CPU's fp32 exp/log (closed-source
MKL VML) by a correctly-rounded exp/log — a change of at most one ulp per call — moves a sizeable
share of the reference's own per-record losses by more than 1e-4 relative, while the batch mean
(what mixed_loss returns) stays within 1e-4.  The GPU parity tests therefore use, per record,
|gpu - ref| <= 1e-4 |ref| + 2 * multinomial_loss_floor and, per batch, 1e-4 on the mean.
"""
import torch
import torch.nn.functional as F

from oracle import diffusion_oracle as orc
from tests.helpers import case_params, load_case


def _perturbed_loss(fn):
    oe, ol, ote, ols = torch.exp, torch.log, torch.Tensor.exp, F.log_softmax
    cexp = lambda x: oe(x.double()).float() if x.dtype == torch.float32 else oe(x)  # noqa: E731
    clog = lambda x: ol(x.double()).float() if x.dtype == torch.float32 else ol(x)  # noqa: E731
    torch.exp, torch.log = cexp, clog
    torch.Tensor.exp = lambda self: cexp(self)
    F.log_softmax = lambda x, dim: ols(x.double(), dim=dim).float()
    try:
        return fn()
    finally:
        torch.exp, torch.log, torch.Tensor.exp, F.log_softmax = oe, ol, ote, ols


def test_reference_formula_is_ill_conditioned_per_record_but_not_in_the_mean():
    c = load_case("M_mixed")
    s = orc.make_schedule(c["T"], c["sched"])
    g = torch.Generator().manual_seed(1)
    B, K, Nn, S = 2048, c["K"], c["Nn"], c["S"]
    x0 = torch.cat([torch.randn((B, Nn), generator=g).clamp(-5.2, 5.2)] + [torch.randint(0, int(k), (B, 1), generator=g).float() for k in K], 1)
    t = torch.randint(0, c["T"], (B,), generator=g)
    eps, u = torch.randn((B, Nn), generator=g), torch.rand((B, S), generator=g)
    xnt, idx = orc.q_sample_fused(s, K, Nn, x0, t, eps, u)
    out = orc.mlp_forward(case_params(c), orc.model_input(xnt, idx, K), t)
    lm, _ = orc.loss_rows(s, K, Nn, out, x0, idx, eps, t)
    lm2, _ = _perturbed_loss(lambda: orc.loss_rows(s, K, Nn, out, x0, idx, eps, t))
    rel = ((lm2 - lm).abs() / lm.abs().clamp_min(1e-12)).double()
    assert float((rel > 1e-4).float().mean()) > 0.02  # per-record 1e-4 is not a well-posed bar
    assert abs(float(lm2.mean() - lm.mean())) <= 1e-4 * abs(float(lm.mean()))  # the mean is
    floor = orc.multinomial_loss_floor(s, K, Nn, out, x0, idx, t)
    assert bool(((lm2 - lm).abs().double() <= 1e-4 * lm.abs().double() + 2 * floor).all())
