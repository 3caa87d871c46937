"""Inverse quantile normalisation: the oracle restatement against the outputs of the fitted
object the reference builds (tests/golden/quantile_*.npz, tools/make_golden.py gen_quantile), and the HIP kernel
against both.

Bar: the oracle is bit-exact (same scipy ndtr and numpy interp as scikit-learn runs).  The kernel evaluates Phi with
the device erfc instead of cephes ndtr, which differs by a few ulp of the UNIFORM image; the interpolation carries that
over unchanged, so the bar is 1e-9 of the column's fitted range per element (measured ~1e-13), NaN positions equal,
and values in either bound region (|x| > 5.2) exactly the end quantiles."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TAGS = ["small", "big"]


def load(tag):
    z = np.load(os.path.join(GOLD, f"quantile_{tag}.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_fitted_transformer(tag):
    z = load(tag)
    got = orc.quantile_inverse(z["x"], z["quantiles"], z["references"], bool(z["output_normal"]))
    assert np.array_equal(got, z["want"], equal_nan=True)
    assert np.isnan(z["want"]).sum() == 1  # the fixture does hold a NaN and both bound regions
    assert (z["want"][64] == z["quantiles"][-1]).all() and (z["want"][65] == z["quantiles"][0]).all()


def test_oracle_uniform_output_distribution():
    z = load("big")
    u = np.random.default_rng(3).random((512, z["quantiles"].shape[1]))
    u[0], u[1] = 0.0, 1.0
    got = orc.quantile_inverse(u, z["quantiles"], z["references"], False)
    for c in range(u.shape[1]):
        assert np.array_equal(got[:, c], np.interp(u[:, c], z["references"], z["quantiles"][:, c]))


class _Fitted:
    def __init__(self, z, dist="normal"):
        self.quantiles_, self.references_, self.output_distribution = z["quantiles"], z["references"], dist


def _check(got, want, quantiles):
    span = (quantiles[-1] - quantiles[0])[None, :]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    err = np.abs(np.nan_to_num(got) - np.nan_to_num(want))
    assert (err <= 1e-9 * span).all(), f"max err / span {float((err / span).max()):.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_gpu_inverse_matches_fitted_transformer(tag):
    import midst_toolkit_amd as mtk

    z = load(tag)
    dev = torch.device("cuda:0")
    qi = mtk.QuantileInverse(_Fitted(z), dev)
    x = torch.from_numpy(z["x"]).to(dev)
    got = qi.inverse_transform(x).cpu().numpy()
    _check(got, z["want"], z["quantiles"])
    assert np.array_equal(got[64], z["quantiles"][-1]) and np.array_equal(got[65], z["quantiles"][0])
    # strided view (the numeric block of a sampled [B, Nn + C] table) and an fp32 input
    wide = torch.zeros((x.shape[0], x.shape[1] + 3), dtype=torch.float64, device=dev)
    wide[:, : x.shape[1]] = x
    got2 = qi.inverse_transform(wide[:, : x.shape[1]]).cpu().numpy()
    assert np.array_equal(got2, got, equal_nan=True)
    x32 = x.float()
    want32 = orc.quantile_inverse(x32.double().cpu().numpy(), z["quantiles"], z["references"], True)
    _check(qi.inverse_transform(x32).cpu().numpy(), want32, z["quantiles"])


@pytest.mark.gpu
def test_gpu_inverse_uniform_empty_and_large():
    import midst_toolkit_amd as mtk

    z = load("big")
    dev = torch.device("cuda:0")
    qi = mtk.QuantileInverse(_Fitted(z, "uniform"), dev)
    u = np.random.default_rng(4).random((1000, z["quantiles"].shape[1]))
    u[0], u[1] = 0.0, 1.0
    got = qi.inverse_transform(torch.from_numpy(u).to(dev)).cpu().numpy()
    _check(got, orc.quantile_inverse(u, z["quantiles"], z["references"], False), z["quantiles"])
    assert qi.inverse_transform(torch.empty((0, u.shape[1]), dtype=torch.float64, device=dev)).shape == (0, u.shape[1])
    with pytest.raises(ValueError):
        qi.inverse_transform(torch.zeros((4, u.shape[1] + 1), device=dev))
    # full size: 1e6 rows; monotone map => sorting commutes with it, and the image stays inside the fitted range
    qn = mtk.QuantileInverse(_Fitted(z), dev)
    g = torch.Generator(device="cpu").manual_seed(9)
    big = torch.randn((1_000_000, u.shape[1]), generator=g, dtype=torch.float64).to(dev)
    out = qn.inverse_transform(big)
    a = torch.sort(out, dim=0).values
    b = qn.inverse_transform(torch.sort(big, dim=0).values)
    assert torch.equal(a, b)
    q = torch.from_numpy(z["quantiles"]).to(dev)
    assert bool((out >= q[0]).all()) and bool((out <= q[-1]).all())


@pytest.mark.gpu
def test_sampled_rows_stay_on_device_through_the_inverse_transform():
    """results_on_device: sample_all hands back device tensors (same values as the CPU return of the reference-shaped call,
    same torch draws), and QuantileInverse maps their numeric block without a host copy."""
    import midst_toolkit_amd as mtk

    z = load("small")
    dev = torch.device("cuda:0")
    n = z["quantiles"].shape[1]
    net = mtk.MLPDiffusion(n, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([64, 64], 0.0))
    d = mtk.GaussianMultinomialDiffusion(np.array([0]), n, net, num_timesteps=20, device=dev).to(dev)
    torch.manual_seed(3)
    x_cpu, y_cpu = d.sample_all(96, 48, torch.ones(1))
    d.results_on_device = True
    torch.manual_seed(3)
    x_dev, y_dev = d.sample_all(96, 48, torch.ones(1))
    assert x_dev.is_cuda and y_dev.is_cuda and not x_cpu.is_cuda
    assert torch.equal(x_dev.cpu(), x_cpu) and torch.equal(y_dev.cpu(), y_cpu)
    got = mtk.QuantileInverse(_Fitted(z), dev).inverse_transform(x_dev[:, :n]).cpu().numpy()
    _check(got, orc.quantile_inverse(x_cpu.numpy()[:, :n], z["quantiles"], z["references"], True), z["quantiles"])


# ------------------------------------------------------------------------------------------------ the other two normalisers
def _affine():
    z = np.load(os.path.join(GOLD, "affine.npz"))
    return {k: z[k] for k in z.files}


def test_oracle_affine_inverse_matches_fitted_scalers():
    z = _affine()
    assert np.array_equal(orc.affine_inverse(z["standard_x"], "standard", z["standard_scale"], z["standard_mean"]), z["standard_want"])
    assert np.array_equal(orc.affine_inverse(z["minmax_x"], "minmax", z["minmax_min"], z["minmax_scale"]), z["minmax_want"])
    with pytest.raises(ValueError):
        orc.affine_inverse(z["minmax_x"], "robust", z["minmax_min"], z["minmax_scale"])


class _Standard:
    def __init__(self, z):
        self.scale_, self.mean_, self.with_std, self.with_mean = z["standard_scale"], z["standard_mean"], True, True


class _MinMax:
    def __init__(self, z):
        self.min_, self.scale_ = z["minmax_min"], z["minmax_scale"]


@pytest.mark.gpu
def test_gpu_affine_inverse_bit_exact():
    """StandardScaler / MinMaxScaler inverse on the device: two fp64 roundings in scikit-learn's order -> bit-exact."""
    import midst_toolkit_amd as mtk

    z = _affine()
    dev = torch.device("cuda:0")
    for tag, fitted in (("standard", _Standard(z)), ("minmax", _MinMax(z))):
        inv = mtk.inverse_transform_for(fitted, dev)
        assert isinstance(inv, mtk.AffineInverse)
        got = inv.inverse_transform(torch.from_numpy(z[f"{tag}_x"]).to(dev)).cpu().numpy()
        assert np.array_equal(got, z[f"{tag}_want"]), tag
        assert inv.inverse_transform(torch.empty((0, got.shape[1]), dtype=torch.float64, device=dev)).shape == (0, got.shape[1])
    assert isinstance(mtk.inverse_transform_for(_Fitted(load("small")), dev), mtk.QuantileInverse)
    with pytest.raises(ValueError):
        mtk.AffineInverse(object(), dev)
