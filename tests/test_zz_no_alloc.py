"""GPU: mtk_mlp_forward_noalloc (the entry point that allocates nothing: its activation ping-pong is a file-scope device
array of the library) against mtk_mlp_forward(MTK_PREC_FP32), through the C ABI.  Same kernels, so the bar is bit-exact.
Two calls in a row, the second with a smaller input: the shared array keeps the first call's contents and nothing of
them may leak into the second result.  Collected last (test_zz_*): the round's end runs `pytest -x`."""
import ctypes as C

import pytest
import torch

from oracle import diffusion_oracle as orc

pytestmark = pytest.mark.gpu


def _net(D, layers, dev):
    import midst_toolkit_amd as mtk

    params = orc.init_mlp_params(D, layers, seed=7)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0)).to(dev)
    net.load_state_dict(params, strict=True)
    return net, params


@pytest.mark.parametrize("shape", [(24, [512] * 4), (120, [1024] * 4), (37, [128, 256])])
def test_noalloc_forward_matches_allocating_forward(shape):
    from midst_toolkit_amd import _lib

    assert torch.cuda.is_available(), "these tests need the MI355X"
    dev = torch.device("cuda:0")
    D, layers = shape
    net, params = _net(D, layers, dev)
    eng = net.engine(dev)
    g = torch.Generator().manual_seed(3)
    for B in (3000, 200):  # two calls in a row on the same GPU's array, the second smaller
        x = torch.randn((B, D), generator=g).to(dev)
        t = torch.randint(0, 1000, (B,), generator=g).to(dev)
        want = eng.mlp_forward(x, t, precision="fp32")
        got = torch.full((B, D), float("nan"), device=dev)
        eng.mlp_forward_noalloc(x, t, got)
        assert torch.equal(got, want), (shape, B, float((got - want).abs().max()))
        # column-major operands and a scalar timestep through the same entry point
        got_cm = torch.full((D, B), float("nan"), device=dev)
        eng.mlp_forward_noalloc(x.t().contiguous(), 7, got_cm, x_strides=(1, B), out_strides=(1, B), B=B)
        assert torch.equal(got_cm.t().contiguous(), eng.mlp_forward(x, 7, precision="fp32"))
        if B == 200:  # and against the oracle, like the allocating path's own parity test
            ref = orc.mlp_forward(params, x.cpu(), t.cpu())
            scale = float(ref.abs().max())
            assert float((got.cpu() - ref).abs().max()) <= 1e-4 * scale
    # sizes beyond the static scratch are rejected with a clear error, nothing is launched
    big = torch.zeros((_lib.NOALLOC_MAX_ROWS + 1, D), device=dev)
    with pytest.raises(ValueError, match="exceed the static scratch"):
        eng.mlp_forward_noalloc(big, 0, torch.empty_like(big))


def test_noalloc_rejects_wide_layers():
    dev = torch.device("cuda:0")
    net, _ = _net(16, [2048, 64], dev)
    eng = net.engine(dev)
    x = torch.zeros((8, 16), device=dev)
    with pytest.raises(ValueError, match="exceeds the static scratch"):
        eng.mlp_forward_noalloc(x, 0, torch.empty_like(x))
    assert eng.mlp_forward(x, 0, precision="fp32").shape == (8, 16)  # the allocating entry point still serves it
