"""Drop-in for TabSyn's EDM denoiser (models/tabsyn/model/modules.py:23-219, models/tabsyn/model/utils.py:20-87) on MI355X
(SURVEY section 8f rank 4): ``PositionalEmbedding``, ``MLPDiffusion``, ``Preconditioner``, ``EDMLoss``, ``Model`` with the
reference's constructors and parameter names (a reference state dict loads), and the Euler-Heun ``sample`` / ``sample_step``.

The network forward, the preconditioning and the loss arithmetic run in libmtk_hip.so (``mtk_edm_forward`` / ``mtk_edm_denoise``
/ ``mtk_edm_loss``: fp32 MFMA GEMMs, fused element-wise kernels either side); nothing goes through torch ops.  Training: the loss
the reference differentiates - ``Model.forward`` / ``EDMLoss`` (modules.py:165-219) - builds an autograd graph whose backward is
the library's hand-written chain (``mtk_edm_loss_keep`` / ``mtk_edm_backward``), so ``loss.backward(); optimizer.step()`` of the
reference's TabSyn loop runs against the drop-in; ``MLPDiffusion.forward`` / ``Preconditioner.forward`` on their own (sampling,
evaluation) are values, not graphs.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from torch import Tensor, nn

from . import _lib
from ._lib import check, ptr

SIGMA_MIN, SIGMA_MAX, rho, S_churn, S_min, S_max, S_noise = 0.002, 80, 7, 1, 0, float("inf"), 1  # utils.py:11-17


class PositionalEmbedding(nn.Module):
    def __init__(self, num_channels: int, max_positions: int = 10000, endpoint: bool = False):
        super().__init__()
        self.num_channels, self.max_positions, self.endpoint = num_channels, max_positions, endpoint

    def frequencies(self) -> Tensor:
        """modules.py:46-48, in the reference's fp32 torch ops (on the CPU: these are handed to the library as they are)."""
        freqs = torch.arange(start=0, end=self.num_channels // 2, dtype=torch.float32)
        freqs = freqs / (self.num_channels // 2 - (1 if self.endpoint else 0))
        return (1 / self.max_positions) ** freqs


class MLPDiffusion(nn.Module):
    def __init__(self, d_in: int, dim_t: int = 512):
        """modules.py:54-77."""
        super().__init__()
        self.d_in, self.dim_t = d_in, dim_t
        self.proj = nn.Linear(d_in, dim_t)
        self.mlp = nn.Sequential(nn.Linear(dim_t, dim_t * 2), nn.SiLU(), nn.Linear(dim_t * 2, dim_t * 2), nn.SiLU(),
                                 nn.Linear(dim_t * 2, dim_t), nn.SiLU(), nn.Linear(dim_t, d_in))
        self.map_noise = PositionalEmbedding(num_channels=dim_t)
        self.time_embed = nn.Sequential(nn.Linear(dim_t, dim_t), nn.SiLU(), nn.Linear(dim_t, dim_t))
        self._handle = None
        self._handle_key = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_handle"], state["_handle_key"] = None, None
        return state

    def __del__(self):
        try:
            h = self.__dict__.get("_handle")
            self.__dict__["_handle"] = None
            if h:
                _lib.lib().mtk_edm_destroy(h)
        except Exception:
            pass

    def _layers(self):
        return [self.proj, self.time_embed[0], self.time_embed[2], self.mlp[0], self.mlp[2], self.mlp[4], self.mlp[6]]

    def _engine(self, device: torch.device, differentiable: bool = False):
        if not differentiable and any(p.requires_grad for p in self.parameters()) and torch.is_grad_enabled():
            raise NotImplementedError("tabsyn.MLPDiffusion / Preconditioner forward do not build an autograd graph: call them under "
                                      "torch.no_grad() (sampling, evaluation); the differentiable entry point is the loss (Model / EDMLoss)")
        if device.type != "cuda":
            raise RuntimeError("tabsyn.MLPDiffusion: midst_toolkit_amd has no CPU path; move the module and inputs to the MI355X")
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._handle is not None and key == self._handle_key:
            return self._handle
        keep = [(l.weight.detach().float().contiguous().to(device), l.bias.detach().float().contiguous().to(device)) for l in self._layers()]
        Wp = (C.c_void_p * 7)(*[W.data_ptr() for W, _ in keep])
        bp = (C.c_void_p * 7)(*[b.data_ptr() for _, b in keep])
        if self._handle is not None and self._handle_key is not None and self._handle_key[0] == key[0]:
            # same device, same shapes (the module's layers are fixed at construction): an optimizer step or a load changed the
            # values - written into the handle in place (mtk_edm_set_weights), no rebuild
            with torch.cuda.device(device):
                check(_lib.lib().mtk_edm_set_weights(self._handle, Wp, bp, _lib.stream_handle(device)))
            self._handle_key = key
            return self._handle
        if self._handle is not None:
            _lib.lib().mtk_edm_destroy(self._handle)
            self._handle = None
        fr = np.ascontiguousarray(self.map_noise.frequencies().numpy().astype(np.float32))
        handle = C.c_void_p()
        index = device.index if device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(device):
            check(_lib.lib().mtk_edm_create(index, self.d_in, self.dim_t, Wp, bp, fr.ctypes.data, _lib.stream_handle(device), C.byref(handle)))
        self._handle, self._handle_key = handle, key
        return handle

    def forward(self, x: Tensor, noise_labels: Tensor, class_labels: Tensor | None = None) -> Tensor:
        """modules.py:79-95.  ``noise_labels``: [B], or one value broadcast over the batch."""
        dev = x.device
        h = self._engine(dev)
        xf = x.detach().float().contiguous()
        B = xf.shape[0]
        lab = noise_labels.detach().to(dev, torch.float32).reshape(-1).contiguous()
        if lab.numel() not in (1, B):
            raise ValueError(f"noise_labels: expected 1 or {B} values, got {lab.numel()}")
        out = torch.empty((B, self.d_in), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().mtk_edm_forward(h, B, ptr(xf), xf.stride(0), ptr(lab), int(lab.numel() == B), ptr(out), _lib.stream_handle(dev)))
        return out


class Preconditioner(nn.Module):
    def __init__(self, denoise_model: nn.Module, sigma_min: float = 0, sigma_max: float = float("inf"), sigma_data: float = 0.5):
        """modules.py:99-119."""
        super().__init__()
        self.sigma_min, self.sigma_max, self.sigma_data = sigma_min, sigma_max, sigma_data
        self.denoise_model = denoise_model

    def forward(self, x: Tensor, sigma: Tensor) -> Tensor:
        """modules.py:121-144: c_skip x + c_out F(c_in x, log(sigma) / 4), one library call."""
        net = self.denoise_model
        dev = x.device
        h = net._engine(dev)
        xf = x.detach().float().contiguous()
        B = xf.shape[0]
        sg = sigma.detach().to(dev, torch.float32).reshape(-1).contiguous()
        if sg.numel() not in (1, B):
            raise ValueError(f"sigma: expected 1 or {B} values, got {sg.numel()}")
        out = torch.empty((B, net.d_in), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().mtk_edm_denoise(h, B, ptr(xf), xf.stride(0), ptr(sg), int(sg.numel() == B), float(self.sigma_data), ptr(out),
                                             _lib.stream_handle(dev)))
        return out


class _EDMLossFunction(torch.autograd.Function):
    """EDMLoss with a graph: forward = mtk_edm_loss_keep (activations kept on the handle), backward = mtk_edm_backward into
    fresh gradient tensors for the denoiser's 14 parameters (order of MLPDiffusion._layers)."""

    @staticmethod
    def forward(ctx, net, y, rnd, n, p_mean, p_std, sigma_data, *params):
        dev = y.device
        h = net._engine(dev, differentiable=True)
        B = y.shape[0]
        loss = torch.empty((B, net.d_in), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().mtk_edm_loss_keep(h, B, ptr(y), y.stride(0), ptr(rnd), ptr(n), float(p_mean), float(p_std), float(sigma_data),
                                               ptr(loss), _lib.stream_handle(dev)))
        net._kept = getattr(net, "_kept", 0) + 1
        ctx.net, ctx.h, ctx.B, ctx.token = net, h, B, net._kept
        ctx.needs = [p.requires_grad for p in params]
        return loss

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        if ctx.token != getattr(net, "_kept", None) or ctx.h is not net._handle:
            raise RuntimeError("backward through a tabsyn loss after another differentiable loss call (or a weight change) on the same "
                               "model: the handle keeps the activations of one forward pass at a time")
        g = g.to(torch.float32).contiguous()
        grads = [(torch.empty_like(l.weight.data, dtype=torch.float32), torch.empty_like(l.bias.data, dtype=torch.float32)) for l in net._layers()]
        Wp = (C.c_void_p * 7)(*[W.data_ptr() for W, _ in grads])
        bp = (C.c_void_p * 7)(*[b.data_ptr() for _, b in grads])
        with torch.cuda.device(g.device):
            check(_lib.lib().mtk_edm_backward(ctx.h, ctx.B, ptr(g), g.stride(0), Wp, bp, _lib.stream_handle(g.device)))
        flat = [t for pair in grads for t in pair]
        return (None,) * 7 + tuple(t if need else None for t, need in zip(flat, ctx.needs))


class EDMLoss:
    def __init__(self, p_mean: float = -1.2, p_std: float = 1.2, sigma_data: float = 0.5) -> None:
        self.p_mean, self.p_std, self.sigma_data = p_mean, p_std, sigma_data

    def __call__(self, denoise_fn: nn.Module, data: Tensor, _injected: tuple[Tensor, Tensor] | None = None) -> Tensor:
        """modules.py:165-185: [B, d_in] weighted squared errors.  The two draws (``randn(B)``, ``randn_like(data)``) come from
        torch's generator in the reference's order, or from ``_injected`` (tests); noising, preconditioning, network and loss are
        one library call.  ``denoise_fn`` must be a ``Preconditioner`` over a ``tabsyn.MLPDiffusion``."""
        if not isinstance(denoise_fn, Preconditioner) or not isinstance(denoise_fn.denoise_model, MLPDiffusion):
            raise TypeError("EDMLoss on the MI355X path takes a tabsyn.Preconditioner over a tabsyn.MLPDiffusion")
        if float(denoise_fn.sigma_data) != float(self.sigma_data):
            raise ValueError("EDMLoss and its Preconditioner disagree on sigma_data")
        net = denoise_fn.denoise_model
        dev = data.device
        y = data.detach().float().contiguous()
        B = y.shape[0]
        if _injected is not None:
            rnd, n = (v.to(dev, torch.float32).contiguous() for v in _injected)
        else:
            rnd = torch.randn(B, device=dev)
            n = torch.randn_like(y)
        if torch.is_grad_enabled() and any(p.requires_grad for p in net.parameters()):
            params = [p for l in net._layers() for p in (l.weight, l.bias)]
            return _EDMLossFunction.apply(net, y, rnd, n, self.p_mean, self.p_std, self.sigma_data, *params)
        h = net._engine(dev)
        loss = torch.empty((B, net.d_in), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.lib().mtk_edm_loss(h, B, ptr(y), y.stride(0), ptr(rnd), ptr(n), float(self.p_mean), float(self.p_std), float(self.sigma_data),
                                          ptr(loss), _lib.stream_handle(dev)))
        return loss


class Model(nn.Module):
    def __init__(self, denoise_model: nn.Module, p_mean: float = -1.2, p_std: float = 1.2, sigma_data: float = 0.5):
        """modules.py:189-207."""
        super().__init__()
        self.preconditioner = Preconditioner(denoise_model, sigma_data=sigma_data)
        self.loss_fn = EDMLoss(p_mean, p_std, sigma_data)

    def forward(self, x: Tensor, _injected: tuple[Tensor, Tensor] | None = None) -> Tensor:
        """modules.py:209-219: the mean loss; with grad enabled and trainable parameters it carries a graph (module docstring)."""
        return self.loss_fn(self.preconditioner, x, _injected).mean(-1).mean()


def _karras_sigmas(net: Preconditioner, num_steps: int, device: torch.device) -> Tensor:
    """The noise levels of utils.py:37-45: num_steps points between sigma_max and sigma_min, uniform in sigma^(1/rho), then 0."""
    hi, lo = min(SIGMA_MAX, net.sigma_max) ** (1 / rho), max(SIGMA_MIN, net.sigma_min) ** (1 / rho)
    ramp = torch.arange(num_steps, dtype=torch.float32, device=device) / (num_steps - 1)
    return torch.cat([(hi + ramp * (lo - hi)) ** rho, torch.zeros(1, dtype=torch.float32, device=device)])


def sample_step(net: Preconditioner, num_steps: int, i: int, t_cur: Tensor, t_next: Tensor, x_next: Tensor, _noise: Tensor | None = None) -> Tensor:
    """utils.py:56-87, one step of the stochastic Euler-Heun sampler: churn the state up to sigma_hat, take the Euler step
    along d = (x - D(x, sigma)) / sigma, and (all but the last step) average it with the slope at the arrival point.  Both
    denoiser evaluations are library calls (mtk_edm_denoise); the state updates are a few element-wise ops on [N, dim]."""
    churn = min(S_churn / num_steps, np.sqrt(2) - 1) if S_min <= t_cur <= S_max else 0
    sigma_hat = t_cur + churn * t_cur
    eps = torch.randn_like(x_next) if _noise is None else _noise
    x_hat = x_next + (sigma_hat**2 - t_cur**2).sqrt() * S_noise * eps
    step = t_next - sigma_hat
    slope = (x_hat - net(x_hat, sigma_hat).to(torch.float32)) / sigma_hat
    x_euler = x_hat + step * slope
    if i >= num_steps - 1:
        return x_euler
    slope_next = (x_euler - net(x_euler, t_next).to(torch.float32)) / t_next
    return x_hat + step * (0.5 * slope + 0.5 * slope_next)


def sample(net: Preconditioner, num_samples: int, dim: int, num_steps: int = 50, device: torch.device | None = None,
           _injected: dict | None = None) -> Tensor:
    """utils.py:20-53.  ``_injected``: {"latents": [N, dim], "noise": [num_steps x [N, dim]]} (tests)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    latents = torch.randn([num_samples, dim], device=device) if _injected is None else _injected["latents"].to(device)
    sigmas = _karras_sigmas(net, num_steps, latents.device)
    x = latents.to(torch.float32) * sigmas[0]
    with torch.no_grad():
        for i in range(num_steps):
            x = sample_step(net, num_steps, i, sigmas[i], sigmas[i + 1], x, None if _injected is None else _injected["noise"][i].to(device))
    return x
