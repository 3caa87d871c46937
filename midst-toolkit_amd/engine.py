"""Thin object wrapper over an ``mtk_ctx`` (include/mtk.h).  Tensors in, tensors out; all compute
is enqueued on torch's current stream by libmtk_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import PREC_BF16, PREC_BF16X3, PREC_FP32, check, ptr, require_cuda

PRECISIONS = {"fp32": PREC_FP32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3}  # "bf16x3": split-bf16 (hi + lo, 3 MFMAs per product), fp32-parity grade


def _no_engine():
    """What a pickled Engine unpickles to: nothing.  The owner builds a new context lazily."""
    return None


class Engine:
    """One device context: schedule tables + categorical layout (+ packed denoiser weights)."""

    def __init__(self, device: torch.device, T: int, Nn: int, K, tables: dict[str, torch.Tensor]):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"midst_toolkit_amd needs an MI355X (ROCm 'cuda') device, got {device}")
        self.device = device
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        self.T, self.Nn = int(T), int(Nn)
        K = np.asarray(K, dtype=np.int64).reshape(-1)
        self.K = K if (len(K) and int(K[0]) != 0) else np.zeros((0,), dtype=np.int64)
        self.C = len(self.K)
        self.S = int(self.K.sum())
        self.D = self.Nn + self.S
        tab = np.stack([tables[name].double().cpu().numpy() for name in _lib.TABLE_ORDER]).astype(np.float64)
        assert tab.shape == (_lib.NTAB, self.T), tab.shape
        tab = np.ascontiguousarray(tab)
        k32 = np.ascontiguousarray(self.K.astype(np.int32))
        # log K exactly as the reference forms it: torch.log of an int64 tensor → fp32 (GMD:739)
        logk = np.ascontiguousarray(torch.log(torch.from_numpy(self.K)).numpy().astype(np.float32)) if self.C else np.zeros(1, np.float32)
        handle = C.c_void_p()
        check(_lib.lib().mtk_ctx_create(self.index, self.T, self.Nn, self.C, k32.ctypes.data if self.C else None,
                                        logk.ctypes.data if self.C else None, tab.ctypes.data, C.byref(handle)))
        self._h = handle
        self._weights_key = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().mtk_ctx_destroy(h)
            except Exception:
                pass

    def __deepcopy__(self, memo):
        """A device context is not copied with the module that holds it (deepcopy of a denoiser / diffusion
        object, TRAINER:40): the copy builds its own lazily."""
        return None

    def __reduce__(self):
        """A device context is a raw handle (ctypes pointers cannot be pickled) and means nothing in another process: the
        reference's orchestration pickles whole diffusion objects (models/clavaddpm/train.py:122-123, 298), so an Engine
        pickles as None, exactly like ``__deepcopy__``; ``GaussianMultinomialDiffusion.engine()`` / ``MLPDiffusion.engine()``
        rebuild it on first use after loading."""
        return (_no_engine, ())

    def set_math(self, mode: str) -> None:
        """"reference" (parity, default) or "fast" (hardware transcendentals, exact block log-sum-exp)."""
        check(_lib.lib().mtk_ctx_set_math(self._h, {"reference": 0, "fast": 1}[mode]))

    def set_parametrization(self, gaussian_x0: bool, multinomial_direct: bool) -> None:
        """GaussianParametrization.X0 / Parametrization.DIRECT of the reference's constructor (mtk_ctx_set_parametrization)."""
        check(_lib.lib().mtk_ctx_set_parametrization(self._h, int(bool(gaussian_x0)), int(bool(multinomial_direct))))

    def reserve(self, max_rows: int) -> None:
        """Size every workspace for calls of up to ``max_rows`` rows: afterwards no entry point allocates or synchronises
        (needed before capturing calls into a HIP graph)."""
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_ctx_reserve(self._h, int(max_rows), self._stream()))

    def workspace_generation(self) -> int:
        """Moves whenever the context frees / reallocates a buffer its kernels use (mtk_ctx_workspace_generation): a captured
        graph is only valid while this equals its value at capture time."""
        return int(_lib.lib().mtk_ctx_workspace_generation(self._h))

    def alloc_count(self) -> int:
        """hipMalloc calls made by the weight path so far (mtk_ctx_alloc_count): unchanged across a ``set_weights`` with the same
        layer shapes."""
        return int(_lib.lib().mtk_ctx_alloc_count(self._h))

    def set_seed_source(self, seed_row0: torch.Tensor | None) -> None:
        """``seed_row0``: int64 [2] device tensor {seed, row0} the reverse-chain kernels read instead of their arguments."""
        if seed_row0 is not None:
            assert seed_row0.is_cuda and seed_row0.dtype == torch.int64 and seed_row0.numel() == 2 and seed_row0.is_contiguous()
        self._seed_src = seed_row0  # keeps it alive
        check(_lib.lib().mtk_ctx_set_seed_source(self._h, ptr(seed_row0)))

    def _stream(self):
        return _lib.stream_handle(self.device)

    # ------------------------------------------------------------------ denoiser
    def set_weights(self, weights: list[tuple[torch.Tensor, torch.Tensor]], dims: list[int], freqs: torch.Tensor) -> None:
        """weights: [(W, b)] in the order proj, temb.0, temb.2, blocks..., head (fp32, on device)."""
        n_hidden = len(weights) - 4
        self._hidden_widths = [int(w.shape[0]) for w, _ in weights[3:3 + n_hidden]]
        keep = []
        for W, b in weights:
            require_cuda(W, "set_weights")
            keep += [W.detach().float().contiguous(), b.detach().float().contiguous()]
        Wp = (C.c_void_p * len(weights))(*[keep[2 * i].data_ptr() for i in range(len(weights))])
        bp = (C.c_void_p * len(weights))(*[keep[2 * i + 1].data_ptr() for i in range(len(weights))])
        d = np.ascontiguousarray(np.asarray(dims, dtype=np.int32))
        f = np.ascontiguousarray(freqs.detach().cpu().numpy().astype(np.float32))
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_mlp_set_weights(self._h, n_hidden, d.ctypes.data, Wp, bp, f.ctypes.data, self._stream()))
        self._train_state = None  # setting weights detaches a training image
        self._image_owner = None

    def set_label_embedding(self, weight: torch.Tensor | None, period: int = 0) -> None:
        """MODEL:610-612 as table rows (mtk_mlp_set_label_embedding): call before set_weights; None clears."""
        with torch.cuda.device(self.device):
            if weight is None:
                check(_lib.lib().mtk_mlp_set_label_embedding(self._h, None, 0, 0, 0, self._stream()))
                return
            require_cuda(weight, "set_label_embedding")
            w = weight.detach().float().contiguous()
            check(_lib.lib().mtk_mlp_set_label_embedding(self._h, w.data_ptr(), int(w.shape[0]), int(w.shape[1]), int(period), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()

    def mlp_forward(self, x: torch.Tensor, t, out: torch.Tensor | None = None, precision: str = "fp32",
                    x_strides=None, out_strides=None, B: int | None = None) -> torch.Tensor:
        """x: [B, D] fp32 (any strides).  t: int64 [B] tensor or a Python int."""
        require_cuda(x, "mlp_forward")
        if B is None:
            B = x.shape[0]
        xs = x_strides if x_strides is not None else (x.stride(0), x.stride(1))
        if out is None:
            out = torch.empty((B, self.D), dtype=torch.float32, device=x.device)
        os_ = out_strides if out_strides is not None else (out.stride(0), out.stride(1))
        tv, ts = (t, 0) if isinstance(t, torch.Tensor) else (None, int(t))
        if tv is not None:
            assert tv.dtype == torch.int64 and tv.is_contiguous() and tv.numel() == B
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_mlp_forward(self._h, B, ptr(x), xs[0], xs[1], ptr(tv), ts, ptr(out), os_[0], os_[1],
                                             PRECISIONS[precision], self._stream()))
        return out

    def state_input_ok(self, precision: str) -> bool:
        """Can the denoiser read the reverse chain's state (z, z_idx) directly at this precision (mtk_mlp_state_input_ok)?"""
        return bool(_lib.lib().mtk_mlp_state_input_ok(self._h, PRECISIONS[precision]))

    def mlp_forward_state(self, z: torch.Tensor, z_idx: torch.Tensor, t: int, out: torch.Tensor, B: int, precision: str = "bf16",
                          out_strides=None) -> torch.Tensor:
        """The denoiser of one reverse step on the chain's state: z [max(Nn, 1), B] fp64 and z_idx [max(C, 1), B] int32 (column-major
        state of mtk_sample_init / mtk_p_sample_step) instead of the materialised [B, D] input (mtk_mlp_forward_state)."""
        assert z.dtype == torch.float64 and z_idx.dtype == torch.int32 and z.is_contiguous() and z_idx.is_contiguous()
        assert z.shape[1] == B and z_idx.shape[1] == B
        os_ = out_strides if out_strides is not None else (out.stride(0), out.stride(1))
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_mlp_forward_state(self._h, B, ptr(z), ptr(z_idx), B, int(t), ptr(out), os_[0], os_[1], PRECISIONS[precision],
                                                   self._stream()))
        return out

    def mlp_forward_noalloc(self, x: torch.Tensor, t, out: torch.Tensor, x_strides=None, out_strides=None, B: int | None = None) -> torch.Tensor:
        """mtk_mlp_forward_noalloc: the exact-fp32 denoiser through the library's static device scratch (no hipMalloc /
        hipFree / synchronisation inside the call; B <= 4096 rows, layers <= 1024 wide).  ``out`` is caller-allocated."""
        require_cuda(x, "mlp_forward_noalloc")
        if B is None:
            B = x.shape[0]
        xs = x_strides if x_strides is not None else (x.stride(0), x.stride(1))
        os_ = out_strides if out_strides is not None else (out.stride(0), out.stride(1))
        tv, ts = (t, 0) if isinstance(t, torch.Tensor) else (None, int(t))
        if tv is not None:
            assert tv.dtype == torch.int64 and tv.is_contiguous() and tv.numel() == B
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_mlp_forward_noalloc(self._h, B, ptr(x), xs[0], xs[1], ptr(tv), ts, ptr(out), os_[0], os_[1], self._stream()))
        return out

    # ------------------------------------------------------------------ loss side
    def q_sample(self, x0, t, eps, u, x_in, xi_strides, xt_idx, seed=0, row0=0):
        B = x0.shape[0]
        assert x0.dtype == torch.float32 and x0.stride(1) == 1 and t.dtype == torch.int64
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_q_sample(self._h, B, ptr(x0), x0.stride(0), ptr(t), ptr(eps), ptr(u), ptr(x_in), xi_strides[0],
                                          xi_strides[1], ptr(xt_idx), seed, row0, self._stream()))

    def loss_rows(self, model_out, mo_strides, x0, xt_idx, eps, t, loss_multi, loss_gauss, seed=0, row0=0):
        B = t.shape[0]
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_loss_rows(self._h, B, ptr(model_out), mo_strides[0], mo_strides[1], ptr(x0),
                                           x0.stride(0) if x0 is not None else 0, ptr(xt_idx), ptr(eps), ptr(t), ptr(loss_multi),
                                           ptr(loss_gauss), seed, row0, self._stream()))

    def gauss_kl_rows(self, model_out, mo_strides, x0, x_t, xt_strides, t, loss):
        """KL Gaussian loss rows (GaussianLossType.KL, GMD:431-484): ``loss`` is a [B] fp64 tensor."""
        assert loss.dtype == torch.float64 and x0.stride(1) == 1
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_gauss_kl_rows(self._h, t.shape[0], ptr(model_out), mo_strides[0], mo_strides[1], ptr(x0), x0.stride(0),
                                               ptr(x_t), xt_strides[0], xt_strides[1], ptr(t), ptr(loss), self._stream()))

    def tf_score(self, x: torch.Tensor, noise: torch.Tensor, t_list, add_t: int = 0, precision: str = "fp32",
                 out: torch.Tensor | None = None) -> torch.Tensor:
        require_cuda(x, "tf_score")
        B, P = x.shape[0], noise.shape[0]
        assert x.dtype == torch.float32 and x.stride(1) == 1 and noise.is_contiguous() and noise.shape[1] == self.Nn
        tl = np.ascontiguousarray(np.asarray(t_list, dtype=np.int32))
        if out is None:
            out = torch.empty((len(tl), B, P), dtype=torch.float32, device=x.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_tf_score(self._h, B, P, ptr(x), x.stride(0), ptr(noise), len(tl), tl.ctypes.data, add_t, ptr(out),
                                          PRECISIONS[precision], self._stream()))
        return out

    # ------------------------------------------------------------------ reverse chain
    def sample_init(self, B, z0, u0, z, z_idx, x_in, xi_strides, seed=0, row0=0):
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_sample_init(self._h, B, ptr(z0), ptr(u0), ptr(z), ptr(z_idx), ptr(x_in), xi_strides[0], xi_strides[1],
                                             seed, row0, self._stream()))

    def p_sample_step(self, B, t, first, model_out, mo_strides, z, z_idx, noise, u, x_in_next, xi_strides, seed=0, row0=0, grad=None):
        """``grad``: optional [B, Nn] fp32 contiguous classifier-guidance gradient (condition_mean, GMD:585)."""
        nf64 = int(noise is not None and noise.dtype == torch.float64)
        if grad is not None and (grad.dtype != torch.float32 or not grad.is_contiguous() or tuple(grad.shape) != (B, self.Nn)):
            raise ValueError(f"guidance gradient must be a contiguous fp32 [{B}, {self.Nn}] tensor, got {tuple(grad.shape)} {grad.dtype}")
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_p_sample_step_guided(self._h, B, int(t), int(first), ptr(model_out), mo_strides[0], mo_strides[1], ptr(z),
                                                      ptr(z_idx), ptr(noise), nf64, ptr(u), ptr(grad), ptr(x_in_next), xi_strides[0],
                                                      xi_strides[1], seed, row0, self._stream()))

    def ddim_step(self, B, t, model_out, mo_strides, z, z_idx, u, x_in_next, xi_strides, seed=0, row0=0, eta=0.0, noise=None, grad=None):
        """gaussian_ddim_step + multinomial_ddim_step (GMD:1070-1118, 1211-1257).  ``noise``: [B, Nn] fp32 (``randn_like``) or None ->
        Philox, only read when eta > 0; ``grad``: [B, Nn] fp32 conditioning gradient of condition_score (GMD:617-661) or None."""
        for name, v in (("noise", noise), ("guidance gradient", grad)):
            if v is not None and (v.dtype != torch.float32 or not v.is_contiguous() or tuple(v.shape) != (B, self.Nn)):
                raise ValueError(f"{name} must be a contiguous fp32 [{B}, {self.Nn}] tensor, got {tuple(v.shape)} {v.dtype}")
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_ddim_step_ex(self._h, B, int(t), float(eta), ptr(model_out), mo_strides[0], mo_strides[1], ptr(z), ptr(z_idx),
                                              ptr(noise), ptr(u), ptr(grad), ptr(x_in_next), xi_strides[0], xi_strides[1], seed, row0,
                                              self._stream()))

    def ddim_reverse_step(self, B, t, model_out, mo_strides, z, x_in_next, xi_strides):
        """gaussian_ddim_reverse_step (GMD:1160-1186) on the numeric state."""
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_ddim_reverse_step(self._h, B, int(t), ptr(model_out), mo_strides[0], mo_strides[1], ptr(z), ptr(x_in_next),
                                                   xi_strides[0], xi_strides[1], self._stream()))

    def sample_decode(self, B, z, z_idx, out):
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_sample_decode(self._h, B, ptr(z), ptr(z_idx), ptr(out), self._stream()))

    # ------------------------------------------------------------------ training (a19)
    IMAGES = {"weights": 0, "grads": 1, "exp_avg": 2, "exp_avg_sq": 3, "ema": 4}

    def train_attach(self, freqs: torch.Tensor) -> torch.Tensor:
        """Allocate the [5, n] training state (weights, grads, Adam m, Adam v, EMA) and attach it."""
        n = int(_lib.lib().mtk_train_image_elems(self._h))
        if n <= 0:
            raise RuntimeError("train_attach: set the denoiser weights first")
        state = torch.empty((5, n), dtype=torch.float32, device=self.device)
        f = np.ascontiguousarray(freqs.detach().cpu().numpy().astype(np.float32))
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_attach(self._h, ptr(state), f.ctypes.data, self._stream()))
        self._train_state = state  # keeps the buffer alive while the context reads weights from it
        self._image_owner = None   # who writes image 0: "trainer" (ClavaDDPMTrainer) or "autograd" (the mixed_loss seam)
        return state

    def train_step(self, x0, t, eps, u, inv_global_batch, loss_multi=None, loss_gauss=None, seed=0, row0=0) -> None:
        B = x0.shape[0]
        assert x0.dtype == torch.float32 and x0.stride(1) == 1 and (t is None or (t.dtype == torch.int64 and t.is_contiguous()))
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_step(self._h, B, ptr(x0), x0.stride(0), ptr(t), ptr(eps), ptr(u), float(inv_global_batch),
                                            ptr(loss_multi), ptr(loss_gauss), seed, row0, self._stream()))

    def train_bucket_layout(self) -> list[tuple[int, int]]:
        """(offset, length) of each gradient bucket inside one image, in the order the backward completes them: head, hidden
        layers last to first, proj, time-embedding MLP (mtk_train_bucket_layout)."""
        n = int(_lib.lib().mtk_train_bucket_layout(self._h, 0, None, None))
        if n < 0:
            check(n)
        off, ln = (C.c_int64 * n)(), (C.c_int64 * n)()
        rc = int(_lib.lib().mtk_train_bucket_layout(self._h, n, off, ln))
        if rc < 0:
            check(rc)
        return [(int(off[i]), int(ln[i])) for i in range(n)]

    def train_set_bucket_events(self, events) -> None:
        """``events``: one torch.cuda.Event per gradient bucket (None clears).  The backward records event i on its stream when
        bucket i of the gradient image is final.  The events are kept alive here for as long as the library holds their handles."""
        if not events:
            check(_lib.lib().mtk_train_set_bucket_events(self._h, 0, None))
            self._bucket_events = None
            return
        with torch.cuda.device(self.device):
            for e in events:
                e.record()  # torch creates the hipEvent_t lazily: after one record() the handle exists
        handles = (C.c_void_p * len(events))(*[int(e.cuda_event) for e in events])
        check(_lib.lib().mtk_train_set_bucket_events(self._h, len(events), handles))
        self._bucket_events = list(events)

    def train_set_precision(self, precision: str) -> None:
        """"fp32" (default, the parity path) or "bf16": bf16 operands / fp32 accumulation in every GEMM of the iteration, fp32
        master weights and optimizer (mtk_train_set_precision)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"training precision {precision!r}: fp32 or bf16")
        check(_lib.lib().mtk_train_set_precision(self._h, PRECISIONS[precision]))

    def train_set_gaussian_loss(self, kind: str) -> None:
        """"mse" (GMD:506-532) or "kl" (the variational-bound term, GMD:431-484, 534-541): which Gaussian loss the training
        iteration (mtk_train_step / mtk_train_forward / mtk_train_backward) evaluates and differentiates."""
        if kind not in ("mse", "kl"):
            raise ValueError(f"Gaussian loss {kind!r}: mse or kl")
        check(_lib.lib().mtk_train_set_gaussian_loss(self._h, int(kind == "kl")))

    def train_set_deterministic(self, on: bool) -> None:
        """True (default): fixed-order reductions, gradients bit-identical run to run; False: fp32 hardware atomics."""
        check(_lib.lib().mtk_train_set_deterministic(self._h, int(bool(on))))

    def train_set_label_classes(self, n_labels: int) -> None:
        """A trainable label embedding [n_labels, dim_t] as a layer of the training image (mtk_train_set_label_classes): after
        set_weights, before train_attach; export / import then carry it after the time-embedding MLP's two layers."""
        check(_lib.lib().mtk_train_set_label_classes(self._h, int(n_labels)))

    def train_set_labels(self, y: torch.Tensor | None) -> None:
        """The int64 labels [B] of the next iteration's rows (kept alive here until replaced)."""
        if y is not None:
            assert y.dtype == torch.int64 and y.is_contiguous()
        self._labels = y
        check(_lib.lib().mtk_train_set_labels(self._h, ptr(y)))

    def train_set_loss_means(self, means: torch.Tensor | None) -> None:
        """``means`` (2 fp32 on the device, kept alive here until replaced): later iterations also write the batch means of their
        two per-record loss vectors there (mtk_train_set_loss_means); None switches it off."""
        if means is not None:
            assert means.dtype == torch.float32 and means.is_contiguous() and means.numel() == 2
        self._loss_means = means
        check(_lib.lib().mtk_train_set_loss_means(self._h, ptr(means)))

    def train_set_deferred_reduce(self, on: bool) -> None:
        """Single-device training: train_step leaves the split-batch partials of the gradients to train_apply's optimizer pass
        (mtk_train_set_deferred_reduce); the gradient image is complete after train_apply or train_export("grads")."""
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_set_deferred_reduce(self._h, int(bool(on)), self._stream()))

    def train_timesteps(self, B: int) -> torch.Tensor:
        """The timesteps the last iteration drew itself (train_step with ``t=None``; mtk_train_timesteps)."""
        t = torch.empty((B,), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_timesteps(self._h, B, ptr(t), self._stream()))
        return t

    def train_set_dropout(self, p: float, seed: int = 0) -> None:
        """MODEL:162-180 in training mode: hidden activations dropped with probability ``p`` by Philox masks keyed by ``seed``
        (mtk_train_set_dropout); 0 switches it off."""
        check(_lib.lib().mtk_train_set_dropout(self._h, float(p), int(seed) % (1 << 64)))

    def train_dropout_mask(self, layer: int, B: int, row0: int = 0) -> torch.Tensor:
        """uint8 [B, width] keep mask of hidden layer ``layer`` under the current setting (mtk_train_dropout_mask)."""
        out = torch.empty((B, self._hidden_widths[layer]), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_dropout_mask(self._h, layer, B, row0, ptr(out), self._stream()))
        return out

    def train_forward(self, x0, t, eps, u, loss_multi=None, loss_gauss=None, seed=0, row0=0) -> int:
        """Forward half of train_step (activations kept on the context).  Returns a token that train_backward checks."""
        B = x0.shape[0]
        assert x0.dtype == torch.float32 and x0.stride(1) == 1 and t.dtype == torch.int64 and t.is_contiguous()
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_forward(self._h, B, ptr(x0), x0.stride(0), ptr(t), ptr(eps), ptr(u), ptr(loss_multi), ptr(loss_gauss),
                                               seed, row0, self._stream()))
        self._fwd_token = getattr(self, "_fwd_token", 0) + 1
        return self._fwd_token

    def train_backward(self, token, x0, t, eps, w_multi, w_gauss, seed=0, row0=0) -> None:
        """Backward half: gradient image <- d(w_multi sum loss_multi + w_gauss sum loss_gauss)/d weights of the forward
        that returned ``token`` (a later forward on this context has overwritten its activations: RuntimeError)."""
        if token != getattr(self, "_fwd_token", None):
            raise RuntimeError("backward through mixed_loss after another differentiable mixed_loss call on the same model: "
                               "the context keeps the activations of one forward pass at a time")
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_backward(self._h, x0.shape[0], ptr(x0), x0.stride(0), ptr(t), ptr(eps), float(w_multi), float(w_gauss),
                                                seed, row0, self._stream()))

    def train_backward_dev(self, token, x0, t, eps, w_dev: torch.Tensor, seed=0, row0=0) -> None:
        """train_backward with the weights of the two loss sums in a device tensor ``w_dev`` = [w_multi, w_gauss] (fp32): no host
        read of an autograd gradient, so no stream synchronisation (mtk_train_backward_dev)."""
        if token != getattr(self, "_fwd_token", None):
            raise RuntimeError("backward through mixed_loss after another differentiable mixed_loss call on the same model: "
                               "the context keeps the activations of one forward pass at a time")
        assert w_dev.is_cuda and w_dev.dtype == torch.float32 and w_dev.numel() == 2 and w_dev.is_contiguous()
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_backward_dev(self._h, x0.shape[0], ptr(x0), x0.stride(0), ptr(t), ptr(eps), ptr(w_dev), seed, row0,
                                                    self._stream()))

    def train_apply(self, lr, weight_decay, step, betas=(0.9, 0.999), eps=1e-8, ema_rate=0.999) -> None:
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_train_apply(self._h, float(lr), float(weight_decay), float(betas[0]), float(betas[1]), float(eps),
                                             int(step), float(ema_rate), self._stream()))

    def _image_io(self, fn, image, tensors):
        Wp = (C.c_void_p * len(tensors))(*[W.data_ptr() for W, _ in tensors])
        bp = (C.c_void_p * len(tensors))(*[b.data_ptr() for _, b in tensors])
        with torch.cuda.device(self.device):
            check(fn(self._h, self.IMAGES[image] if isinstance(image, str) else int(image), Wp, bp, self._stream()))

    def train_export(self, image, tensors) -> None:
        """tensors: [(W, b)] contiguous fp32 device tensors in set_weights order, overwritten in place."""
        for W, b in tensors:
            assert W.is_cuda and W.is_contiguous() and b.is_contiguous() and W.dtype == torch.float32
        self._image_io(_lib.lib().mtk_train_export, image, tensors)

    def train_import(self, image, tensors) -> None:
        keep = [(W.detach().float().contiguous(), b.detach().float().contiguous()) for W, b in tensors]
        self._image_io(_lib.lib().mtk_train_import, image, keep)
