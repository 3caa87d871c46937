"""Drop-in for the reference's ``GaussianMultinomialDiffusion`` (GMD:155) on MI355X.

Same constructor, buffers (14 registered + 4 plain attributes), method names and return types on
the hot path: ``mixed_loss`` (GMD:1013), ``gaussian_q_sample`` (GMD:302), ``_gaussian_loss``
(GMD:505), ``sample`` (GMD:1370), ``sample_all`` (GMD:1426).  All arithmetic runs in
libmtk_hip.so; this file only draws random numbers in the reference's order (so a seeded torch
generator reproduces the reference's injected noise) and moves pointers.

Extra, not in the reference: ``mixed_loss_rows`` returns the per-record losses whose means
``mixed_loss`` returns; ``rng="philox"`` draws inside the kernels instead of through torch.
"""
from __future__ import annotations

from enum import Enum
from typing import Any

import numpy as np
import torch
from torch import Tensor

from . import schedule
from .engine import Engine
from .schedule import SchedulerType


class GaussianLossType(Enum):
    MSE = "mse"
    KL = "kl"


class GaussianParametrization(Enum):
    EPS = "eps"
    X0 = "x0"


class Parametrization(Enum):
    X0 = "x0"
    DIRECT = "direct"


class SampleMethod(Enum):
    """GMD:70."""

    UNIFORM = "uniform"
    IMPORTANCE = "importance"


class FoundNaNsError(BaseException):
    """DU:305."""

    def __init__(self, message: str = "Found NANs during sampling."):
        super().__init__(message)


def _default_device() -> torch.device:
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _has_label_table(net) -> bool:
    return bool(getattr(net, "label_conditioned", False)) and getattr(net, "num_classes", 0) > 0


def _image_tensors(net):
    """(W, b) per layer in the training image's order (a label embedding sits after the time-embedding MLP: trainer._tensors)."""
    from .trainer import ClavaDDPMTrainer

    return ClavaDDPMTrainer._tensors(net)


class _MixedLossFunction(torch.autograd.Function):
    """The autograd seam of ``mixed_loss`` (TRAINER:89-91: ``loss = lm + lg; loss.backward()``): forward is
    mtk_train_forward (noising, denoiser forward with the activations kept on the device context, per-record losses),
    backward is mtk_train_backward into the context's gradient image, handed to autograd as the parameters' gradients.
    fp32 MFMA arithmetic, like ClavaDDPMTrainer's fused iteration (same kernels)."""

    @staticmethod
    def forward(ctx, diffusion, x0, t, eps, u, seed, y, *params):
        eng = diffusion._autograd_engine()
        if y is not None:
            eng.train_set_labels(y)  # MODEL:610-612: the label-conditioned denoiser's targets (GMD:1042)
        B, dev = x0.shape[0], x0.device
        lm = torch.empty((B,), dtype=torch.float32, device=dev) if eng.C > 0 else None
        lg = torch.empty((B,), dtype=torch.float32, device=dev) if eng.Nn > 0 else None
        net = diffusion._denoise_fn
        p_drop = net.training_dropout() if hasattr(net, "training_dropout") else 0.0
        diffusion._dropout_calls = getattr(diffusion, "_dropout_calls", 0) + 1
        eng.train_set_dropout(p_drop, (seed + 0xD1B54A32D192ED03 * diffusion._dropout_calls) % (1 << 63) if p_drop else 0)  # MODEL:162-180
        ctx.token = eng.train_forward(x0, t, eps, u, lm, lg, seed=seed)
        ctx.eng, ctx.net, ctx.seed, ctx.B = eng, diffusion._denoise_fn, seed, B
        ctx.inputs = (x0, t, eps)
        ctx.needs = [p.requires_grad for p in params]
        zero = torch.zeros((), dtype=torch.float32, device=dev)
        return (lm.mean() if lm is not None else zero), (lg.mean() if lg is not None else zero)

    @staticmethod
    def backward(ctx, g_multi, g_gauss):
        x0, t, eps = ctx.inputs
        # the incoming gradients stay on the device ([w_multi, w_gauss] = g / B, read by the loss-gradient kernel): no host sync
        zero = torch.zeros((), dtype=torch.float32, device=x0.device)
        w_dev = torch.stack([(g_multi if g_multi is not None else zero).to(x0.device, torch.float32).reshape(()),
                             (g_gauss if g_gauss is not None else zero).to(x0.device, torch.float32).reshape(())]) / ctx.B
        ctx.eng.train_backward_dev(ctx.token, x0, t, eps, w_dev.contiguous(), seed=ctx.seed)
        bufs = [(torch.empty_like(W.data), torch.empty_like(b.data)) for W, b in _image_tensors(ctx.net)]
        ctx.eng.train_export("grads", bufs)
        label = bufs.pop(3)[0] if _has_label_table(ctx.net) else None  # image order: ..., temb.2, label embedding, blocks... (parameters: table last)
        flat = [g for pair in bufs for g in pair] + ([label] if label is not None else [])
        return (None,) * 7 + tuple(g if need else None for g, need in zip(flat, ctx.needs))


class GaussianMultinomialDiffusion(torch.nn.Module):
    def __init__(
        self,
        num_classes: np.ndarray,
        num_numerical_features: int,
        denoise_fn: torch.nn.Module,
        num_timesteps: int = 1000,
        gaussian_loss_type: GaussianLossType = GaussianLossType.MSE,
        gaussian_parametrization: GaussianParametrization = GaussianParametrization.EPS,
        parametrization: Parametrization = Parametrization.X0,
        scheduler_type: SchedulerType = SchedulerType.COSINE,
        device: torch.device | None = None,
    ):
        super().__init__()
        device = torch.device(device) if device is not None else _default_device()
        num_classes = np.asarray(num_classes)
        # the reference raises ValueError lazily for these enums (GMD:408, 543, 832); do it up front
        if getattr(gaussian_loss_type, "value", None) not in ("mse", "kl"):
            raise ValueError(f"Unsupported Gaussian loss type: {gaussian_loss_type}")
        if getattr(gaussian_parametrization, "value", None) not in ("eps", "x0"):
            raise ValueError(f"Unsupported Gaussian parametrization: {gaussian_parametrization}")
        if getattr(parametrization, "value", None) not in ("x0", "direct"):
            raise ValueError(f"Unsupported parametrization: {parametrization}")
        scheduler_type = SchedulerType(getattr(scheduler_type, "value", scheduler_type))

        self.num_numerical_features = num_numerical_features
        self.num_classes = num_classes
        self.num_classes_expanded = torch.from_numpy(
            np.concatenate([num_classes[i].repeat(num_classes[i]) for i in range(len(num_classes))])
        ).to(device)
        self.slices_for_classes = [np.arange(self.num_classes[0])]
        offsets = np.cumsum(self.num_classes)
        for i in range(1, len(offsets)):
            self.slices_for_classes.append(np.arange(offsets[i - 1], offsets[i]))
        self.offsets = torch.from_numpy(np.append([0], offsets)).to(device)

        self._denoise_fn = denoise_fn
        self.gaussian_loss_type = gaussian_loss_type
        self.gaussian_parametrization = gaussian_parametrization
        self.num_timesteps = num_timesteps
        self.parametrization = parametrization
        self.scheduler_type = scheduler_type
        self.device = device

        buffers = schedule.buffer_values(scheduler_type, num_timesteps)
        post = schedule.posterior_values(buffers)
        self.posterior_variance = post["posterior_variance"]  # fp64, never cast (GMD:223)
        self.posterior_log_variance_clipped = post["posterior_log_variance_clipped"].to(device)
        self.posterior_mean_coef1 = post["posterior_mean_coef1"].to(device)
        self.posterior_mean_coef2 = post["posterior_mean_coef2"].to(device)
        self._host_tables = {k: v.float() for k, v in buffers.items()}
        self._host_tables.update({k: post[k] for k in ("posterior_mean_coef1", "posterior_mean_coef2", "model_log_variance", "model_variance",
                                                       "posterior_log_variance_clipped")})
        for key, value in buffers.items():
            self.register_buffer(key, value.float().to(device))

        self.results_on_device = False  # True: sample / sample_all return device tensors (the reference moves them to the CPU, GMD:1423),
        #                                 so that e.g. transforms.QuantileInverse runs on them without a host round trip
        self.rng = "torch"  # "torch": draw like the reference does; "philox": draw inside the kernels
        self.state_input = True  # reverse chain: let the denoiser read the chain's state (z, z_idx) where the library has that path
        #                          (mtk_mlp_forward_state: no [B, D] input matrix is written or converted per step); False: always materialise it
        self.use_graph = False  # rng="philox" only: run the reverse chain of sample() as one captured HIP graph (_replay_chain_graph)
        self.math = None    # None: "fast" with rng="philox", "reference" otherwise; or force "reference"/"fast"
        self.precision = getattr(denoise_fn, "precision", "fp32")
        self._engine: Engine | None = None

    # ------------------------------------------------------------------ plumbing
    def __getstate__(self):
        """Pickles (TRAIN:122-123, 298 dump the whole object) without the device contexts: rebuilt lazily after loading."""
        state = dict(self.__dict__)
        state["_engine"] = None
        state.pop("_aux", None)
        state.pop("_graphs", None)
        return state

    @property
    def _has_cat(self) -> bool:
        return len(self.num_classes) > 0 and int(self.num_classes[0]) != 0

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        # follow .to(device) like the reference's buffers do
        self.device = self.alphas.device
        for name in ("posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "num_classes_expanded", "offsets"):
            setattr(self, name, fn(getattr(self, name)))
        return out

    def engine(self) -> Engine:
        dev = self.alphas.device
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(dev, self.num_timesteps, self.num_numerical_features, self.num_classes, self._host_tables)
            if getattr(self._denoise_fn, "label_conditioned", False) and getattr(self._denoise_fn, "num_classes", 0) > 0:
                self._denoise_fn.num_timesteps = self.num_timesteps  # own context: (classes + 1) x T table rows (model.MLPDiffusion.engine)
            elif hasattr(self._denoise_fn, "attach_engine"):
                self._denoise_fn.attach_engine(self._engine)
        if self._label_net():
            self._denoise_fn.engine(dev)
        elif hasattr(self._denoise_fn, "sync_weights"):
            if getattr(self._engine, "_train_state", None) is not None and self.precision != "fp32" and not getattr(self, "_keep_training_image", False):
                self._denoise_fn._synced_key = None  # the packed bf16 images do not follow a training image: detach and re-pack
            self._denoise_fn.sync_weights()
        self._engine.set_math(self.math or ("fast" if self.rng == "philox" else "reference"))
        # GMD:399-408 / 822-832: what the denoiser's two output blocks mean (mtk_ctx_set_parametrization)
        self._engine.set_parametrization(getattr(self.gaussian_parametrization, "value", self.gaussian_parametrization) == "x0",
                                         getattr(self.parametrization, "value", self.parametrization) == "direct")
        return self._engine

    def _label_net(self) -> bool:
        """IsTargetConditioned.EMBEDDING with classes (MODEL:610-612): the denoiser runs on its own context and takes ``y``."""
        net = self._denoise_fn
        return bool(getattr(net, "label_conditioned", False)) and getattr(net, "num_classes", 0) > 0

    def _denoise_cm(self, eng: Engine, x_in: Tensor, t, B: int, out: Tensor, y: Tensor | None = None) -> None:
        """Denoiser on the column-major [D][B] intermediates (``y``: the labels of a label-conditioned denoiser)."""
        net = self._denoise_fn
        if not hasattr(net, "sync_weights"):
            raise TypeError("the MI355X path needs midst_toolkit_amd.MLPDiffusion as denoise_fn")
        if self._label_net():
            tv = t if isinstance(t, Tensor) else torch.full((B,), int(t), dtype=torch.int64, device=eng.device)
            eng = net.engine(eng.device)
            t = net.table_rows(tv, y)
        eng.mlp_forward(x_in, t, out=out, precision=self.precision, x_strides=(1, B), out_strides=(1, B), B=B)

    # ------------------------------------------------------------------ Gaussian pieces used by callers (TF:86-93)
    def gaussian_q_sample(self, features_start: Tensor, timestep: Tensor, noise: Tensor | None = None) -> Tensor:
        """GMD:302 via mtk_q_sample with the categorical block switched off."""
        if noise is None:
            noise = torch.randn_like(features_start)
        assert noise.shape == features_start.shape
        x = features_start.float().contiguous()
        B, n = x.shape
        eng = self._aux_engine(n)
        out = torch.empty_like(x)
        eng.q_sample(x, timestep.to(x.device, torch.int64).contiguous(), noise.float().contiguous(), None, out, (n, 1), None)
        return out

    def _gaussian_loss(self, model_out: Tensor, features_start: Tensor, features_timestep: Tensor, timestep: Tensor,
                       noise: Tensor, model_kwargs: dict[str, Any] | None = None) -> Tensor:
        """GMD:505.  MSE branch: mean_d (noise - model_out)^2 per record (fp32); KL branch (GMD:534-541): the variational
        bound term of each record in bits per dimension (fp64, like the reference's)."""
        mo = model_out.float().contiguous()
        B, n = mo.shape
        eng = self._aux_engine(n)
        if getattr(self.gaussian_loss_type, "value", "mse") == "kl":
            x0, xt = features_start.float().contiguous(), features_timestep.float().contiguous()
            if x0.shape != mo.shape or xt.shape != mo.shape:
                raise AssertionError(f"Expected shapes to be equal, but got {tuple(mo.shape)}, {tuple(x0.shape)}, {tuple(xt.shape)}")
            out = torch.empty((B,), dtype=torch.float64, device=mo.device)
            eng.gauss_kl_rows(mo, (n, 1), x0, xt, (n, 1), timestep.to(mo.device, torch.int64).contiguous(), out)
            return out
        out = torch.empty((B,), dtype=torch.float32, device=mo.device)
        eng.loss_rows(mo, (n, 1), None, None, noise.float().contiguous(), timestep.to(mo.device, torch.int64).contiguous(), None, out)
        return out

    def _aux_engine(self, n: int) -> Engine:
        """Numeric-only context of width n for the stand-alone Gaussian helpers."""
        cache = self.__dict__.setdefault("_aux", {})
        dev = self.alphas.device
        key = (n, str(dev))
        if cache.get(key) is None:
            cache[key] = Engine(dev, self.num_timesteps, n, [0], self._host_tables)
        cache[key].set_parametrization(getattr(self.gaussian_parametrization, "value", self.gaussian_parametrization) == "x0", False)
        return cache[key]

    # ------------------------------------------------------------------ loss entry point
    def sample_time(self, batch_size: int, device: torch.device, method: SampleMethod = SampleMethod.UNIFORM) -> tuple[Tensor, Tensor]:
        """GMD:945-980 (host logic, as in the reference): (timesteps, their probabilities).  UNIFORM is the draw ``mixed_loss`` makes
        (GMD:1025).  IMPORTANCE weighs timesteps by sqrt(lt_history) once every timestep has more than 10 recorded losses in
        ``lt_count`` and falls back to UNIFORM until then; nothing in the reference writes those two buffers, so the caller owns
        their bookkeeping here as there."""
        method = SampleMethod(getattr(method, "value", method))
        if method == SampleMethod.IMPORTANCE:
            if not bool((self.lt_count > 10).all()):
                return self.sample_time(batch_size, device, method=SampleMethod.UNIFORM)
            lt_sqrt = torch.sqrt(self.lt_history + 1e-10) + 0.0001
            lt_sqrt[0] = lt_sqrt[1]  # GMD:967: the decoder term is overwritten with L1
            pt_all = (lt_sqrt / lt_sqrt.sum()).to(device)
            timestep = torch.multinomial(pt_all, num_samples=batch_size, replacement=True).to(device)
            return timestep, pt_all.gather(dim=0, index=timestep)
        timestep = torch.randint(0, self.num_timesteps, (batch_size,), device=device).long()
        return timestep, torch.ones_like(timestep).float() / self.num_timesteps

    def _draw_loss_noise(self, features: Tensor):
        """The reference's draw order (SURVEY finding 0.5): randint → randn_like → per-column rand_like."""
        B = features.shape[0]
        dev = features.device
        t = torch.randint(0, self.num_timesteps, (B,), device=dev).long()
        Nn = self.num_numerical_features
        eps = torch.randn_like(features[:, :Nn]) if Nn > 0 else None
        u = None
        if self._has_cat and features.shape[1] - Nn > 0:
            u = torch.cat([torch.rand((B, int(k)), device=dev, dtype=torch.float32) for k in self.num_classes], dim=1)
        return t, eps, u

    @torch.no_grad()
    def mixed_loss_rows(self, features: Tensor, timestep: Tensor | None = None, noise: Tensor | None = None,
                        uniform: Tensor | None = None, seed: int = 0, row0: int = 0, y: Tensor | None = None):
        """Per-record (multinomial / C, gaussian) losses.  With ``timestep``/``noise``/``uniform`` given
        they are injected; otherwise drawn per ``self.rng``.  Returns 1-element zero tensors on CPU
        for an absent side, exactly like GMD:1047-1048."""
        if not features.is_cuda:
            raise RuntimeError("mixed_loss: midst_toolkit_amd has no CPU path; move the model and batch to the MI355X")
        eng = self.engine()
        x0 = features.float().contiguous()
        B = x0.shape[0]
        Nn, C, D = eng.Nn, eng.C, eng.D
        dev = x0.device
        if timestep is None:
            if self.rng == "torch":
                timestep, noise, uniform = self._draw_loss_noise(x0)
            else:
                timestep = torch.randint(0, self.num_timesteps, (B,), device=dev).long()
                seed = int(torch.randint(0, 2**62, (1,)).item()) if seed == 0 else seed
        t = timestep.to(dev, torch.int64).contiguous()
        eps = noise.float().contiguous() if noise is not None else None
        u = uniform.float().contiguous() if uniform is not None else None
        x_in = torch.empty((D, B), dtype=torch.float32, device=dev)
        out = torch.empty((D, B), dtype=torch.float32, device=dev)
        xt_idx = torch.empty((max(C, 1), B), dtype=torch.int32, device=dev)
        eng.q_sample(x0, t, eps, u, x_in, (1, B), xt_idx, seed, row0)
        self._denoise_cm(eng, x_in, t, B, out, y)
        lm = torch.empty((B,), dtype=torch.float32, device=dev) if C > 0 else None
        kl = getattr(self.gaussian_loss_type, "value", "mse") == "kl"  # GMD:534: _vb_terms_bpd instead of the MSE, fp64 like the reference's
        lg = torch.empty((B,), dtype=torch.float64 if kl else torch.float32, device=dev) if Nn > 0 else None
        if kl and Nn > 0:
            eng.gauss_kl_rows(out, (1, B), x0, x_in, (1, B), t, lg)
            if C > 0:
                eng.loss_rows(out, (1, B), x0, xt_idx, eps, t, lm, None, seed, row0)
        else:
            eng.loss_rows(out, (1, B), x0, xt_idx, eps, t, lm, lg, seed, row0)
        zero = torch.zeros((1,)).float()
        return (lm if lm is not None else zero), (lg if lg is not None else zero)

    def mixed_loss(self, features: Tensor, outputs: dict[str, Tensor]) -> tuple[Tensor, Tensor]:
        """GMD:1013: (mean multinomial loss / C, mean Gaussian loss).  ``outputs['y']`` reaches the denoiser only when it is
        label-conditioned (MODEL:610); every other denoiser ignores it, as the reference's does."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            y = (outputs or {}).get("y") if self._label_net() else None
            if self._label_net() and y is None:
                raise ValueError("mixed_loss: a label-conditioned denoiser needs outputs['y'] (GMD:1042)")
            return self._mixed_loss_autograd(features, y)
        lm, lg = self.mixed_loss_rows(features, y=(outputs or {}).get("y") if self._label_net() else None)
        return lm.mean(), lg.mean()

    # ------------------------------------------------------------------ differentiable entry point (TRAINER:89-91)
    def _autograd_engine(self) -> Engine:
        """The context with a training image attached and image 0 equal to the module's current parameters."""
        from .model import timestep_frequencies

        net = self._denoise_fn
        if self._engine is None or self._engine.device != self.alphas.device:
            self.engine()
        eng = self._engine
        key = tuple((p.data_ptr(), p._version) for wb in net._weights() for p in wb)
        if _has_label_table(net):
            key += ((net.label_embedding.weight.data_ptr(), net.label_embedding.weight._version),)
        if getattr(eng, "_train_state", None) is None and _has_label_table(net):
            # the diffusion object's context holds no denoiser for such a net (inference runs on the net's own table-row context):
            # layers, then an image with the label embedding as one more layer, filled from the module
            freqs = timestep_frequencies(net.timestep_dimension)
            eng.set_weights(net._weights(), net._dims(), freqs)
            eng.train_set_label_classes(int(net.num_classes))
            eng.train_attach(freqs)
            eng.train_import("weights", _image_tensors(net))
            eng._image_owner = "autograd"
            self._ag_key = key
        elif getattr(eng, "_train_state", None) is None:
            net.sync_weights()  # layer shapes and current weights into the context, then the [5, n] image on top
            eng.train_attach(timestep_frequencies(net.timestep_dimension))
            eng._image_owner = "autograd"
            self._ag_key = key
        elif getattr(eng, "_image_owner", None) == "trainer":
            # A ClavaDDPMTrainer trains on this context: image 0 IS the model (the nn.Parameters are only refreshed by
            # sync_modules, so they usually still hold the initial weights).  A differentiable mixed_loss between trainer steps
            # - a validation loss computed without torch.no_grad() - evaluates those live weights and leaves them alone;
            # importing the stale parameters here would silently reset the run while Adam's moments and the EMA carried on.
            pass
        elif key != getattr(self, "_ag_key", None):  # an optimizer step (or a load) changed the parameters
            eng.train_import("weights", _image_tensors(net))
            self._ag_key = key
        return eng

    def _mixed_loss_autograd(self, features: Tensor, y: Tensor | None = None) -> tuple[Tensor, Tensor]:
        """``mixed_loss`` with an autograd graph: ``(lm + lg).backward()`` fills ``param.grad`` of the denoiser's 12 tensors
        from the HIP backward pass, so the reference's own ``ClavaDDPMTrainer`` loop (zero_grad, mixed_loss, backward,
        ``torch.optim.AdamW.step``; TRAINER:70-94) and ``clavaddpm_fine_tuning.py`` run against the drop-in unchanged.
        One differentiable forward is outstanding per model at a time (its activations live on the device context)."""
        if not features.is_cuda:
            raise RuntimeError("mixed_loss: midst_toolkit_amd has no CPU path; move the model and batch to the MI355X")
        x0 = features.float().contiguous()
        B, dev = x0.shape[0], x0.device
        seed = 0
        if self.rng == "torch":
            t, eps, u = self._draw_loss_noise(x0)
        else:
            t = torch.randint(0, self.num_timesteps, (B,), device=dev).long()
            eps = u = None
            seed = int(torch.randint(1, 2**62, (1,)).item())
        t = t.to(dev, torch.int64).contiguous()
        eps = eps.float().contiguous() if eps is not None and eps.numel() else None
        u = u.float().contiguous() if u is not None and u.numel() else None
        net = self._denoise_fn
        eng = self._autograd_engine()
        eng.set_math(self.math or ("fast" if self.rng == "philox" else "reference"))
        eng.train_set_gaussian_loss(getattr(self.gaussian_loss_type, "value", "mse"))  # GMD:534: MSE or the variational-bound term
        params = [p for wb in net._weights() for p in wb] + ([net.label_embedding.weight] if _has_label_table(net) else [])
        labels = y.to(dev).long().reshape(-1).contiguous() if y is not None else None
        lm, lg = _MixedLossFunction.apply(self, x0, t, eps, u, seed, labels, *params)
        zero = torch.zeros((1,)).float().mean()  # GMD:1047-1048, 1068: an absent side is the mean of a CPU zeros((1,)), 0-dim
        return (lm if eng.C > 0 else zero), (lg if eng.Nn > 0 else zero)

    @classmethod
    def from_reference_state(cls, state_dict: dict[str, Tensor], ctor_args: dict[str, Any], device: torch.device | None = None) -> "GaussianMultinomialDiffusion":
        """Model ingestion without unpickling reference classes: ``state_dict`` is the reference object's
        ``state_dict()`` (26 entries: ``_denoise_fn.*`` parameters + the 14 schedule buffers; extra or missing buffers are
        checked by a strict load) and ``ctor_args`` the keyword arguments of GMD:156-168 except ``denoise_fn`` -
        ``num_classes``, ``num_numerical_features`` and optionally ``num_timesteps``, ``gaussian_loss_type``,
        ``gaussian_parametrization``, ``parametrization``, ``scheduler_type`` (enums or their string values).  The denoiser's
        shape (input width, time-embedding width, hidden layers) is read off the weight tensors."""
        from .model import DiffusionParameters, IsTargetConditioned, MLPDiffusion

        sd = {k: (v if isinstance(v, Tensor) else torch.as_tensor(v)) for k, v in state_dict.items()}
        proj = sd["_denoise_fn.proj.weight"]
        dim_t, width = int(proj.shape[0]), int(proj.shape[1])
        layers = []
        while f"_denoise_fn.mlp.blocks.{len(layers)}.linear.weight" in sd:
            layers.append(int(sd[f"_denoise_fn.mlp.blocks.{len(layers)}.linear.weight"].shape[0]))
        args = dict(ctor_args)
        args.pop("denoise_fn", None)
        num_classes = np.asarray(args.pop("num_classes"))
        nn_ = int(args.pop("num_numerical_features"))
        expect = nn_ + (int(num_classes.sum()) if len(num_classes) and int(num_classes[0]) != 0 else 0)
        if expect != width:
            raise AssertionError(f"state dict holds a denoiser {width} wide, num_numerical_features + sum(num_classes) = {expect}")
        coerce = {"gaussian_loss_type": GaussianLossType, "gaussian_parametrization": GaussianParametrization,
                  "parametrization": Parametrization, "scheduler_type": SchedulerType}
        for k, enum in coerce.items():
            if k in args:
                args[k] = enum(getattr(args[k], "value", args[k]))
        lab = sd.get("_denoise_fn.label_embedding.weight")  # IsTargetConditioned.EMBEDDING: [classes, dim_t] (an nn.Linear(1, dim_t) when classes = 0)
        n_lab = 0 if lab is None or "_denoise_fn.label_embedding.bias" in sd else int(lab.shape[0])
        net = MLPDiffusion(width, n_lab, IsTargetConditioned.NONE if lab is None else IsTargetConditioned.EMBEDDING,
                           DiffusionParameters(layers, 0.0), timestep_dimension=dim_t)
        device = torch.device(device) if device is not None else args.pop("device", None) or _default_device()
        args.pop("device", None)
        model = cls(num_classes, nn_, net, device=device, **args)
        model.load_state_dict(sd, strict=True)
        return model.to(device)

    # ------------------------------------------------------------------ reverse chain
    def _reverse_chain(self, B: int, draw_outputs, model_kwargs, conditioning_function, inj: dict | None,
                       ddim: bool = False, row0: int = 0, seed: int | None = None):
        """The loop shared by sample (GMD:1403-1424) and conditional_sample (GMD:1344-1367): T x (denoiser,
        mtk_p_sample_step[_guided]) on a device-resident fp64 state, then decode.  ``conditioning_function(features,
        timestep, **model_kwargs)`` is the caller's (GMD:614): it sees the current numeric state as the reference's
        would (fp32 on the first step, fp64 after) and returns d log p(y|x)/dx, which the step kernel folds into the mean.

        ``draw_outputs()`` returns the ``{"y": ...}`` dict; it is called AFTER the initial state draws, which is where the
        reference draws its labels (GMD:1392-1401: randn, per-column rand, then multinomial), so a seeded generator
        shared by ``target_dist`` and the state (both on the GPU) yields the reference's stream.
        ``row0`` / ``seed``: in-kernel Philox noise is keyed by (seed, row0 + row, column, step), so disjoint row ranges
        under one seed are independent streams - what sharded sampling uses (distributed.sample_all_sharded)."""
        eng = self.engine()
        dev = eng.device
        Nn, C, D, T = eng.Nn, eng.C, eng.D, self.num_timesteps
        philox = self.rng == "philox" and inj is None
        if philox:
            seed = int(torch.randint(1, 2**62, (1,)).item()) if seed is None else int(seed)
        else:
            seed = 0
        row0 = int(row0)
        model_kwargs = model_kwargs or {}

        def draw_u():
            return torch.cat([torch.rand((B, int(k)), device=dev, dtype=torch.float32) for k in self.num_classes], dim=1)

        z0 = u0 = None
        if inj is not None:
            z0, u0 = inj["z0"], inj.get("u0")
        elif not philox:
            z0 = torch.randn((B, Nn), device=dev)
            u0 = draw_u() if C else None
        outputs = draw_outputs()
        y = (outputs or {}).get("y") if self._label_net() else None  # GMD:1406: the denoiser takes **outputs
        if y is not None:
            y = y.to(dev)
        if philox and not ddim and conditioning_function is None and getattr(self, "use_graph", False) and y is None:
            sample = self._replay_chain_graph(eng, B, seed, row0)
            return (sample if getattr(self, "results_on_device", False) else sample.cpu()), outputs
        z = torch.empty((max(Nn, 1), B), dtype=torch.float64, device=dev)
        z_idx = torch.empty((max(C, 1), B), dtype=torch.int32, device=dev)
        from_state = self._state_route(eng, ddim, y)
        x_in = None if from_state else torch.empty((D, B), dtype=torch.float32, device=dev)
        out = torch.empty((D, B), dtype=torch.float32, device=dev)
        eng.sample_init(B, z0, u0, z, z_idx, x_in, (1, B), seed, row0)
        # Classifier guidance (GMD:1319-1368) needs the state and the timestep, not the denoiser's output: its forward / input
        # gradient runs on a side stream beside the denoiser and joins before the posterior step (a guided step at caller-sized
        # batches is two chains of small dependent launches; `guidance_stream = False` keeps everything on one stream)
        side = None
        # (only for this package's own conditioning function, classifier.get_conditioning_function: it draws no random numbers and
        # issues all its work on the current stream; any other callable keeps its place after the step's draws on the one stream)
        if (conditioning_function is not None and Nn and not ddim and getattr(self, "guidance_stream", True)
                and getattr(conditioning_function, "_mtk_stream_safe", False)):
            side = self.__dict__.get("_guidance_side")
            if side is None or side.device != dev:
                side = self.__dict__["_guidance_side"] = torch.cuda.Stream(device=dev)
        for i, t in enumerate(reversed(range(T))):
            grad_side = None
            if side is not None:
                cur = torch.cuda.current_stream(dev)
                side.wait_stream(cur)  # the previous step's state
                with torch.cuda.stream(side):
                    feats = z[:Nn].t()
                    if i == 0:
                        feats = feats.float()
                    tvec = torch.full((B,), t, device=dev, dtype=torch.long)
                    grad_side = conditioning_function(feats, tvec, **model_kwargs).detach().to(device=dev, dtype=torch.float32).contiguous()
            if from_state:  # GMD:1406's cat[z_norm, log_z].float() is formed inside the denoiser's first block
                eng.mlp_forward_state(z, z_idx, t, out, B, self.precision, out_strides=(1, B))
            else:
                self._denoise_cm(eng, x_in, t, B, out, y)
            n = u = grad = None
            if inj is not None:
                n = inj["noise"][i] if Nn and not ddim else None
                u = inj["u"][i] if C else None
            elif not philox:
                if Nn and not ddim:
                    n = torch.randn((B, Nn), device=dev, dtype=torch.float32 if i == 0 else torch.float64)
                elif Nn:
                    torch.randn((B, Nn), device=dev)  # GMD:1113 draws (and, eta = 0, discards) it: keep the generator in step
                u = draw_u() if C else None
            if ddim:  # GMD:1292-1308, eta = 0: no Gaussian noise enters (the reference draws it and multiplies by sigma = 0)
                if conditioning_function is not None and Nn:  # condition_score (GMD:1104-1105): the state is fp32 here
                    tvec = torch.full((B,), t, device=dev, dtype=torch.long)
                    grad = conditioning_function(z[:Nn].t().float().contiguous(), tvec, **model_kwargs)
                    grad = grad.detach().to(device=dev, dtype=torch.float32).contiguous()
                eng.ddim_step(B, t, out, (1, B), z, z_idx, u, x_in, (1, B), seed, row0, grad=grad)
                continue
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
                grad = grad_side
            elif conditioning_function is not None and Nn:
                feats = z[:Nn].t()  # [B, Nn] view of the column-major state: the classifier's input conversion makes the one copy
                if i == 0:
                    feats = feats.float()
                tvec = torch.full((B,), t, device=dev, dtype=torch.long)
                grad = conditioning_function(feats, tvec, **model_kwargs).detach().to(device=dev, dtype=torch.float32).contiguous()
            eng.p_sample_step(B, t, i == 0, out, (1, B), z, z_idx, n, u, x_in, (1, B), seed, row0, grad=grad)
        sample = torch.empty((B, Nn + C), dtype=torch.float64, device=dev)
        eng.sample_decode(B, z, z_idx, sample)
        sample = sample.float() if ddim else sample  # the reference's DDIM chain is fp32 end to end
        return (sample if getattr(self, "results_on_device", False) else sample.cpu()), outputs

    def _state_route(self, eng: Engine, ddim: bool, y) -> bool:
        """Does this chain's denoiser read the state directly (Engine.mlp_forward_state) instead of a materialised input?"""
        return bool(getattr(self, "state_input", True)) and not ddim and y is None and not self._label_net() and eng.state_input_ok(self.precision)

    def _replay_chain_graph(self, eng: Engine, B: int, seed: int, row0: int) -> Tensor:
        """The whole reverse chain of ``B`` rows - mtk_sample_init, T x (denoiser, mtk_p_sample_step), mtk_sample_decode: 2 T + 2
        launches - as ONE HIP graph (GMD:1403-1417 is a Python loop of T iterations in the reference).  In-kernel Philox noise
        only: no host draw happens inside the loop, and the kernels read {seed, row0} from a 16-byte device buffer
        (mtk_ctx_set_seed_source), so one captured graph serves every later call with the same batch size and weights.
        Captured once per (batch size, packed weights, precision, math) after mtk_ctx_reserve has sized the workspaces, and again
        whenever the context has reallocated a workspace since (mtk_ctx_workspace_generation);
        replays allocate nothing and make one launch call.  Same kernels and arguments as the eager loop: same samples."""
        dev = eng.device
        Nn, C, D, T = eng.Nn, eng.C, eng.D, self.num_timesteps
        from_state = self._state_route(eng, False, None)
        key = (B, str(dev), self.precision, self.math, getattr(self._denoise_fn, "_synced_key", None), id(eng), from_state)
        cache = self.__dict__.setdefault("_graphs", {})

        def generation():
            # the captured launches hold raw addresses of the contexts' workspaces: a call with more rows than were reserved
            # (eager or a capture at another batch size) reallocates them, and the counter says so
            gens = [eng.workspace_generation()]
            if self._label_net():
                gens.append(self._denoise_fn.engine(dev).workspace_generation())
            return tuple(gens)

        ent = cache.get(key)
        if ent is not None and ent["generation"] != generation():
            cache.pop(key)  # stale: replaying it would read and write freed device memory
            ent = None
        if ent is None:
            while len(cache) >= 4:  # static buffers per entry: keep a few batch sizes
                cache.pop(next(iter(cache)))
            ent = {"seed": torch.zeros(2, dtype=torch.int64, device=dev),
                   "z": torch.empty((max(Nn, 1), B), dtype=torch.float64, device=dev),
                   "z_idx": torch.empty((max(C, 1), B), dtype=torch.int32, device=dev),
                   "x_in": None if from_state else torch.empty((D, B), dtype=torch.float32, device=dev),
                   "out": torch.empty((D, B), dtype=torch.float32, device=dev),
                   "sample": torch.empty((B, Nn + C), dtype=torch.float64, device=dev)}
            eng.reserve(B)

            def chain(steps):
                eng.sample_init(B, None, None, ent["z"], ent["z_idx"], ent["x_in"], (1, B), 0, 0)
                for i, t in enumerate(reversed(range(T))):
                    if i >= steps:
                        break
                    if from_state:
                        eng.mlp_forward_state(ent["z"], ent["z_idx"], t, ent["out"], B, self.precision, out_strides=(1, B))
                    else:
                        self._denoise_cm(eng, ent["x_in"], t, B, ent["out"])
                    eng.p_sample_step(B, t, i == 0, ent["out"], (1, B), ent["z"], ent["z_idx"], None, None, ent["x_in"], (1, B), 0, 0)
                eng.sample_decode(B, ent["z"], ent["z_idx"], ent["sample"])

            eng.set_seed_source(ent["seed"])
            try:
                chain(2)  # eager warm-up: one-time function attributes and device queries happen outside the capture
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    chain(T)
            finally:
                eng.set_seed_source(None)
            ent["graph"] = graph
            ent["generation"] = generation()
            cache[key] = ent
        ent["seed"].copy_(torch.tensor([seed, row0], dtype=torch.int64))
        ent["graph"].replay()
        return ent["sample"].clone()

    def _target_fn(self, batch_size: int, target_dist: Tensor, inj: dict | None, targets: Tensor | None = None):
        def draw():
            if targets is not None:  # labels decided by the caller (sharded sampling: one global draw, distributed.shared_labels)
                target = targets
            elif inj is not None and "y" in inj:
                target = inj["y"]
            else:
                target = torch.multinomial(target_dist, num_samples=batch_size, replacement=True)
            return {"y": target.long().to(self.alphas.device)}

        return draw

    @torch.no_grad()
    def sample(self, batch_size: int, target_dist: Tensor, model_kwargs: dict[str, Any] | None = None,
               conditioning_function=None, _injected: dict | None = None, row0: int = 0, seed: int | None = None,
               _targets: Tensor | None = None) -> tuple[Tensor, dict[str, Tensor]]:
        """GMD:1370.  Returns ([B, Nn + C] fp64 on CPU, {"y": labels}).  ``conditioning_function``: classifier guidance
        of the numeric features (gaussian_p_sample GMD:699-702).  ``row0`` / ``seed`` (not in the reference): Philox
        stream position of row 0 and the shared seed, for sharded sampling; ``_targets``: the batch's labels, when the
        caller has drawn them (instead of ``torch.multinomial(target_dist)`` here)."""
        return self._reverse_chain(batch_size, self._target_fn(batch_size, target_dist, _injected, _targets), model_kwargs, conditioning_function,
                                   _injected, row0=row0, seed=seed)

    @torch.no_grad()
    def sample_ddim(self, batch_size: int, target_dist: Tensor, model_kwargs: dict[str, Any] | None = None,
                    conditioning_function=None, _injected: dict | None = None, row0: int = 0, seed: int | None = None,
                    _targets: Tensor | None = None) -> tuple[Tensor, dict[str, Tensor]]:
        """GMD:1259 (eta = 0).  Returns ([B, Nn + C] fp32 on CPU, {"y": labels}).  ``conditioning_function``: score-based
        guidance of the numeric features (condition_score, GMD:617-661)."""
        return self._reverse_chain(batch_size, self._target_fn(batch_size, target_dist, _injected, _targets), model_kwargs, conditioning_function,
                                   _injected, ddim=True, row0=row0, seed=seed)

    @torch.no_grad()
    def gaussian_ddim_sample(self, noise: Tensor, num_timesteps: int, outputs: dict[str, Tensor], eta: float = 0.0,
                             model_kwargs: Any | None = None, conditioning_function=None, _injected: list | None = None) -> Tensor:
        """GMD:1119: the Gaussian DDIM chain from ``noise`` [B, D] over ``reversed(range(num_timesteps))`` for any eta, with
        optional score-based guidance.  The denoiser sees the whole feature vector, as in the reference; the table must be
        numeric-only (the reference feeds ``features`` straight to the denoiser, so any categorical block would have to be
        part of them).  ``_injected``: the per-step ``randn_like`` draws (tests).  Returns fp32 [B, D] on the state's device."""
        return self._gaussian_ddim_loop(noise, list(reversed(range(num_timesteps))), eta, model_kwargs, conditioning_function, _injected, False)

    @torch.no_grad()
    def gaussian_ddim_reverse_sample(self, features: Tensor, num_timesteps: int, outputs: dict[str, Tensor]) -> Tensor:
        """GMD:1189: the deterministic DDIM map run from data towards noise, ``range(num_timesteps)``."""
        return self._gaussian_ddim_loop(features, list(range(num_timesteps)), 0.0, None, None, None, True)

    def _gaussian_ddim_loop(self, start: Tensor, steps: list, eta: float, model_kwargs, conditioning_function, inj, reverse: bool) -> Tensor:
        if self._has_cat:
            raise ValueError("gaussian_ddim_sample / gaussian_ddim_reverse_sample: numeric-only tables (num_classes = [0])")
        eng = self.engine()
        dev = eng.device
        Nn, D = eng.Nn, eng.D
        B = int(start.shape[0])
        if tuple(start.shape) != (B, D):
            raise ValueError(f"expected [B, {D}] features, got {tuple(start.shape)}")
        philox = self.rng == "philox" and inj is None
        seed = int(torch.randint(1, 2**62, (1,)).item()) if philox else 0
        model_kwargs = model_kwargs or {}
        z = start.to(device=dev, dtype=torch.float64).t().contiguous()
        z_idx = torch.empty((1, B), dtype=torch.int32, device=dev)
        x_in = start.to(device=dev, dtype=torch.float32).t().contiguous()
        out = torch.empty((D, B), dtype=torch.float32, device=dev)
        y = model_kwargs.get("y") if self._label_net() else None  # GMD:1147, 1212: the denoiser takes **outputs
        for i, t in enumerate(steps):
            self._denoise_cm(eng, x_in, t, B, out, y.to(dev) if y is not None else None)
            if reverse:
                eng.ddim_reverse_step(B, t, out, (1, B), z, x_in, (1, B))
                continue
            n = grad = None
            if inj is not None:
                n = inj[i].to(device=dev, dtype=torch.float32).contiguous()
            elif not philox:
                n = torch.randn((B, Nn), device=dev)  # GMD:1113, drawn at every step whatever eta is
            if conditioning_function is not None:
                tvec = torch.full((B,), t, device=dev, dtype=torch.long)
                grad = conditioning_function(z.t().float().contiguous(), tvec, **model_kwargs)
                grad = grad.detach().to(device=dev, dtype=torch.float32).contiguous()
            eng.ddim_step(B, t, out, (1, B), z, z_idx, None, x_in, (1, B), seed, 0, eta=eta, noise=n, grad=grad)
        return z.t().float().contiguous()

    def conditional_sample(self, targets: Tensor, model_kwargs: dict[str, Any] | None = None, conditioning_function=None,
                           _injected: dict | None = None) -> tuple[Tensor, dict[str, Tensor]]:
        """GMD:1319.  The reference's loop never draws an initial categorical state (``log_z`` stays ``[B, 0]``), so it
        only runs on numeric-only tables — the multi-table pipeline's case (SYNTH:385-423); the same restriction here."""
        if self._has_cat:
            raise ValueError("conditional_sample: the reference's loop (GMD:1344) has no initial categorical state; "
                             "it is only defined for num_classes = [0]")
        with torch.no_grad():
            dev = self.engine().device
            return self._reverse_chain(len(targets), lambda: {"y": targets.long().to(dev)}, model_kwargs, conditioning_function, _injected)

    def sample_all(self, num_samples: int, batch_size: int, target_dist: Tensor, ddim: bool = False,
                   model_kwargs: dict[str, Any] | None = None, conditioning_function=None, row0: int = 0,
                   seed: int | None = None, _targets: Tensor | None = None) -> tuple[Tensor, Tensor]:
        """GMD:1426.  ``row0`` / ``seed`` (not in the reference): see ``sample``; batch i of this call continues the
        Philox row range at ``row0 + i * batch_size``.  ``_targets`` [num_samples]: the labels of this call's rows when the
        caller has drawn them (sharded sampling); a last batch that runs past them repeats the final label for the rows
        that are cut off anyway."""
        all_targets, all_samples = [], []
        num_generated = 0
        while num_generated < num_samples:
            sample_fn = self.sample_ddim if ddim else self.sample  # GMD:1452-1455
            tb = None
            if _targets is not None:
                tb = _targets[num_generated:num_generated + batch_size].long()
                if tb.shape[0] < batch_size:
                    tb = torch.cat([tb, tb[-1:].expand(batch_size - tb.shape[0])])
            sample, outputs = sample_fn(batch_size, target_dist, model_kwargs=model_kwargs, conditioning_function=conditioning_function,
                                        row0=row0 + num_generated, seed=seed, _targets=tb)
            mask_nan = torch.any(sample.isnan(), dim=1)
            sample = sample[~mask_nan]
            outputs["y"] = outputs["y"].to(sample.device)[~mask_nan]
            all_samples.append(sample)
            all_targets.append(outputs["y"])
            if sample.shape[0] != batch_size:
                raise FoundNaNsError
            num_generated += sample.shape[0]
        if not all_samples:  # num_samples == 0 (an empty shard): the reference's torch.cat([]) would raise
            dev = self.alphas.device if getattr(self, "results_on_device", False) else torch.device("cpu")
            width = self.num_numerical_features + (len(self.num_classes) if self._has_cat else 0)
            return (torch.empty((0, width), dtype=torch.float32 if ddim else torch.float64, device=dev),
                    torch.empty((0,), dtype=torch.long, device=dev))
        return torch.cat(all_samples, dim=0)[:num_samples], torch.cat(all_targets, dim=0)[:num_samples]
