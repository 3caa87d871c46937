"""Drop-in for the reference's ``ClavaDDPMTrainer`` (models/clavaddpm/trainer.py:17-155).

Same constructor, attributes (``loss_history``, ``ema_model``, ``log_every`` ...) and ``train()`` loop.  One
iteration = ``mtk_train_step`` (noising -> denoiser forward -> loss -> backward, all HIP) [+ one RCCL all-reduce of
the gradient image when ``torch.distributed`` is initialised] + ``mtk_train_apply`` (AdamW + EMA).  The weights,
gradients, Adam moments and the EMA live in one ``[5, n]`` device tensor (``self.state``); the ``nn.Parameter``
tensors of the denoiser (and of ``ema_model``) are refreshed from it when ``train()`` returns or on
:meth:`sync_modules`.
"""
from __future__ import annotations

from collections.abc import Iterator
from copy import deepcopy

import numpy as np
import torch
from torch import Tensor, nn

from .diffusion import GaussianMultinomialDiffusion
from .distributed import BucketedGradientExchange, world
from .model import timestep_frequencies


class _AdamWState:
    """Holds what the reference reads/writes on ``torch.optim.AdamW``: ``param_groups[0]['lr']`` (TRAINER:67)."""

    def __init__(self, lr: float, weight_decay: float):
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": (0.9, 0.999), "eps": 1e-8}]
        self.steps_done = 0

    def zero_grad(self) -> None:  # the gradient image is zeroed inside mtk_train_step
        pass


class ClavaDDPMTrainer:
    def __init__(self, diffusion_model: GaussianMultinomialDiffusion, trainer_iterator: Iterator[list[Tensor]],
                 learning_rate: float, weight_decay: float, steps: int, device: torch.device | None = None):
        import pandas as pd

        self.diffusion_model = diffusion_model
        self.ema_model = deepcopy(self.diffusion_model._denoise_fn)  # TRAINER:40-42
        self.ema_model._engine, self.ema_model._synced_key, self.ema_model._engine_shared = None, None, False
        for param in self.ema_model.parameters():
            param.detach_()
        self.trainer_iterator = trainer_iterator
        self.steps = steps
        self.initial_learning_rate = learning_rate
        self.optimizer = _AdamWState(learning_rate, weight_decay)
        self.device = torch.device(device) if device is not None else diffusion_model.alphas.device
        self.loss_history = pd.DataFrame(columns=["step", "mloss", "gloss", "loss"])
        self.log_every = 100
        self.print_every = 500
        self.ema_every = 1000
        self.ema_rate = 0.999
        self.seed = 0  # base seed of the in-kernel noise when diffusion_model.rng == "philox"

        net = diffusion_model._denoise_fn
        self._label = bool(getattr(net, "label_conditioned", False))
        if self._label and getattr(net, "num_classes", 0) == 0:
            raise NotImplementedError("ClavaDDPMTrainer: IsTargetConditioned.EMBEDDING with num_classes == 0 (a Linear on the raw regression "
                                      "target, MODEL:587) is not on the MI355X path")
        if getattr(net, "precision", "fp32") != "fp32" or diffusion_model.precision != "fp32":
            raise ValueError("training runs on the fp32 MFMA path: set model.precision = diffusion.precision = 'fp32'")
        self._engine = diffusion_model.engine()
        if getattr(self._engine, "_train_state", None) is not None:
            # the autograd seam of mixed_loss (or an earlier trainer) left a training image on this context: write its weights
            # back into the module and re-pack, so that this trainer starts from them with fresh optimizer state
            self._engine.train_export("weights", self._tensors(net))
            net._synced_key = None
            self._engine = diffusion_model.engine()
        if self._label:
            # MODEL:583-585, 610-612.  For inference such a net runs on its own context (label, timestep) table rows; the training
            # iteration runs on the diffusion object's context, which so far holds no denoiser: give it the layers, then an image
            # with the label embedding as one more layer, filled from the module
            self._engine.set_weights(net._weights(), net._dims(), timestep_frequencies(net.timestep_dimension))
            self._engine.train_set_label_classes(int(net.num_classes))
        self.state = self._engine.train_attach(timestep_frequencies(net.timestep_dimension))
        if self._label:
            self._engine.train_import("weights", self._tensors(net))
            self._engine.train_import("ema", self._tensors(net))
        self._engine._image_owner = "trainer"  # image 0 is the live model from here on: the autograd seam must not overwrite it
        self._engine.train_set_gaussian_loss(getattr(diffusion_model.gaussian_loss_type, "value", "mse"))  # GMD:534
        self._rank, self._world = world()
        # data parallel: one all-reduce per gradient bucket, issued under the backward (distributed.BucketedGradientExchange)
        self._exchange = BucketedGradientExchange(self._engine, self.state[1]) if self._world > 1 else None

    def use_exchange(self, overlapped: bool | None = None) -> None:
        """(Re)build the gradient exchange: ``overlapped=True`` runs the side-stream, per-bucket RCCL all-reduce even at world
        size 1 (how the path is rehearsed on a one-GPU box); ``False`` is the blocking per-bucket form; ``None`` the default."""
        if self._exchange is not None:
            self._exchange.close()
        self._exchange = BucketedGradientExchange(self._engine, self.state[1], overlapped=overlapped)

    # ------------------------------------------------------------------ TRAINER:58
    def _anneal_learning_rate(self, step: int) -> None:
        frac_done = step / self.steps
        lr = self.initial_learning_rate * (1 - frac_done)
        for param_group in self.optimizer.param_groups:
            param_group["lr"] = lr

    # ------------------------------------------------------------------ TRAINER:70
    def _train_step(self, input_tensor: Tensor, target: Tensor | None = None, injected=None) -> tuple[Tensor, Tensor]:
        """``injected = (t, eps, u)`` replays given draws (parity tests); otherwise they are drawn per
        ``diffusion_model.rng`` in the reference's order."""
        d, eng = self.diffusion_model, self._engine
        x0 = input_tensor.to(self.device).float().contiguous()
        B = x0.shape[0]
        seed = 0
        if injected is not None:
            t, eps, u = injected
        elif d.rng == "torch":
            t, eps, u = d._draw_loss_noise(x0)
        else:
            # the library draws the timesteps with the noise (uniform over [0, T): GMD:1011 sample_time(UNIFORM)) and takes the two
            # batch means on the iteration's side stream: no launch of the host framework is left in the iteration
            t = eps = u = None
            seed = (self.seed + 0x9E3779B97F4A7C15 * (self.optimizer.steps_done + 1)) % (1 << 63)
        if self._label:  # GMD:1042: the denoiser takes the batch's labels (TRAINER:88 hands mixed_loss the iterator's target dict)
            y = target.get("y") if isinstance(target, dict) else target
            if y is None:
                raise ValueError("ClavaDDPMTrainer: a label-conditioned denoiser needs the batch's labels ({'y': ...})")
            eng.train_set_labels(y.to(x0.device).long().reshape(-1).contiguous())
        eng.set_math(d.math or ("fast" if (d.rng == "philox" and injected is None) else "reference"))
        p_drop = d._denoise_fn.training_dropout() if hasattr(d._denoise_fn, "training_dropout") else 0.0
        # MODEL:162-180: a fresh mask per iteration, whatever the noise source (its key follows the step count like the Philox noise's)
        eng.train_set_dropout(p_drop, (self.seed + 0xD1B54A32D192ED03 * (self.optimizer.steps_done + 1)) % (1 << 63) if p_drop else 0)
        t = t.to(x0.device, torch.int64).contiguous() if t is not None else None
        eps = eps.to(x0.device).float().contiguous() if eps is not None and eps.numel() else None
        u = u.to(x0.device).float().contiguous() if u is not None and u.numel() else None
        # per-record losses (one buffer for both) and their batch means, the latter taken by the library inside the iteration
        rows = torch.empty((2, B), dtype=torch.float32, device=x0.device) if (eng.C > 0 and eng.Nn > 0) else None
        lm = rows[0] if rows is not None else (torch.empty((B,), dtype=torch.float32, device=x0.device) if eng.C > 0 else None)
        lg = rows[1] if rows is not None else (torch.empty((B,), dtype=torch.float32, device=x0.device) if eng.Nn > 0 else None)
        means = torch.empty((2,), dtype=torch.float32, device=x0.device)  # a fresh pair per iteration: the caller may keep what is returned
        eng.train_set_loss_means(means)
        # one device: nothing reads the gradient image between the backward and the optimizer pass, which then adds the split-batch
        # partials itself (same order, same bits; one launch fewer).  Switched per iteration: direct engine calls keep the default.
        defer = self._exchange is None
        if defer:
            eng.train_set_deferred_reduce(True)
        try:
            eng.train_step(x0, t, eps, u, 1.0 / (B * self._world), lm, lg, seed=seed, row0=self._rank * B)
            eng.train_set_loss_means(None)
            if self._exchange is not None:  # the one exchange of an iteration, bucket by bucket under the backward
                self._exchange.launch()
                self._exchange.wait()
            g = self.optimizer.param_groups[0]
            self.optimizer.steps_done += 1
            eng.train_apply(g["lr"], g["weight_decay"], self.optimizer.steps_done, g["betas"], g["eps"], self.ema_rate)
        finally:
            if defer:  # also after an error: a later direct engine call must find the default mode (and a complete gradient image)
                eng.train_set_deferred_reduce(False)
        return means[0], means[1]

    # ------------------------------------------------------------------ TRAINER:96
    def train(self) -> None:
        step = 0
        dev = self.device
        curr_loss_multi = torch.zeros((), dtype=torch.float64, device=dev)
        curr_loss_gauss = torch.zeros((), dtype=torch.float64, device=dev)
        curr_count = 0
        while step < self.steps:
            input_tensor, output = next(self.trainer_iterator)
            batch_loss_multi, batch_loss_gauss = self._train_step(input_tensor, output)
            self._anneal_learning_rate(step)
            n = len(input_tensor)
            curr_count += n
            curr_loss_multi += batch_loss_multi.double() * n  # stays on the device: no per-step sync
            curr_loss_gauss += batch_loss_gauss.double() * n
            if (step + 1) % self.log_every == 0:
                mloss = np.around(curr_loss_multi.item() / curr_count, 4)
                gloss = np.around(curr_loss_gauss.item() / curr_count, 4)
                if (step + 1) % self.print_every == 0:
                    print(f"Step {step + 1}/{self.steps} MLoss: {mloss} GLoss: {gloss} Sum: {mloss + gloss}", flush=True)
                self.loss_history.loc[len(self.loss_history)] = [step + 1, mloss, gloss, mloss + gloss]
                curr_count = 0
                curr_loss_gauss.zero_()
                curr_loss_multi.zero_()
            step += 1
        self.sync_modules()

    # ------------------------------------------------------------------ state <-> modules
    @staticmethod
    def _tensors(net: nn.Module):
        """(W, b) per layer in the training image's order; a label-conditioned net carries its embedding table after the time-embedding
        MLP's two layers (mtk_train_set_label_classes), with an unused bias slot."""
        out = [(W.data, b.data) for W, b in net._weights()]
        if getattr(net, "label_conditioned", False) and getattr(net, "num_classes", 0) > 0:
            w = net.label_embedding.weight.data
            if getattr(net, "_label_bias_slot", None) is None or net._label_bias_slot.device != w.device:
                net._label_bias_slot = torch.zeros((w.shape[0],), dtype=torch.float32, device=w.device)
            out.insert(3, (w, net._label_bias_slot))
        return out

    def sync_modules(self) -> None:
        """Write the trained weights into ``diffusion_model._denoise_fn`` and the EMA into ``ema_model``."""
        net = self.diffusion_model._denoise_fn
        self.ema_model.to(self.device)
        self._engine.train_export("weights", self._tensors(net))
        self._engine.train_export("ema", self._tensors(self.ema_model))
        # the export writes the parameters through raw pointers (no version bump): the EMA copy's own packed image is stale, and so
        # is the table-row context a label-conditioned net keeps for inference
        self.ema_model._synced_key = None
        if self._label:
            net._synced_key = None
        torch.cuda.synchronize(self.device)

    def set_train_precision(self, precision: str) -> None:
        """Not in the reference (it trains in fp32): "bf16" runs the iteration's GEMMs on bf16 MFMA with fp32 accumulation, fp32
        master weights / AdamW / EMA (mtk_train_set_precision; tolerances in tests/test_gpu_train.py); "fp32" is the parity path."""
        self._engine.train_set_precision(precision)

    def finish(self) -> None:
        """Detach the training state: the denoiser is re-packed (fp32 + bf16 images) on its next use."""
        self.sync_modules()
        self.diffusion_model._denoise_fn._synced_key = None
        if self._exchange is not None:
            self._exchange.close()


def update_ema(target_params: Iterator[nn.Parameter], source_params: Iterator[nn.Parameter], rate: float = 0.999) -> None:
    """TRAINER:138 for callers that hold module parameters (the trainer itself fuses this into mtk_train_apply)."""
    with torch.no_grad():
        for targ, src in zip(target_params, source_params):
            targ.detach().mul_(rate).add_(src.detach(), alpha=1 - rate)
