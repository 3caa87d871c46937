"""Drop-in for the reference's guidance classifier (models/clavaddpm/model.py:45-120) and its conditioning function
(models/clavaddpm/synthesizer.py:429-451) on MI355X.

``Classifier`` has the reference's constructor, parameter names and ``forward(input_tensor, timesteps)``; a state dict of
the reference's trained classifier loads into it (its ``transformer_layer.*`` entries - an ``nn.Transformer`` the
reference constructs at MODEL:73 and never calls - are accepted and dropped).  In eval mode, which is how the synthesizer
uses it (TRAIN:447), the forward pass and the input gradient of ``log_softmax(logits)[y]`` run in libmtk_hip.so
(``mtk_clf_guidance``: fp32 MFMA GEMMs forward, a hand-written backward; no autograd).  In training mode
(``classifier.train()``, models/clavaddpm/train.py:409-425) ``forward`` runs BatchNorm on batch statistics and Dropout in the
library too (``mtk_clft_forward``), updates the running statistics as torch does, and - with grad enabled - returns logits that
carry an autograd graph whose backward is the library's hand-written chain (``mtk_clft_backward``): the reference's
``cross_entropy(logits, y).backward(); optimizer.step()`` loop runs against the drop-in.
"""
from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np
import torch
from torch import Tensor, nn

from . import _lib
from ._lib import check, ptr
from .model import timestep_frequencies


class Classifier(nn.Module):
    def __init__(self, input_dimension: int, output_dimension: int, timestep_dimension: int, hidden_sizes: list[int],
                 dropout_prob: float = 0.5, num_heads: int = 2, num_layers: int = 1, num_timesteps: int = 1000):
        """MODEL:46-101.  ``num_timesteps`` (not in the reference): size of the time-embedding table."""
        super().__init__()
        self.timestep_dimension = timestep_dimension
        self.input_dimension, self.output_dimension = input_dimension, output_dimension
        self.num_timesteps = num_timesteps
        self.proj = nn.Linear(input_dimension, timestep_dimension)
        self.timestep_embedding = nn.Sequential(nn.Linear(timestep_dimension, timestep_dimension), nn.SiLU(),
                                                nn.Linear(timestep_dimension, timestep_dimension))
        layers: list[nn.Module] = [nn.Linear(timestep_dimension, hidden_sizes[0]), nn.ReLU(), nn.BatchNorm1d(hidden_sizes[0]), nn.Dropout(p=dropout_prob)]
        for i in range(len(hidden_sizes) - 1):
            layers += [nn.Linear(hidden_sizes[i], hidden_sizes[i + 1]), nn.LeakyReLU(), nn.BatchNorm1d(hidden_sizes[i + 1]), nn.Dropout(p=dropout_prob)]
        layers.append(nn.Linear(hidden_sizes[-1], output_dimension))
        self.model = nn.Sequential(*layers)
        self._handle = None
        self._handle_key = None
        # "fp32" (the reference's arithmetic) or "bf16x3": the guidance GEMMs as split-bf16 products (mtk_clf_set_precision; ~1e-3 of
        # the gradient's scale at several times the fp32 rate) - for long guided chains; not in the reference
        self.precision = "fp32"
        self._train_handle = None   # mtk_clft: training-mode forward / backward
        self._train_key = None
        self.dropout_seed = 0       # base key of the Philox dropout masks (a fresh mask per training forward)
        self._train_calls = 0

    # ------------------------------------------------------------------ state
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Accepts the reference classifier's state dict: the never-called nn.Transformer's entries are dropped."""
        kept = {k: v for k, v in state_dict.items() if not k.startswith("transformer_layer.")}
        return super().load_state_dict(kept, strict=strict, assign=assign)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_handle"], state["_handle_key"] = None, None
        state["_train_handle"], state["_train_key"] = None, None
        return state

    def __del__(self):
        try:
            h = self.__dict__.get("_handle")
            self.__dict__["_handle"] = None
            if h:
                _lib.lib().mtk_clf_destroy(h)
            h = self.__dict__.get("_train_handle")
            self.__dict__["_train_handle"] = None
            if h:
                _lib.lib().mtk_clft_destroy(h)
        except Exception:
            pass

    # ------------------------------------------------------------------ training mode (MODEL:45-120 in train(), TRAIN:409-425)
    def _stages(self):
        """([Linear...] incl. the output layer, [BatchNorm1d...], [activation slope...], dropout p) of ``model``."""
        lins = [m for m in self.model if isinstance(m, nn.Linear)]
        bns = [m for m in self.model if isinstance(m, nn.BatchNorm1d)]
        slopes = [0.0 if isinstance(m, nn.ReLU) else float(m.negative_slope) for m in self.model if isinstance(m, (nn.ReLU, nn.LeakyReLU))]
        drops = [float(m.p) for m in self.model if isinstance(m, nn.Dropout)]
        assert len(lins) == len(bns) + 1 == len(slopes) + 1 and len(set(drops)) <= 1 and len({float(m.eps) for m in bns}) == 1
        return lins, bns, slopes, (drops[0] if drops else 0.0)

    def _train_layers(self):
        lins, bns, _, _ = self._stages()
        return [self.proj, self.timestep_embedding[0], self.timestep_embedding[2]] + lins, bns

    def _train_engine(self, device: torch.device):
        lins, bns, slopes, _ = self._stages()
        if self._train_handle is None or self._train_key != str(device):
            if self._train_handle is not None:
                _lib.lib().mtk_clft_destroy(self._train_handle)
                self._train_handle = None
            widths = np.ascontiguousarray(np.asarray([int(m.out_features) for m in lins], dtype=np.int32))
            sl = np.ascontiguousarray(np.asarray(slopes, dtype=np.float32))
            fr = np.ascontiguousarray(timestep_frequencies(self.timestep_dimension).numpy().astype(np.float32))
            handle = C.c_void_p()
            index = device.index if device.index is not None else torch.cuda.current_device()
            with torch.cuda.device(device):
                check(_lib.lib().mtk_clft_create(index, self.input_dimension, self.timestep_dimension, len(bns), widths.ctypes.data, sl.ctypes.data,
                                                 fr.ctypes.data, _lib.stream_handle(device), C.byref(handle)))
            self._train_handle, self._train_key = handle, str(device)
        return self._train_handle

    def dropout_masks(self, B: int) -> list[Tensor]:
        """uint8 [B, width] keep masks of every Dropout stage in the last training-mode forward (mtk_clft_dropout_mask)."""
        lins, bns, _, _ = self._stages()
        dev = torch.device(self._train_key)
        out = [torch.empty((B, int(l.out_features)), dtype=torch.uint8, device=dev) for l in lins[:-1]]
        with torch.cuda.device(dev):
            for i, m in enumerate(out):
                check(_lib.lib().mtk_clft_dropout_mask(self._train_handle, i, B, ptr(m), _lib.stream_handle(dev)))
        return out

    def _forward_train(self, x: Tensor, timesteps: Tensor) -> Tensor:
        params = [p for l in self._train_layers()[0] for p in (l.weight, l.bias)] + [p for bn in self._train_layers()[1] for p in (bn.weight, bn.bias)]
        return _ClassifierTrainFunction.apply(self, x.detach().float().contiguous(), timesteps.to(x.device, torch.int64).contiguous(), *params)

    # ------------------------------------------------------------------ the network as the library takes it
    def _folded_layers(self):
        """[(W, b)] for proj, timestep_embedding.0, timestep_embedding.2 and the Linear layers of ``model`` with every
        eval-mode BatchNorm1d (y * s + o per feature, s = gamma / sqrt(var + eps), o = beta - mean * s) folded into the
        Linear that follows it: W' = W diag(s), b' = b + W o.  Plus the activation slope after each hidden Linear."""
        out = [(self.proj.weight, self.proj.bias), (self.timestep_embedding[0].weight, self.timestep_embedding[0].bias),
               (self.timestep_embedding[2].weight, self.timestep_embedding[2].bias)]
        slopes, widths = [], []
        scale = offset = None
        for m in self.model:
            if isinstance(m, nn.Linear):
                W, b = m.weight.detach().double(), m.bias.detach().double()
                if scale is not None:
                    b = b + W @ offset
                    W = W * scale[None, :]
                    scale = offset = None
                out.append((W.float().contiguous(), b.float().contiguous()))
                widths.append(int(m.out_features))
            elif isinstance(m, nn.ReLU):
                slopes.append(0.0)
            elif isinstance(m, nn.LeakyReLU):
                slopes.append(float(m.negative_slope))
            elif isinstance(m, nn.BatchNorm1d):
                s = m.weight.detach().double() / torch.sqrt(m.running_var.detach().double() + m.eps)
                scale, offset = s, m.bias.detach().double() - m.running_mean.detach().double() * s
        return out, slopes, widths

    def _engine(self, device: torch.device):
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._handle is not None and key == self._handle_key:
            return self._handle
        if self._handle is not None:
            _lib.lib().mtk_clf_destroy(self._handle)
            self._handle = None
        layers, slopes, widths = self._folded_layers()
        keep = [(W.detach().float().contiguous().to(device), b.detach().float().contiguous().to(device)) for W, b in layers]
        Wp = (C.c_void_p * len(keep))(*[W.data_ptr() for W, _ in keep])
        bp = (C.c_void_p * len(keep))(*[b.data_ptr() for _, b in keep])
        w32 = np.ascontiguousarray(np.asarray(widths, dtype=np.int32))
        sl = np.ascontiguousarray(np.asarray(slopes + [0.0], dtype=np.float32))
        fr = np.ascontiguousarray(timestep_frequencies(self.timestep_dimension).numpy().astype(np.float32))
        handle = C.c_void_p()
        index = device.index if device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(device):
            check(_lib.lib().mtk_clf_create(index, self.num_timesteps, self.input_dimension, self.timestep_dimension, len(widths),
                                            w32.ctypes.data, sl.ctypes.data, Wp, bp, fr.ctypes.data, _lib.stream_handle(device), C.byref(handle)))
        self._handle, self._handle_key = handle, key
        return handle

    def _run(self, x: Tensor, timesteps: Tensor, y: Tensor | None, scale: float, want_logits: bool, want_grad: bool):
        if not x.is_cuda:
            raise RuntimeError("Classifier: midst_toolkit_amd has no CPU path; move the module and inputs to the MI355X")
        if self.training:
            raise NotImplementedError("Classifier.guidance_gradient is the eval-mode network's input gradient (BatchNorm running statistics, no "
                                      "Dropout), which is how the synthesizer calls it (TRAIN:447): call classifier.eval() first")
        dev = x.device
        h = self._engine(dev)
        precision = getattr(self, "precision", "fp32")  # objects pickled before the attribute existed
        if precision not in ("fp32", "bf16x3"):
            raise ValueError(f"Classifier.precision: {precision!r} (fp32 or bf16x3)")
        check(_lib.lib().mtk_clf_set_precision(h, _lib.PREC_BF16X3 if precision == "bf16x3" else _lib.PREC_FP32))
        xd = x.detach()
        if xd.dtype == torch.float32 and xd.is_contiguous():
            xf = xd
        else:  # one kernel also for a transposed fp64 view (the chain's state): conversion and layout in the same copy
            xf = torch.empty(tuple(xd.shape), dtype=torch.float32, device=dev)
            xf.copy_(xd)
        B = xf.shape[0]
        t = timesteps.to(dev, torch.int64).contiguous()
        yy = y.to(dev, torch.int64).view(-1).contiguous() if y is not None else None
        logits = torch.empty((B, self.output_dimension), dtype=torch.float32, device=dev) if want_logits else None
        grad = torch.empty((B, self.input_dimension), dtype=torch.float32, device=dev) if want_grad else None
        with torch.cuda.device(dev):
            check(_lib.lib().mtk_clf_guidance(h, B, ptr(xf), xf.stride(0), ptr(t), ptr(yy), float(scale), ptr(logits), ptr(grad), _lib.stream_handle(dev)))
        return logits, grad

    def forward(self, input_tensor: Tensor, timesteps: Tensor) -> Tensor:
        """MODEL:103-120: logits [B, output_dimension].  Eval mode: running statistics, no Dropout, a value.  Training mode: batch
        statistics, Dropout, running statistics updated, and - with grad enabled - a graph over the parameters."""
        if self.training:
            if not input_tensor.is_cuda:
                raise RuntimeError("Classifier: midst_toolkit_amd has no CPU path; move the module and inputs to the MI355X")
            return self._forward_train(input_tensor, timesteps)
        if torch.is_grad_enabled() and input_tensor.requires_grad:
            raise NotImplementedError("Classifier.forward builds no autograd graph; use guidance_gradient / get_conditioning_function for "
                                      "the input gradient (hand-written backward in libmtk_hip.so)")
        return self._run(input_tensor, timesteps, None, 1.0, True, False)[0]

    def guidance_gradient(self, features: Tensor, timesteps: Tensor, target: Tensor, scale: float = 1.0) -> Tensor:
        """SYNTH:443-449: ``scale * d log_softmax(classifier(x, t))[range(B), target].sum() / d x`` ([B, input_dimension] fp32)."""
        return self._run(features, timesteps, target, scale, False, True)[1]


class _ClassifierTrainFunction(torch.autograd.Function):
    """Training-mode forward (mtk_clft_forward) with a graph over the classifier's parameters; backward = mtk_clft_backward."""

    @staticmethod
    def forward(ctx, clf: Classifier, x: Tensor, t: Tensor, *params):
        dev = x.device
        h = clf._train_engine(dev)
        layers, bns = clf._train_layers()
        _, _, _, p_drop = clf._stages()
        keep = [(l.weight.detach().float().contiguous(), l.bias.detach().float().contiguous()) for l in layers]
        kbn = [(bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()) for bn in bns]
        arr = lambda ts: (C.c_void_p * len(ts))(*[v.data_ptr() for v in ts])  # noqa: E731
        B = x.shape[0]
        logits = torch.empty((B, clf.output_dimension), dtype=torch.float32, device=dev)
        means = [torch.empty((bn.num_features,), dtype=torch.float32, device=dev) for bn in bns]
        vars_ = [torch.empty((bn.num_features,), dtype=torch.float32, device=dev) for bn in bns]
        clf._train_calls += 1
        seed = (int(clf.dropout_seed) + 0x9E3779B97F4A7C15 * clf._train_calls) % (1 << 63)
        with torch.cuda.device(dev):
            st = _lib.stream_handle(dev)
            check(_lib.lib().mtk_clft_set_params(h, arr([w for w, _ in keep]), arr([b for _, b in keep]), arr([g for g, _ in kbn]), arr([b for _, b in kbn]), st))
            check(_lib.lib().mtk_clft_forward(h, B, ptr(x), x.stride(0), ptr(t), float(p_drop), seed, float(bns[0].eps), ptr(logits), arr(means), arr(vars_), st))
        with torch.no_grad():  # nn.BatchNorm1d's bookkeeping in training mode: momentum on the mean and the UNBIASED variance
            for bn, m, v in zip(bns, means, vars_):
                if bn.track_running_stats and bn.running_mean is not None:
                    bn.num_batches_tracked += 1
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1 - mom).add_(m, alpha=mom)
                    bn.running_var.mul_(1 - mom).add_(v * (B / (B - 1)), alpha=mom)
        ctx.clf, ctx.h, ctx.B, ctx.token = clf, h, B, clf._train_calls
        ctx.needs = [p.requires_grad for p in params]
        ctx.seed = seed
        return logits

    @staticmethod
    def backward(ctx, g):
        clf = ctx.clf
        if ctx.token != clf._train_calls or ctx.h is not clf._train_handle:
            raise RuntimeError("backward through Classifier logits after another training-mode forward on the same module: the library "
                               "keeps the activations of one forward pass at a time")
        layers, bns = clf._train_layers()
        g = g.to(torch.float32).contiguous()
        gw = [(torch.empty_like(l.weight.data, dtype=torch.float32), torch.empty_like(l.bias.data, dtype=torch.float32)) for l in layers]
        gbn = [(torch.empty_like(bn.weight.data, dtype=torch.float32), torch.empty_like(bn.bias.data, dtype=torch.float32)) for bn in bns]
        arr = lambda ts: (C.c_void_p * len(ts))(*[v.data_ptr() for v in ts])  # noqa: E731
        with torch.cuda.device(g.device):
            check(_lib.lib().mtk_clft_backward(ctx.h, ctx.B, ptr(g), g.stride(0), arr([w for w, _ in gw]), arr([b for _, b in gw]), arr([a for a, _ in gbn]),
                                               arr([b for _, b in gbn]), _lib.stream_handle(g.device)))
        flat = [t for pair in gw for t in pair] + [t for pair in gbn for t in pair]
        return (None, None, None) + tuple(t if need else None for t, need in zip(flat, ctx.needs))


def get_conditioning_function(classifier: Classifier, classifier_scale: float):
    """Drop-in for ``_get_conditioning_function`` (SYNTH:429-451): the callable ``conditional_sample`` / ``sample`` take."""

    def conditioning_function(features: Tensor, timestep: Tensor, **kwargs: Any) -> Tensor:
        assert "y" in kwargs and kwargs["y"] is not None, "The kwargs parameter `y` must be provided."
        assert isinstance(kwargs["y"], Tensor), "The kwargs parameter `y` must be a Tensor."
        x = features[:, 1:] if kwargs.get("remove_first_col", False) else features
        return classifier.guidance_gradient(x, timestep, kwargs["y"], classifier_scale)

    conditioning_function._mtk_stream_safe = True  # no random draws, all work on the current stream: the chain may run it on a side stream
    return conditioning_function
