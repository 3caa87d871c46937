"""Device-resident feature driver for the Tartan-Federer attack.

Replaces the outer loops of the reference that surround its scoring call:
  * ``get_score`` (TF:259-341), which re-reads the CSV and unpickles the checkpoint for EVERY (model, timestep,
    additional timestep) triple and scores one batch per call;
  * the models x timesteps x additional_timesteps nest of ``train_tartan_federer_attack_classifier`` (TF:507-594) that fills
    ``x_train`` / ``x_val`` slice by slice through host numpy;
  * the challenge-point scoring nest of ``tartan_federer_attack`` (TF:695-717) with its min-max normalisation (TF:718-724).

Here every model is ingested once (``from_reference_state`` or a drop-in object) and stays on the GPU, the records stay on
the GPU, all timesteps x noise draws of one additional timestep run in ONE ``mtk_tf_score`` call, and the feature matrix
``[N, n_t * n_add * P]`` is assembled on the device in the reference's column order (column = (i_t * n_add + i_add) * P + p,
TF:531-548) - ready for ``classification.fit_model`` (classification.py:120), which takes arrays or tensors.  No numeric change:
the scores are those of ``scoring.mixed_loss`` (same kernels), so features are identical to calling it in the reference's loop.
"""
from __future__ import annotations

from typing import Any, Hashable, Iterable, Sequence

import torch
from torch import Tensor

from .diffusion import GaussianMultinomialDiffusion
from .scoring import score_records


class AttackFeatureDriver:
    def __init__(self, input_noise: Sequence[Sequence[float]] | Tensor, timesteps: Iterable[int], additional_timesteps: Iterable[int] = (0,),
                 precision: str | None = None, device: torch.device | None = None):
        """``input_noise``: the attack's P noise vectors (TF:488: ``num_noise_per_time_step`` draws of length
        ``num_numerical_features``), shared by every model, timestep and record."""
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.noise = torch.as_tensor(input_noise, dtype=torch.float32).to(self.device).contiguous()
        assert self.noise.ndim == 2, "input_noise is [P, num_numerical_features]"
        self.timesteps = [int(t) for t in timesteps]
        self.additional_timesteps = [int(a) for a in additional_timesteps]
        self.precision = precision
        self.models: dict[Hashable, GaussianMultinomialDiffusion] = {}

    @property
    def feature_dimension(self) -> int:
        """TF:489: len(input_noise) * len(timesteps) * len(additional_timesteps)."""
        return self.noise.shape[0] * len(self.timesteps) * len(self.additional_timesteps)

    # ------------------------------------------------------------------ models, resident
    def add_model(self, key: Hashable, model: GaussianMultinomialDiffusion | None = None, state_dict: dict[str, Tensor] | None = None,
                  ctor_args: dict[str, Any] | None = None) -> GaussianMultinomialDiffusion:
        """Once per shadow / target model instead of one unpickle per get_score call: a drop-in object, or the reference
        object's ``state_dict()`` + constructor arguments (no reference class needs to be importable)."""
        if model is None:
            model = GaussianMultinomialDiffusion.from_reference_state(state_dict, ctor_args, device=self.device)
        model = model.to(self.device)
        if model.num_numerical_features != self.noise.shape[1]:
            raise AssertionError(f"model {key!r} has {model.num_numerical_features} numeric features, the noise vectors {self.noise.shape[1]}")
        if self.precision is not None:
            model.precision = self.precision
        self.models[key] = model
        return model

    # ------------------------------------------------------------------ features
    @torch.no_grad()
    def features(self, key: Hashable, records: Tensor) -> Tensor:
        """[B, n_t * n_add * P] for the records of one model (every row, not only the first batch): what the
        reference accumulates into x_train[rows, timestep_count * P : (timestep_count + 1) * P] (TF:546-549)."""
        model = self.models[key]
        x = records.to(self.device)
        B, P, nt, na = x.shape[0], self.noise.shape[0], len(self.timesteps), len(self.additional_timesteps)
        out = torch.empty((B, nt, na, P), dtype=torch.float32, device=self.device)
        for ia, add in enumerate(self.additional_timesteps):
            s = score_records(model, x, self.noise, self.timesteps, add, precision=self.precision)  # [n_t, B, P]
            out[:, :, ia, :] = s.permute(1, 0, 2)
        return out.view(B, nt * na * P)

    def matrix(self, parts: Iterable[tuple[Hashable, Tensor, Tensor | None]]) -> tuple[Tensor, Tensor | None]:
        """(x, y) over several models, model-major like TF:540-559: ``parts`` yields (model key, records [B_i, .],
        labels [B_i] or None).  Device tensors; ``fit_model(regression_model, x, y, ...)`` takes them as they are."""
        xs, ys = [], []
        for key, records, labels in parts:
            xs.append(self.features(key, records))
            if labels is not None:
                ys.append(torch.as_tensor(labels, dtype=torch.float32).to(self.device))
        return torch.cat(xs, 0), (torch.cat(ys, 0) if ys else None)

    @torch.no_grad()
    def challenge_predictions(self, key: Hashable, records: Tensor, regression_model: torch.nn.Module) -> Tensor:
        """TF:695-724 for one model: features -> the fitted attack classifier -> min-max normalised to [0, 1]."""
        feats = self.features(key, records)
        pred = regression_model.to(self.device)(feats).detach().float()
        lo, hi = pred.min(), pred.max()
        return (pred - lo) / (hi - lo)
