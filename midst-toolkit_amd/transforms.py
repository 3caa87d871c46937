"""Device-side inverse of the pipeline's numeric normalisation.

The reference un-normalises every synthesised table on the host with the fitted scikit-learn object
(``dataset.numerical_transform.inverse_transform``, synthesizer.py:251-277).  ``QuantileInverse`` takes that fitted
``QuantileTransformer`` (only its ``quantiles_``, ``references_`` and ``output_distribution`` are read), keeps the two
tables on the GPU and applies the same map to a device tensor — the sampled rows never have to visit the host first."""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .engine import check, ptr


class QuantileInverse:
    def __init__(self, quantile_transformer, device: torch.device | str):
        dist = getattr(quantile_transformer, "output_distribution", "normal")
        if dist not in ("normal", "uniform"):
            raise ValueError(f"Unsupported output distribution: {dist}")
        self.device = torch.device(device)
        self.normal = int(dist == "normal")
        q = np.ascontiguousarray(np.asarray(quantile_transformer.quantiles_, dtype=np.float64))
        r = np.ascontiguousarray(np.asarray(quantile_transformer.references_, dtype=np.float64))
        if q.ndim != 2 or r.shape != (q.shape[0],):
            raise ValueError(f"quantiles_ {q.shape} / references_ {r.shape} do not look like a fitted QuantileTransformer")
        self.n_quantiles, self.n_columns = q.shape
        self.quantiles = torch.from_numpy(q).to(self.device)
        self.references = torch.from_numpy(r).to(self.device)

    def inverse_transform(self, x: Tensor) -> Tensor:
        """[B, n_columns] device tensor (any float dtype, any row stride) -> new fp64 [B, n_columns] device tensor."""
        if x.dim() != 2 or x.shape[1] != self.n_columns:
            raise ValueError(f"expected [B, {self.n_columns}], got {tuple(x.shape)}")
        if x.device != self.device:
            raise ValueError(f"input on {x.device}, tables on {self.device}")
        x = x.double()
        if x.stride(1) != 1:
            x = x.contiguous()
        out = torch.empty((x.shape[0], self.n_columns), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_quantile_inverse(self.device.index or 0, x.shape[0], self.n_columns, ptr(x), x.stride(0) if x.shape[0] > 1 else self.n_columns,
                                                  ptr(self.quantiles), ptr(self.references), self.n_quantiles, self.normal, ptr(out),
                                                  self.n_columns, torch.cuda.current_stream(self.device).cuda_stream))
        return out


class AffineInverse:
    """The same for the reference's other two normalisers (common/dataset_transformations.py:66-69): a fitted
    ``StandardScaler`` (``scale_``, ``mean_``) or ``MinMaxScaler`` (``min_``, ``scale_``)."""

    def __init__(self, scaler, device: torch.device | str):
        self.device = torch.device(device)
        if hasattr(scaler, "mean_") and hasattr(scaler, "scale_") and not hasattr(scaler, "min_"):
            self.kind = 0
            scale = scaler.scale_ if getattr(scaler, "with_std", True) and scaler.scale_ is not None else np.ones_like(scaler.mean_)
            mean = scaler.mean_ if getattr(scaler, "with_mean", True) and scaler.mean_ is not None else np.zeros_like(scale)
            p0, p1 = scale, mean
        elif hasattr(scaler, "min_") and hasattr(scaler, "scale_"):
            self.kind = 1
            p0, p1 = scaler.min_, scaler.scale_
        else:
            raise ValueError(f"Unsupported normalizer: {type(scaler).__name__}")
        self.p0 = torch.from_numpy(np.ascontiguousarray(np.asarray(p0, dtype=np.float64))).to(self.device)
        self.p1 = torch.from_numpy(np.ascontiguousarray(np.asarray(p1, dtype=np.float64))).to(self.device)
        self.n_columns = int(self.p0.shape[0])

    def inverse_transform(self, x: Tensor) -> Tensor:
        if x.dim() != 2 or x.shape[1] != self.n_columns:
            raise ValueError(f"expected [B, {self.n_columns}], got {tuple(x.shape)}")
        if x.device != self.device:
            raise ValueError(f"input on {x.device}, parameters on {self.device}")
        x = x.double()
        if x.stride(1) != 1:
            x = x.contiguous()
        out = torch.empty((x.shape[0], self.n_columns), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().mtk_affine_inverse(self.device.index or 0, x.shape[0], self.n_columns, ptr(x), x.stride(0) if x.shape[0] > 1 else self.n_columns,
                                                self.kind, ptr(self.p0), ptr(self.p1), ptr(out), self.n_columns,
                                                torch.cuda.current_stream(self.device).cuda_stream))
        return out


def inverse_transform_for(normalizer, device) -> "QuantileInverse | AffineInverse":
    """The device-side inverse of whichever normaliser ``normalize`` fitted (dataset.numerical_transform)."""
    if hasattr(normalizer, "quantiles_"):
        return QuantileInverse(normalizer, device)
    return AffineInverse(normalizer, device)
