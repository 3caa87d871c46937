"""Tartan-Federer denoising-loss scoring on MI355X.

``mixed_loss`` keeps the reference's signature and semantics (TF:40-95) so ``get_score`` (TF:259)
can call it unchanged; ``score_records`` is the device-resident driver for many timesteps at once
(SURVEY section 8f rank 1): records stay on the GPU, all (timestep, noise) pairs run in one call.
"""
from __future__ import annotations

import torch
from torch import Tensor

from .diffusion import GaussianMultinomialDiffusion


@torch.no_grad()
def score_records(diffusion_model: GaussianMultinomialDiffusion, features: Tensor, noise: Tensor, timesteps,
                  additional_timestep: int = 0, precision: str | None = None) -> Tensor:
    """[n_t, B, P] scores: mean_d (noise_p - eps_hat(x_t(record, noise_p, t + add), t))^2."""
    if not features.is_cuda:
        raise RuntimeError("score_records: midst_toolkit_amd has no CPU path; move the records to the MI355X")
    if diffusion_model._label_net():
        raise NotImplementedError("Tartan-Federer scoring with a label-conditioned denoiser (IsTargetConditioned.EMBEDDING) is not on the MI355X path")
    eng = diffusion_model.engine()
    x = features[:, : diffusion_model.num_numerical_features].float()
    if x.stride(1) != 1:
        x = x.contiguous()
    nz = noise.to(x.device, torch.float32).contiguous()
    return eng.tf_score(x, nz, list(timesteps), additional_timestep, precision or diffusion_model.precision)


def mixed_loss(diffusion_model: GaussianMultinomialDiffusion, features: Tensor, outputs: dict[str, Tensor],
               noise: list[list[float]], parallel_batch: int, additional_timestep: int, timestep: int) -> Tensor:
    """TF:40.  Returns [B, parallel_batch]; row r*P + p of the reference's batch is out[r, p]."""
    noise_tensor = torch.tensor(noise, device=features.device, dtype=torch.float)
    assert noise_tensor.shape[0] == parallel_batch
    return score_records(diffusion_model, features, noise_tensor, [timestep], additional_timestep)[0]
