"""Drop-in for the reference's denoiser classes (models/clavaddpm/model.py).

Same constructor arguments, parameter names and ``forward(x, timesteps, y=None)`` signature as
MODEL:147-303 (``MLP``), MODEL:548-615 (``MLPDiffusion``) and MODEL:780-811 (``ModelType``); the
forward pass runs in libmtk_hip.so (fp32-exact MFMA path or the fused bf16 MFMA kernel).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from enum import Enum

import torch
from torch import Tensor, nn

from . import schedule
from .engine import Engine


class IsTargetConditioned(Enum):
    """common/enumerations.py in the reference."""

    CONCAT = "concat"
    EMBEDDING = "embedding"
    NONE = "none"


@dataclass
class DiffusionParameters:
    """MODEL:19-32."""

    layers_dimensions: list[int]
    dropout: float
    input_dimension: int = 0
    output_dimension: int = 0
    embedding_dimension: int = 0
    n_blocks: int = 0
    block_dimension: int = 0
    hidden_dimension: int = 0
    dropout_first: float = 0
    dropout_second: float = 0


@dataclass
class ModelParameters:
    """MODEL:35-42."""

    diffusion_parameters: DiffusionParameters = field(default_factory=lambda: DiffusionParameters([], 0.0))
    input_dimension: int = 0
    num_classes: int = 0
    is_target_conditioned: IsTargetConditioned = IsTargetConditioned.NONE


def timestep_frequencies(output_dimension: int, max_period: int = 10000) -> Tensor:
    """The frequency vector of MODEL:136, computed with the same torch expression."""
    half = output_dimension // 2
    return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)


class MLP(nn.Module):
    """Parameter container with the reference's layout (MODEL:147-303): blocks[i].linear, head."""

    class Block(nn.Module):
        def __init__(self, *, input_dimension: int, output_dimension: int, bias: bool, dropout: float) -> None:
            super().__init__()
            self.linear = nn.Linear(input_dimension, output_dimension, bias)
            self.activation = nn.ReLU()
            self.dropout = nn.Dropout(dropout)

    def __init__(self, *, input_dimension: int, layers_dimensions: list[int], dropouts, output_dimension: int):
        super().__init__()
        if isinstance(dropouts, float):
            dropouts = [dropouts] * len(layers_dimensions)
        assert len(layers_dimensions) == len(dropouts)
        self.blocks = nn.ModuleList(
            [
                MLP.Block(input_dimension=layers_dimensions[i - 1] if i else input_dimension, output_dimension=d, bias=True, dropout=p)
                for i, (d, p) in enumerate(zip(layers_dimensions, dropouts))
            ]
        )
        self.head = nn.Linear(layers_dimensions[-1] if layers_dimensions else input_dimension, output_dimension)

    @classmethod
    def make_baseline(cls, input_dimension: int, layers_dimensions: list[int], dropout: float, output_dimension: int) -> "MLP":
        assert isinstance(dropout, float)
        if len(layers_dimensions) > 2:
            assert len(set(layers_dimensions[1:-1])) == 1, (
                "if d_layers contains more than two elements, then"
                " all elements except for the first and the last ones must be equal."
            )
        return cls(input_dimension=input_dimension, layers_dimensions=layers_dimensions, dropouts=dropout,
                   output_dimension=output_dimension)


class MLPDiffusion(nn.Module):
    """MODEL:548.  ``precision``: "fp32" (exact f32 MFMA) or "bf16" (fused kernel, fp32 accumulate)."""

    def __init__(self, input_dimension: int, num_classes: int, is_target_conditioned: IsTargetConditioned,
                 diffusion_parameters: DiffusionParameters, timestep_dimension: int = 128):
        super().__init__()
        self.timestep_dimension = timestep_dimension
        self.num_classes = num_classes
        self.is_target_conditioned = is_target_conditioned
        self.diffusion_parameters = diffusion_parameters
        self.diffusion_parameters.input_dimension = timestep_dimension
        self.diffusion_parameters.output_dimension = input_dimension
        self.input_dimension = input_dimension
        # MODEL:583-587: the label embedding of IsTargetConditioned.EMBEDDING.  With classes it runs on the MI355X path as extra
        # rows of the time-embedding table (Engine.set_label_embedding); the regression variant (num_classes == 0, a Linear on
        # the raw target) is kept for state-dict compatibility and refuses in forward when a target is passed
        self.label_conditioned = getattr(is_target_conditioned, "value", is_target_conditioned) == "embedding"
        # dropout > 0 (MODEL:162-180): the Dropout modules are kept for state-dict / structure compatibility.  Scoring and sampling
        # run the eval-mode network (identity); the training iteration (ClavaDDPMTrainer, the mixed_loss autograd seam) applies
        # Philox masks in the HIP kernels when the module is in training mode (training_dropout, mtk_train_set_dropout); only a
        # plain forward() in training mode refuses - the inference kernels carry no mask
        self.mlp = MLP.make_baseline(
            input_dimension=timestep_dimension,
            layers_dimensions=self.diffusion_parameters.layers_dimensions,
            dropout=self.diffusion_parameters.dropout,
            output_dimension=input_dimension,
        )
        if self.label_conditioned:
            self.label_embedding = nn.Embedding(num_classes, timestep_dimension) if num_classes > 0 else nn.Linear(1, timestep_dimension)
        self.proj = nn.Linear(input_dimension, timestep_dimension)
        self.timestep_embedding = nn.Sequential(
            nn.Linear(timestep_dimension, timestep_dimension),
            nn.SiLU(),
            nn.Linear(timestep_dimension, timestep_dimension),
        )
        self.precision = "fp32"
        self.num_timesteps = 1000  # size of the time-embedding table when used stand-alone
        self._engine: Engine | None = None
        self._engine_shared = False
        self._synced_key = None

    # ---- engine plumbing -------------------------------------------------------------------
    def __getstate__(self):
        """Pickles without the device context and its bookkeeping (rebuilt lazily; see Engine.__reduce__)."""
        state = dict(self.__dict__)
        state["_engine"], state["_engine_shared"], state["_synced_key"] = None, False, None
        return state

    def _weights(self):
        lin = [self.proj, self.timestep_embedding[0], self.timestep_embedding[2]]
        lin += [b.linear for b in self.mlp.blocks] + [self.mlp.head]
        return [(m.weight, m.bias) for m in lin]

    def _dims(self):
        return [self.input_dimension, self.timestep_dimension] + [int(b.linear.out_features) for b in self.mlp.blocks]

    def _label_rows(self) -> bool:
        return self.label_conditioned and self.num_classes > 0

    def attach_engine(self, engine: Engine) -> None:
        """Share the diffusion object's context (its D equals this network's width)."""
        if self._label_rows():
            raise RuntimeError("a label-conditioned denoiser keeps its own context: its time-embedding table has (classes + 1) x T rows")
        assert engine.D == self.input_dimension, (engine.D, self.input_dimension)
        self._engine, self._engine_shared, self._synced_key = engine, True, None

    def engine(self, device: torch.device) -> Engine:
        rows = self.num_timesteps * (self.num_classes + 1 if self._label_rows() else 1)  # table rows: (label, timestep) pairs
        stale = self._engine is not None and not self._engine_shared and self._engine.T != rows  # num_timesteps changed (a diffusion object set its T)
        if self._engine is None or self._engine.device != torch.device(device) or stale:
            if self._engine is not None and self._engine_shared:
                raise RuntimeError("denoiser and diffusion context live on different devices")
            buf = schedule.buffer_values(schedule.SchedulerType.COSINE, rows)
            tabs = {**buf, **schedule.posterior_values(buf)}
            self._engine = Engine(device, rows, self.input_dimension, [0], tabs)
            self._synced_key = None
        self.sync_weights()
        return self._engine

    def training_dropout(self) -> float:
        """The probability the training iteration drops hidden activations with: the blocks' Dropout p in training mode, 0 in eval."""
        return float(self.diffusion_parameters.dropout) if self.training else 0.0

    def _check_dropout(self) -> None:
        if self.training_dropout() != 0.0:
            raise NotImplementedError("MLPDiffusion.forward with dropout > 0 in training mode: the inference kernels evaluate the "
                                      "eval-mode network (Dropout = identity); call .eval() for scoring / sampling - the training "
                                      "iteration (ClavaDDPMTrainer / mixed_loss under autograd) applies the masks itself")

    def sync_weights(self) -> None:
        """Re-pack the weights into the library when any parameter changed (optimizer step, load)."""
        key = tuple((p.data_ptr(), p._version) for wb in self._weights() for p in wb)
        if self._label_rows():
            lw = self.label_embedding.weight
            key += ((lw.data_ptr(), lw._version, self.num_timesteps),)
        if key != self._synced_key:
            if self._label_rows():
                self._engine.set_label_embedding(self.label_embedding.weight, self.num_timesteps)
            self._engine.set_weights(self._weights(), self._dims(), timestep_frequencies(self.timestep_dimension))
            self._synced_key = key

    def forward(self, input_tensor: Tensor, timesteps: Tensor, y: Tensor | None = None) -> Tensor:
        """MODEL:596.  ``y`` is ignored unless the network is label-conditioned (IsTargetConditioned.EMBEDDING, MODEL:610-612):
        then record i runs with table row ``(1 + y[i]) * num_timesteps + timesteps[i]`` (timesteps must be below
        ``num_timesteps``, the table's period; ``y=None`` takes the label-free rows like the reference)."""
        self._check_dropout()
        if not input_tensor.is_cuda:
            raise RuntimeError("MLPDiffusion.forward: midst_toolkit_amd has no CPU path; move the module and inputs to the MI355X")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(
                "MLPDiffusion.forward does not build an autograd graph: train with midst_toolkit_amd.trainer.ClavaDDPMTrainer "
                "(HIP forward + backward + AdamW), or call under torch.no_grad() for scoring / sampling"
            )
        eng = self.engine(input_tensor.device)
        x = input_tensor.float()
        t = self.table_rows(timesteps.to(device=x.device, dtype=torch.int64), y)
        return eng.mlp_forward(x, t, precision=self.precision)

    def table_rows(self, timesteps: Tensor, y: Tensor | None) -> Tensor:
        """Row of the time-embedding table per record: the timestep, or the (label, timestep) pair of a label-conditioned net."""
        if self.label_conditioned and y is not None:
            if self.num_classes == 0:
                raise NotImplementedError("IsTargetConditioned.EMBEDDING with num_classes == 0 (a Linear on the raw regression target, MODEL:587) "
                                          "is not on the MI355X path")
            labels = y.to(device=timesteps.device).long().reshape(-1)  # MODEL:611 squeezes [B, 1] targets
            timesteps = (labels + 1) * self.num_timesteps + timesteps
        return timesteps.contiguous()


class ModelType(Enum):
    """MODEL:780."""

    MLP = "mlp"
    RESNET = "resnet"

    def get_model(self, model_parameters: ModelParameters) -> nn.Module:
        if self == ModelType.MLP:
            return MLPDiffusion(
                input_dimension=model_parameters.input_dimension,
                num_classes=model_parameters.num_classes,
                is_target_conditioned=model_parameters.is_target_conditioned,
                diffusion_parameters=model_parameters.diffusion_parameters,
            )
        raise ValueError(f"Unsupported model type: {self.value}")
