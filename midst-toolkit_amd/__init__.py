"""midst_toolkit_amd — MI355X-native diffusion hot path of midst-toolkit (drop-in classes)."""
from .attack_features import AttackFeatureDriver  # noqa: F401
from .classifier import Classifier, get_conditioning_function  # noqa: F401
from . import tabsyn  # noqa: F401
from .data import FastTensorDataLoader, endless  # noqa: F401
from .diffusion import (  # noqa: F401
    FoundNaNsError,
    GaussianLossType,
    SampleMethod,
    GaussianMultinomialDiffusion,
    GaussianParametrization,
    Parametrization,
)
from .engine import Engine  # noqa: F401
from .model import (  # noqa: F401
    MLP,
    DiffusionParameters,
    IsTargetConditioned,
    MLPDiffusion,
    ModelParameters,
    ModelType,
    timestep_frequencies,
)
from .schedule import SchedulerType, betas_for_alpha_bar, get_named_beta_schedule  # noqa: F401
from .scoring import mixed_loss as tartan_federer_mixed_loss  # noqa: F401
from .scoring import score_records  # noqa: F401
from .trainer import ClavaDDPMTrainer, update_ema  # noqa: F401
from .transforms import AffineInverse, QuantileInverse, inverse_transform_for  # noqa: F401
