"""Container-only import shim for the read-only reference checkout.

The reference targets Python >= 3.12 (pyproject.toml:9); this image has 3.10.
Two ordinary incompatibilities are patched before import:
  * ``typing.Self``                      -> ``typing_extensions.Self``
  * ``logging.StreamHandler[Any]``       -> give StreamHandler a ``__class_getitem__``
Only ``tools/make_golden.py`` and ``tools/time_reference.py`` use this module. Nothing
under ``tests/``, ``bench.py`` or the package imports it: /root/reference does not exist
on the GPU box.
"""
from __future__ import annotations

import logging
import sys
import typing

REFERENCE_SRC = "/root/reference/src"


def install() -> None:
    import typing_extensions

    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self  # type: ignore[attr-defined]
    if not hasattr(logging.StreamHandler, "__class_getitem__"):
        logging.StreamHandler.__class_getitem__ = classmethod(lambda cls, item: cls)  # type: ignore[attr-defined]
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)


def load():
    """Return (gmd_module, du_module, model_module) of the reference."""
    install()
    from midst_toolkit.models.clavaddpm import diffusion_utils as du
    from midst_toolkit.models.clavaddpm import gaussian_multinomial_diffusion as gmd
    from midst_toolkit.models.clavaddpm import model as model

    return gmd, du, model


# --- optional third-party packages the wider reference imports at module scope but never uses on
# --- the hot path.  They are absent from this image; a lazy stub module is
# --- enough for `import` statements to succeed.  Nothing from them is ever called.
_ABSENT = ("syntheval", "category_encoders", "sdv", "faiss", "omegaconf", "hydra", "xgboost", "gower", "optuna")


class _StubModule(type(sys)):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        val = type(name, (), {"__init__": lambda self, *a, **k: None, "__class_getitem__": classmethod(lambda c, i: c)})
        setattr(self, name, val)
        return val


class _StubFinder:
    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery

        if fullname.split(".")[0] in _ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def load_tf():
    """Return the reference's tartan_federer_attack module (for its ``mixed_loss``, TF:40)."""
    install()
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    from midst_toolkit.attacks.tartan_federer import tartan_federer_attack as tf

    return tf
