"""ctypes binding of libmtk_hip.so (the C ABI declared in include/mtk.h).

PyTorch is used only for device memory and the current stream handle; every compute call goes
through the shared library.  There is no fallback: if the library is missing or a call fails the
caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTK_LIB_PATH") or os.path.join(_HERE, "libmtk_hip.so")  # override: A/B builds of the same ABI

MTK_OK, MTK_ERR_ARG, MTK_ERR_UNSUPPORTED, MTK_ERR_HIP, MTK_ERR_STATE = 0, -1, -2, -3, -4
PREC_FP32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
NTAB = 15
NOALLOC_MAX_ROWS, NOALLOC_MAX_WIDTH = 4096, 1024  # MTK_NOALLOC_MAX_ROWS / _WIDTH of include/mtk.h
TABLE_ORDER = [
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_alpha", "log_1_min_alpha", "log_cumprod_alpha",
    "log_1_min_cumprod_alpha", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
    "posterior_mean_coef2", "model_log_variance", "model_variance", "alphas_cumprod", "alphas_cumprod_prev",
    "posterior_log_variance_clipped",
]

_lib = None

_p, _i64, _i32, _u64 = C.c_void_p, C.c_int64, C.c_int, C.c_uint64
_SIGS = {
    "mtk_last_error": (C.c_char_p, []),
    "mtk_version": (C.c_int, []),
    "mtk_ctx_create": (C.c_int, [_i32, _i32, _i32, _i32, _p, _p, _p, C.POINTER(_p)]),
    "mtk_ctx_destroy": (None, [_p]),
    "mtk_ctx_set_math": (C.c_int, [_p, _i32]),
    "mtk_ctx_set_parametrization": (C.c_int, [_p, _i32, _i32]),
    "mtk_ctx_reserve": (C.c_int, [_p, _i64, _p]),
    "mtk_ctx_set_seed_source": (C.c_int, [_p, _p]),
    "mtk_ctx_workspace_generation": (_i64, [_p]),
    "mtk_ctx_alloc_count": (_i64, [_p]),
    "mtk_mlp_set_weights": (C.c_int, [_p, _i32, _p, _p, _p, _p, _p]),
    "mtk_mlp_set_label_embedding": (C.c_int, [_p, _p, _i32, _i32, _i32, _p]),
    "mtk_mlp_forward": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i32, _p, _i64, _i64, _i32, _p]),
    "mtk_mlp_state_input_ok": (C.c_int, [_p, _i32]),
    "mtk_mlp_forward_state": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _p, _i64, _i64, _i32, _p]),
    "mtk_mlp_forward_noalloc": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i32, _p, _i64, _i64, _p]),
    "mtk_q_sample": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _u64, _u64, _p]),
    "mtk_loss_rows": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _p, _p, _p, _p, _u64, _u64, _p]),
    "mtk_tf_score": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _i32, _p, _i32, _p, _i32, _p]),
    "mtk_p_sample_step": (C.c_int, [_p, _i64, _i32, _i32, _p, _i64, _i64, _p, _p, _p, _i32, _p, _p, _i64, _i64, _u64, _u64, _p]),
    "mtk_gauss_kl_rows": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _p, _p]),
    "mtk_affine_inverse": (C.c_int, [_i32, _i64, _i32, _p, _i64, _i32, _p, _p, _p, _i64, _p]),
    "mtk_quantile_inverse": (C.c_int, [_i32, _i64, _i32, _p, _i64, _p, _p, _i32, _i32, _p, _i64, _p]),
    "mtk_ddim_step": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _u64, _u64, _p]),
    "mtk_ddim_step_ex": (C.c_int, [_p, _i64, _i32, C.c_double, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _u64, _u64, _p]),
    "mtk_ddim_reverse_step": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _p, _p, _i64, _i64, _p]),
    "mtk_p_sample_step_guided": (C.c_int, [_p, _i64, _i32, _i32, _p, _i64, _i64, _p, _p, _p, _i32, _p, _p, _p, _i64, _i64, _u64, _u64, _p]),
    "mtk_sample_init": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _u64, _u64, _p]),
    "mtk_sample_decode": (C.c_int, [_p, _i64, _p, _p, _p, _p]),
    "mtk_train_image_elems": (_i64, [_p]),
    "mtk_train_attach": (C.c_int, [_p, _p, _p, _p]),
    "mtk_train_step": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, C.c_double, _p, _p, _u64, _u64, _p]),
    "mtk_train_bucket_layout": (C.c_int, [_p, _i32, _p, _p]),
    "mtk_train_set_bucket_events": (C.c_int, [_p, _i32, _p]),
    "mtk_train_set_gaussian_loss": (C.c_int, [_p, _i32]),
    "mtk_train_set_precision": (C.c_int, [_p, _i32]),
    "mtk_train_set_deterministic": (C.c_int, [_p, _i32]),
    "mtk_train_set_dropout": (C.c_int, [_p, C.c_double, _u64]),
    "mtk_train_set_label_classes": (C.c_int, [_p, _i32]),
    "mtk_train_set_labels": (C.c_int, [_p, _p]),
    "mtk_train_set_loss_means": (C.c_int, [_p, _p]),
    "mtk_train_set_deferred_reduce": (C.c_int, [_p, C.c_int, _p]),
    "mtk_train_timesteps": (C.c_int, [_p, C.c_int64, _p, _p]),
    "mtk_train_dropout_mask": (C.c_int, [_p, _i32, _i64, _u64, _p, _p]),
    "mtk_train_forward": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _u64, _u64, _p]),
    "mtk_train_backward": (C.c_int, [_p, _i64, _p, _i64, _p, _p, C.c_double, C.c_double, _u64, _u64, _p]),
    "mtk_train_backward_dev": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _u64, _u64, _p]),
    "mtk_train_apply": (C.c_int, [_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64, C.c_double, _p]),
    "mtk_clf_create": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, C.POINTER(_p)]),
    "mtk_clf_destroy": (None, [_p]),
    "mtk_clf_set_precision": (C.c_int, [_p, _i32]),
    "mtk_clf_guidance": (C.c_int, [_p, _i64, _p, _i64, _p, _p, C.c_double, _p, _p, _p]),
    "mtk_edm_create": (C.c_int, [_i32, _i32, _i32, _p, _p, _p, _p, C.POINTER(_p)]),
    "mtk_edm_destroy": (None, [_p]),
    "mtk_edm_forward": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _p, _p]),
    "mtk_edm_denoise": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, C.c_double, _p, _p]),
    "mtk_edm_loss": (C.c_int, [_p, _i64, _p, _i64, _p, _p, C.c_double, C.c_double, C.c_double, _p, _p]),
    "mtk_clft_create": (C.c_int, [_i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "mtk_clft_destroy": (None, [_p]),
    "mtk_clft_set_params": (C.c_int, [_p, _p, _p, _p, _p, _p]),
    "mtk_clft_forward": (C.c_int, [_p, _i64, _p, _i64, _p, C.c_double, _u64, C.c_double, _p, _p, _p, _p]),
    "mtk_clft_dropout_mask": (C.c_int, [_p, _i32, _i64, _p, _p]),
    "mtk_clft_backward": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p]),
    "mtk_edm_loss_keep": (C.c_int, [_p, _i64, _p, _i64, _p, _p, C.c_double, C.c_double, C.c_double, _p, _p]),
    "mtk_edm_backward": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p]),
    "mtk_edm_set_weights": (C.c_int, [_p, _p, _p, _p]),
    "mtk_train_export": (C.c_int, [_p, _i32, _p, _p, _p]),
    "mtk_train_import": (C.c_int, [_p, _i32, _p, _p, _p]),
}


def lib() -> C.CDLL:
    """Load libmtk_hip.so (once).  Raises if it has not been built: there is no other path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). midst_toolkit_amd has no non-HIP path."
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def exported_symbols() -> list[str]:
    return list(_SIGS)


def check(rc: int) -> None:
    if rc == MTK_OK:
        return
    msg = lib().mtk_last_error().decode()
    if rc == MTK_ERR_ARG:
        raise AssertionError(msg)
    if rc == MTK_ERR_UNSUPPORTED:
        raise ValueError(msg)
    raise RuntimeError(f"libmtk_hip: {msg} (status {rc})")


def ptr(t: torch.Tensor | None) -> int | None:
    """Device address of ``t`` for a C-ABI pointer argument (None -> NULL).  Every such argument is dereferenced by a kernel: a host
    tensor here would be a GPU memory fault a few calls later, so it is refused where it is handed over."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"midst_toolkit_amd: a {tuple(t.shape)} {t.dtype} tensor on {t.device} was passed where the library "
                           "reads device memory; move it to the MI355X first (the package has no CPU path)")
    return t.data_ptr()


def stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on {t.device}; midst_toolkit_amd runs on MI355X (ROCm) devices only and has no CPU path"
        )
