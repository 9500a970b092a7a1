"""ctypes binding of ``libwsae_hip.so`` (the C ABI declared in ``include/wsae.h``).

There is no CPU implementation behind this module: if the shared library is missing or a call
fails, a ``WsaeError`` is raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C whisper-sae_amd/csrc``.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_NAME = "libwsae_hip.so"

PREC_BF16, PREC_FP32 = 0, 1
DT_F32, DT_BF16 = 0, 1
PART_ALL, PART_DECODER, PART_ENCODER = -1, 0, 1  # wsae_weight_grads_wire


class WsaeError(RuntimeError):
    """A call into libwsae_hip failed (or the library is not built)."""


class Config(C.Structure):
    _fields_ = [("input_dim", C.c_int32), ("hidden_dim", C.c_int32), ("k", C.c_int32),
                ("max_batch", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32)]


class Stats(C.Structure):
    """Mirror of ``wsae_stats`` (8 x 4 bytes)."""

    _fields_ = [("loss", C.c_float), ("l0", C.c_float), ("grad_norm", C.c_float), ("clip_coef", C.c_float),
                ("dead_ratio", C.c_float), ("dead_count", C.c_int32), ("topk_fallback_rows", C.c_int32),
                ("reserved", C.c_int32)]


STATS_WORDS = C.sizeof(Stats) // 4

_p = C.c_void_p
_i32, _i64, _f32 = C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/wsae.h declares
SIGNATURES = {
    "wsae_last_error": (C.c_char_p, []),
    "wsae_version": (C.c_int, []),
    "wsae_param_count": (_i64, [_i32, _i32]),
    "wsae_param_offsets": (C.c_int, [_i32, _i32, C.POINTER(_i64)]),
    "wsae_ctx_create": (C.c_int, [C.POINTER(Config), C.POINTER(_p)]),
    "wsae_ctx_destroy": (C.c_int, [_p]),
    "wsae_ctx_workspace_bytes": (C.c_size_t, [_p]),
    "wsae_ctx_set_loss_cols": (C.c_int, [_p, _i32]),
    "wsae_ctx_reserve_relu": (C.c_int, [_p]),
    "wsae_ctx_set_relu_fp8": (C.c_int, [_p, _i32]),
    "wsae_ctx_set_relu_l1_weights": (C.c_int, [_p, _p]),
    "wsae_ctx_set_strip_predict": (C.c_int, [_p, _i32, C.c_float]),
    "wsae_ctx_strip_stats": (C.c_int, [_p, _p, _p, _p]),
    "wsae_ctx_set_fired": (C.c_int, [_p, _p]),
    "wsae_ctx_set_wire_metrics": (C.c_int, [_p, _p]),
    "wsae_prepare": (C.c_int, [_p, _p, _p]),
    "wsae_encode_topk": (C.c_int, [_p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _p]),
    "wsae_encode_dense": (C.c_int, [_p, _p, _p, _i32, _p, _i32, _p, _p]),
    "wsae_densify": (C.c_int, [_p, _p, _p, _i32, _p, _p]),
    "wsae_decode_dense": (C.c_int, [_p, _p, _p, _i32, _p, _p]),
    "wsae_feature_topk_workspace_bytes": (_i64, [_i64, _i32]),
    "wsae_feature_topk_update": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "wsae_decode_loss": (C.c_int, [_p, _p, _p, _i32, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _p]),
    "wsae_encode_decode": (C.c_int, [_p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "wsae_weight_grads": (C.c_int, [_p, _p, _p, _i32, _p, _p, _p, _p, _i32, _p, _p]),
    "wsae_last_residual_grad": (C.c_int, [_p, _i32, _p, _p]),
    "wsae_input_grad": (C.c_int, [_p, _p, _p, _p, _i32, _p, _i32, _p]),
    "wsae_wgrad_parts_supported": (C.c_int, [_p]),
    "wsae_ctx_set_comm_reserve": (C.c_int, [_p, _i32]),
    "wsae_weight_grads_wire": (C.c_int, [_p, _p, _p, _i32, _p, _p, _p, _p, _i32, _i32, _p, _i32, _p]),
    "wsae_grads_unpack_wire": (C.c_int, [_p, _p, _i32, _p, _p, _i32, _p, _p]),
    "wsae_adamw_step": (C.c_int, [_p, _p, _p, _p, _p, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _i32, _i32,
                                  _p, _p, _i64, _p, _p]),
    "wsae_normalize_decoder": (C.c_int, [_p, _p, _p]),
    "wsae_dead_scan": (C.c_int, [_p, _p, _p, _i64, _p, _p, _p]),
    "wsae_row_errors": (C.c_int, [_p, _p, _i32, _p, _p, _i32, _p, _p, _p]),
    "wsae_resample_dead": (C.c_int, [_p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    "wsae_ring_create": (C.c_int, [_i32, _i64, _i32, _i32, C.POINTER(_p)]),
    "wsae_ring_destroy": (C.c_int, [_p]),
    "wsae_ring_data": (_p, [_p]),
    "wsae_ring_size": (_i64, [_p]),
    "wsae_ring_push": (C.c_int, [_p, _p, _i32, _i64, _p]),
    "wsae_ring_push_layernorm": (C.c_int, [_p, _p, _i32, _i64, _p, _p, _f32, _p]),
    "wsae_ring_sample": (C.c_int, [_p, C.c_uint64, _i64, _i64, _i32, _p, _p]),
    "wsae_ring_fill_synthetic": (C.c_int, [_p, C.c_uint64, _i64, _p]),
    "wsae_kernel_name": (C.c_char_p, [_i32]),
    "wsae_profile_enable": (C.c_int, [_p, _i32, _i32]),
    "wsae_profile_disable": (C.c_int, [_p]),
    "wsae_profile_read": (C.c_int, [_p, _i32, C.POINTER(_i32), C.POINTER(C.c_double)]),
    "wsae_relu_needs_hidden": (C.c_int, [_p, _i32]),
    "wsae_relu_forward": (C.c_int, [_p, _p, _p, _i32, _p, _i32, _f32, _p, _p, _p, _p, _p]),
    "wsae_relu_backward": (C.c_int, [_p, _p, _p, _i32, _p, _i32, _f32, _p, _p, _p, _p]),
}

_lib = None


def library_path() -> Path:
    env = os.environ.get("WSAE_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / LIB_NAME


def lib() -> C.CDLL:
    """Load (once) and return the shared library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libwsae_hip needs libamdhip64.so.7 and must bind to the SAME HIP runtime instance as
    # PyTorch (it is handed torch's streams and device pointers); whichever copy of that soname is
    # loaded first serves the whole process, and it has to be the one torch ships.
    import torch  # noqa: F401
    path = library_path()
    if not path.exists():
        raise WsaeError(
            f"{path} not found: the HIP extension is not built.  This package has no CPU path; build the "
            f"library with `make -C whisper-sae_amd/csrc` (hipcc, gfx950) or __graft_entry__.build().")
    try:
        handle = C.CDLL(str(path))
    except OSError as exc:  # missing libamdhip64 etc.
        raise WsaeError(f"could not load {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise WsaeError(f"{path} does not export {name}: stale build?") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


def last_error() -> str:
    msg = lib().wsae_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise WsaeError(f"{what or 'libwsae_hip call'} failed (code {rc}): {last_error()}")


def param_count(input_dim: int, hidden_dim: int) -> int:
    return int(lib().wsae_param_count(input_dim, hidden_dim))


def param_offsets(input_dim: int, hidden_dim: int) -> list:
    arr = (_i64 * 5)()
    check(lib().wsae_param_offsets(input_dim, hidden_dim, arr), "wsae_param_offsets")
    return [int(v) for v in arr]


def pack_layout(input_dim: int, hidden_dim: int) -> tuple:
    """(total, offsets) of the flat pack, computed on the host (no library needed).

    Must agree with ``wsae_param_count`` / ``wsae_param_offsets`` (tests/test_native_abi.py checks).
    """
    d, h = input_dim, hidden_dim
    off = [0, d * h, 2 * d * h, 2 * d * h + h, 2 * d * h + h + d]
    return 2 * d * h + h + 2 * d, off


KERNEL_COUNT = 12
K_WGRAD = 5  # WSAE_K_WGRAD: the dominant kernel (bench.py's roofline object)


def profile_read(handle: int) -> dict:
    """{kernel name: (launches, total_ms)} for every kernel with recorded samples."""
    out = {}
    for k in range(KERNEL_COUNT):
        n, ms = _i32(0), C.c_double(0.0)
        check(lib().wsae_profile_read(handle, k, C.byref(n), C.byref(ms)), "wsae_profile_read")
        if n.value:
            out[lib().wsae_kernel_name(k).decode()] = (n.value, ms.value)
    return out


def ptr(t) -> int:
    """Device (or host) address of a torch tensor, or NULL for None."""
    return 0 if t is None else t.data_ptr()

WIRE_METRIC_SLOTS = 24  # include/wsae.h WSAE_WIRE_METRIC_SLOTS
