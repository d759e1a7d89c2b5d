"""ominix_mlx_amd -- host-side binding of libomx_hip.so (MI355X / gfx950).

The shared library is the product: hand-written HIP kernels behind a C ABI that mirrors
the mlx-c boundary of OminiX-MLX (include/omx.h, include/omx_mlx_c.h).  This module only
loads it with ctypes and mirrors the mlx-rs-core operator surface on top (ops.py,
cache.py, engine.py).  There is NO CPU fallback: importing works without a GPU (so the
symbol table can be checked), but any compute call without the library or without a
device raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMX_LIB_VARIANT=name loads libomx_hip_<name>.so (an A/B build made with `make VARIANT=name VARIANT_FLAGS=-D...`, tuning only)
LIB_PATH = os.path.join(_HERE, "libomx_hip%s.so" % ("_" + os.environ["OMX_LIB_VARIANT"] if os.environ.get("OMX_LIB_VARIANT") else ""))


class OmxError(RuntimeError):
    """Mirrors mlx_rs::error::Exception { what } built from the C error slot
    (mlx-rs/src/utils/guard.rs:32-47)."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise OmxError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ominix-mlx_amd/csrc` -- there is no CPU fallback")
    return ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


lib = _load()

c_void_p, c_int, c_int64, c_float, c_size_t, c_uint32 = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_uint32)

# mlx_dtype numbering (mlx/c/array.h:37-52)
BOOL, UINT8, UINT16, UINT32, UINT64, INT8, INT16, INT32, INT64, FLOAT16, FLOAT32, FLOAT64, BFLOAT16, COMPLEX64 = range(14)
MASK_NONE, MASK_CAUSAL, MASK_BOOL, MASK_ADDITIVE = range(4)

# name -> (restype, argtypes).  Every symbol include/omx.h declares must appear here
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "omx_quantize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_dequantize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_quantized_matmul": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_gather_mm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_gather_qmm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_void_p]),
    "omx_version": (ctypes.c_char_p, []),
    "omx_experiments_built": (ctypes.c_int, []),
    "omx_set_error_handler": (None, [c_void_p, c_void_p, c_void_p]),
    "omx_last_error": (ctypes.c_char_p, []),
    "omx_clear_error": (None, []),
    "omx_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "omx_device_name": (c_int, [ctypes.c_char_p, c_size_t]),
    "omx_synchronize": (c_int, [c_void_p]),
    "omx_malloc": (c_int, [ctypes.POINTER(c_void_p), c_size_t]),
    "omx_free": (c_int, [c_void_p]),
    "omx_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "omx_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "omx_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "omx_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "omx_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "omx_fill_uniform": (c_int, [c_void_p, c_size_t, c_uint32, c_float, c_float, c_int, c_void_p]),
    "omx_rms_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "omx_layer_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "omx_rope": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_void_p]),
    "omx_rope_freqs": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_float, c_int, c_int, c_void_p]),
    "omx_fused_swiglu": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "omx_fused_modulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "omx_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_linear_swiglu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_sdpa": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int64, c_int64, c_float, c_int, c_void_p, c_int, c_void_p]),
    "omx_sdpa_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "omx_set_workspace": (c_int, [c_void_p, c_size_t]),
    "omx_argmax": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "omx_random_key": (c_int, [c_void_p, ctypes.c_uint64, c_void_p]),
    "omx_random_split": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "omx_random_bits": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "omx_random_uniform": (c_int, [c_void_p, c_void_p, c_int64, ctypes.c_float, ctypes.c_float, c_void_p]),
    "omx_random_gumbel": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "omx_random_normal": (c_int, [c_void_p, c_void_p, c_int64, ctypes.c_float, ctypes.c_float, c_void_p]),
    "omx_random_categorical": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, ctypes.c_float, c_void_p, c_int, c_void_p]),
    "omx_take_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "omx_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
}


def _bind():
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise OmxError(f"libomx_hip.so does not export: {', '.join(missing)}")


_bind()


def check(status: int) -> None:
    """Status-code convention of the boundary: non-zero => raise with the library's message."""
    if status != 0:
        msg = lib.omx_last_error().decode("utf-8", "replace")
        lib.omx_clear_error()
        raise OmxError(msg or "unknown libomx_hip error")


def version() -> str:
    return lib.omx_version().decode()


def device_count() -> int:
    n = c_int(0)
    check(lib.omx_device_count(ctypes.byref(n)))
    return n.value


def require_device() -> None:
    if device_count() < 1:
        raise OmxError("no MI355X/HIP device visible: libomx_hip has no CPU path")


from . import ops  # noqa: E402,F401
