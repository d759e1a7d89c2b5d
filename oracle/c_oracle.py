"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/c/liboracle_c.so (the plain-C port of
the decode step used as bench.py's cpu_baseline and cross-checked against the numpy oracle)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "c", "liboracle_c.so")
bf16_p = ctypes.POINTER(ctypes.c_uint16)


class LayerCfg(ctypes.Structure):
    _fields_ = [("hidden", ctypes.c_int), ("inter", ctypes.c_int), ("heads", ctypes.c_int), ("kv_heads", ctypes.c_int),
                ("head_dim", ctypes.c_int), ("cap", ctypes.c_int), ("eps", ctypes.c_float),
                ("rope_theta", ctypes.c_float), ("rope_scale", ctypes.c_float)]


class Layer(ctypes.Structure):
    _fields_ = [(n, bf16_p) for n in ("q", "k", "v", "o", "gate", "up", "down", "q_norm", "k_norm", "in_ln",
                                      "post_ln", "kcache", "vcache")]


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


def load():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    lib.oracle_fill_uniform_bf16.argtypes = [bf16_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float, ctypes.c_float]
    lib.oracle_fill_uniform_bf16.restype = None
    lib.oracle_qwen3_layer_decode.argtypes = [ctypes.POINTER(LayerCfg), ctypes.POINTER(Layer), bf16_p, ctypes.c_int, bf16_p]
    lib.oracle_qwen3_layer_decode.restype = None
    lib.oracle_qwen3_scratch_elems.argtypes = [ctypes.POINTER(LayerCfg)]
    lib.oracle_qwen3_scratch_elems.restype = ctypes.c_size_t
    lib.oracle_qwen3_head.argtypes = [bf16_p, bf16_p, bf16_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, bf16_p, bf16_p]
    lib.oracle_qwen3_head.restype = ctypes.c_uint32
    lib.oracle_gemv_bf16.argtypes = [bf16_p, bf16_p, bf16_p, ctypes.c_int, ctypes.c_int]
    lib.oracle_gemv_bf16.restype = None
    return lib


def ptr(a: np.ndarray):
    assert a.dtype == np.uint16 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(bf16_p)
