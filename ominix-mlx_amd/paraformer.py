"""Host mirror of funasr-mlx's `SanmEncoderLayer` (paraformer.rs:573-640) and `CIFPredictor::cif_fire`
(:779-879) over omx_sanm_encoder_layer / omx_cif_fire."""
from __future__ import annotations

import ctypes

import numpy as np

from . import FLOAT32, INT32, check, lib
from .ops import Tensor

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
_FIELDS = ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "out_w", "out_b", "fsmn_w", "norm2_w", "norm2_b", "ffn_up_w", "ffn_up_b",
           "ffn_down_w", "ffn_down_b")


class SanmLayerWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in _FIELDS]


PARAFORMER_SIGNATURES = {
    "omx_sanm_encoder_layer": (c_int, [c_void_p, c_void_p, ctypes.POINTER(SanmLayerWeights), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_cif_fire": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p]),
}
for _n, (_r, _a) in PARAFORMER_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


class SanmEncoderLayer:
    """weights: dict with the keys of SanmLayerWeights (Linear [out,in] + bias, fsmn_w [dim, k]), numpy arrays."""

    def __init__(self, weights: dict, heads: int = 4, kernel_size: int = 11):
        self._t = {k: Tensor.from_numpy(np.asarray(weights[k]), "bf16") for k in _FIELDS}
        self.w = SanmLayerWeights(*[self._t[k].ptr for k in _FIELDS])
        self.heads, self.kernel_size = heads, kernel_size
        self.in_dim = self._t["qkv_w"].shape[1]
        self.dim = self._t["out_w"].shape[0]
        self.ffn_dim = self._t["ffn_up_w"].shape[0]

    def forward(self, x: Tensor) -> Tensor:
        T = x.shape[-2]
        out = Tensor(tuple(x.shape[:-1]) + (self.dim,), x.dtype)
        check(lib.omx_sanm_encoder_layer(out.ptr, x.ptr, ctypes.byref(self.w), T, self.in_dim, self.dim, self.heads, self.ffn_dim,
                                         self.kernel_size, None))
        return out


def cif_fire(hidden: Tensor, alphas: Tensor, threshold: float = 1.0, tail_threshold: float = 0.45):
    """-> (frames Tensor [B, max_tokens, H] f32, counts np.ndarray [B]); max_tokens is trimmed on the host like the
    reference pads to the longest item (paraformer.rs:841-872)."""
    B, T, H = hidden.shape
    frames = Tensor((B, T + 1, H), FLOAT32)
    counts = Tensor((B,), INT32) if False else Tensor((B,), "u32")
    check(lib.omx_cif_fire(frames.ptr, counts.ptr, hidden.ptr, alphas.ptr, B, T, H, threshold, tail_threshold, T + 1, None))
    cnt = counts.numpy().astype(np.int32)
    mx = int(cnt.max()) if B else 0
    return frames.numpy()[:, :mx], cnt
