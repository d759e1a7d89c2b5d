"""Host mirror of funasr-mlx's `MelFrontend` (funasr-mlx/src/paraformer.rs:195-412) over the
omx_mel_frontend_* C ABI: same constructor inputs (ParaformerConfig frontend fields), `set_cmvn`,
`forward(audio) -> [1, T', 560]`, same error for non-finite audio."""
from __future__ import annotations

import ctypes
import sys

import numpy as np

from . import check, lib, require_device
from .ops import Tensor

c_int, c_void_p, c_int64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_int64


class MelConfig(ctypes.Structure):
    _fields_ = [("sample_rate", c_int), ("n_mels", c_int), ("n_fft", c_int), ("hop_length", c_int),
                ("lfr_m", c_int), ("lfr_n", c_int)]


AUDIO_SIGNATURES = {
    "omx_mel_frontend_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(MelConfig)]),
    "omx_mel_frontend_destroy": (c_int, [c_void_p]),
    "omx_mel_frontend_set_cmvn": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "omx_mel_frontend_frames": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "omx_mel_frontend_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
}
for _n, (_r, _a) in AUDIO_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


class MelFrontend:
    def __init__(self, sample_rate=16000, n_mels=80, n_fft=400, hop_length=160, lfr_m=7, lfr_n=6):
        require_device()
        self.cfg = MelConfig(sample_rate, n_mels, n_fft, hop_length, lfr_m, lfr_n)
        self._h = c_void_p()
        check(lib.omx_mel_frontend_create(ctypes.byref(self._h), ctypes.byref(self.cfg)))

    def __del__(self):
        if not sys.is_finalizing() and getattr(self, "_h", None) is not None and self._h.value:
            lib.omx_mel_frontend_destroy(self._h)
            self._h = c_void_p()

    def set_cmvn(self, addshift, rescale) -> None:
        a = np.ascontiguousarray(addshift, np.float32)
        r = np.ascontiguousarray(rescale, np.float32)
        check(lib.omx_mel_frontend_set_cmvn(self._h, a.ctypes.data, r.ctypes.data, a.size))

    def frames(self, n_samples: int):
        nf, nl = c_int(), c_int()
        check(lib.omx_mel_frontend_frames(self._h, n_samples, ctypes.byref(nf), ctypes.byref(nl)))
        return nf.value, nl.value

    def forward(self, audio, return_intermediates: bool = False):
        """audio: Tensor f32 [n] (device) or a numpy array.  Returns Tensor [1, T', lfr_m*n_mels] f32."""
        a = audio if isinstance(audio, Tensor) else Tensor.from_numpy(np.asarray(audio, np.float32).ravel(), "f32")
        n = a.size
        nf, nl = self.frames(n)
        dim = self.cfg.lfr_m * self.cfg.n_mels
        feats = Tensor((1, nl, dim), "f32")
        logmel = Tensor((nf, self.cfg.n_mels), "f32") if return_intermediates else None
        power = Tensor((nf, self.cfg.n_fft // 2 + 1), "f32") if return_intermediates else None
        check(lib.omx_mel_frontend_forward(self._h, a.ptr, n, feats.ptr, logmel.ptr if logmel else None,
                                           power.ptr if power else None, None))
        return (feats, logmel, power) if return_intermediates else feats
