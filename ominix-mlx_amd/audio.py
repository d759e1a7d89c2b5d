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
    "omx_whisper_mel_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    "omx_whisper_mel_frames": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_int)]),
    "omx_whisper_mel_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "omx_sensevoice_mel_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, ctypes.c_float]),
    "omx_sensevoice_mel_frames": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_int)]),
    "omx_sensevoice_mel_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "omx_apply_lfr": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_resample_len": (c_int64, [c_int64, ctypes.c_uint32, ctypes.c_uint32]),
    "omx_resample_sinc": (c_int, [c_void_p, c_int64, ctypes.c_uint32, ctypes.c_uint32, c_void_p, c_int64, ctypes.POINTER(c_int64), c_void_p]),
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
        if sys is not None and not sys.is_finalizing() and getattr(self, "_h", None) is not None and self._h.value:
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


class WhisperMelFrontend:
    """qwen3-asr-mlx/src/audio.rs:32-128 (`MelFrontend::{new, compute_mel_spectrogram}`): WhisperFeatureExtractor-compatible
    log-mel, [n_mels, n_frames] float32 on the device."""

    def __init__(self, sample_rate=16000, n_mels=128, n_fft=400, hop_length=160):
        require_device()
        self.sample_rate, self.n_mels, self.n_fft, self.hop_length = sample_rate, n_mels, n_fft, hop_length
        self._h = c_void_p()
        check(lib.omx_whisper_mel_create(ctypes.byref(self._h), sample_rate, n_mels, n_fft, hop_length))

    def __del__(self):
        if sys is not None and not sys.is_finalizing() and getattr(self, "_h", None) is not None and self._h.value:
            lib.omx_mel_frontend_destroy(self._h)
            self._h = c_void_p()

    def compute_mel_spectrogram(self, samples) -> Tensor:
        a = samples if isinstance(samples, Tensor) else Tensor.from_numpy(np.asarray(samples, np.float32).ravel(), "f32")
        nf = c_int()
        check(lib.omx_whisper_mel_frames(self._h, a.size, ctypes.byref(nf)))
        out = Tensor((self.n_mels, nf.value), "f32")
        check(lib.omx_whisper_mel_forward(self._h, a.ptr, a.size, out.ptr, None))
        return out


class SenseVoiceMelFrontend:
    """funasr-nano-mlx/src/audio.rs:44-157 (`AudioConfig`, `MelFrontend::{new, compute_mel_spectrogram}`): the Fun-ASR-Nano / SenseVoice
    log-mel, [1, n_mels, n_frames] float32 on the device; `apply_lfr` (:345-412) stacks it to [1, ceil(T / n), m * n_mels]."""

    def __init__(self, sample_rate=16000, n_mels=80, n_fft=400, hop_length=160, max_length=30.0):
        require_device()
        self.sample_rate, self.n_mels, self.n_fft, self.hop_length, self.max_length = sample_rate, n_mels, n_fft, hop_length, max_length
        self.n_freqs = n_fft // 2 + 1
        self._h = c_void_p()
        check(lib.omx_sensevoice_mel_create(ctypes.byref(self._h), sample_rate, n_mels, n_fft, hop_length, max_length))

    def __del__(self):
        if sys is not None and not sys.is_finalizing() and getattr(self, "_h", None) is not None and self._h.value:
            lib.omx_mel_frontend_destroy(self._h)
            self._h = c_void_p()

    def compute_mel_spectrogram(self, samples) -> Tensor:
        a = samples if isinstance(samples, Tensor) else Tensor.from_numpy(np.asarray(samples, np.float32).ravel(), "f32")
        if a.size == 0:
            check(lib.omx_sensevoice_mel_frames(self._h, 0, ctypes.byref(c_int())))      # raises the reference's "samples are empty"
        nf = c_int()
        check(lib.omx_sensevoice_mel_frames(self._h, a.size, ctypes.byref(nf)))
        out = Tensor((1, self.n_mels, nf.value), "f32")
        check(lib.omx_sensevoice_mel_forward(self._h, a.ptr, a.size, out.ptr, None))
        return out


def apply_lfr(mel: Tensor, lfr_m: int, lfr_n: int) -> Tensor:
    """funasr-nano-mlx/src/audio.rs:345-412: [1, n_mels, n_frames] -> [1, ceil(n_frames / lfr_n), n_mels * lfr_m] (device to device; the
    reference copies the spectrogram to the host and back for this)."""
    if len(mel.shape) != 3 or mel.shape[0] != 1:
        raise ValueError(f"apply_lfr: expected [1, n_mels, n_frames], got {mel.shape}")
    _, n_mels, n_frames = mel.shape
    out = Tensor((1, (n_frames + lfr_n - 1) // lfr_n, n_mels * lfr_m), "f32")
    check(lib.omx_apply_lfr(out.ptr, mel.ptr, n_mels, n_frames, lfr_m, lfr_n, None))
    return out


def resample_device(samples: Tensor, src_rate: int, target_rate: int) -> Tensor:
    """audio::resample on a device tensor (f32 [n]) -> device tensor (f32 [m])."""
    n = samples.size
    cap = int(lib.omx_resample_len(n, src_rate, target_rate))
    out = Tensor((max(cap, 1),), "f32")
    m = c_int64()
    check(lib.omx_resample_sinc(samples.ptr if n else None, n, src_rate, target_rate, out.ptr, cap, ctypes.byref(m), None))
    if m.value == out.shape[0]:
        return out
    return Tensor((m.value,), "f32", ptr=out.ptr, owner=out)      # a prefix view of the allocation (short inputs give fewer samples)


def resample(samples, src_rate: int, target_rate: int) -> np.ndarray:
    """`audio::resample` (mlx-rs-core/src/audio.rs:178-277): host samples in, host samples out (float32), windowed-sinc interpolation with
    the reference's rubato configuration, evaluated on the GPU (csrc/resample.hip)."""
    x = np.ascontiguousarray(np.asarray(samples, np.float32).ravel())
    if src_rate == target_rate or x.size == 0:              # :179-181
        return x.copy()
    require_device()
    cap = int(lib.omx_resample_len(x.size, src_rate, target_rate))
    src = Tensor.from_numpy(x, "f32")
    out = Tensor((max(cap, 1),), "f32")
    m = c_int64()
    check(lib.omx_resample_sinc(src.ptr, x.size, src_rate, target_rate, out.ptr, cap, ctypes.byref(m), None))
    return out.numpy().ravel()[: m.value].copy()


# ---- WAV container, host side (mlx-rs-core/src/audio.rs:46-163 `load_wav`, :285-326 `save_wav`) ----

def load_wav(path):
    """-> (mono float32 samples in [-1, 1], sample_rate).  PCM 16 / 24 bit and 32-bit float; unknown chunks are skipped;
    multi-channel audio is averaged to mono; errors as the reference's ("Not a RIFF file", "Not a WAVE file",
    "Unsupported bits per sample: N")."""
    import struct
    with open(path, "rb") as fh:
        buf = fh.read()
    if buf[:4] != b"RIFF":
        raise ValueError("Not a RIFF file")
    if buf[8:12] != b"WAVE":
        raise ValueError("Not a WAVE file")
    pos, sample_rate, bits, channels, data = 12, 0, 16, 1, b""
    while pos + 8 <= len(buf):
        chunk_id = buf[pos:pos + 4]
        (size,) = struct.unpack_from("<I", buf, pos + 4)
        pos += 8
        if chunk_id == b"fmt ":
            _fmt, channels, sample_rate, _rate, _align, bits = struct.unpack_from("<HHIIHH", buf, pos)
        elif chunk_id == b"data":
            data = buf[pos:pos + size]
            break
        pos += size
    if bits == 16:
        x = np.frombuffer(data, "<i2", len(data) // 2).astype(np.float32) / np.float32(32768.0)
    elif bits == 24:
        b3 = np.frombuffer(data, np.uint8, len(data) // 3 * 3).reshape(-1, 3).astype(np.int32)
        v = ((b3[:, 0] << 8) | (b3[:, 1] << 16) | (b3[:, 2] << 24)) >> 8          # sign-extending shift, as the Rust
        x = v.astype(np.float32) / np.float32(8388608.0)
    elif bits == 32:
        x = np.frombuffer(data, "<f4", len(data) // 4).astype(np.float32)
    else:
        raise ValueError(f"Unsupported bits per sample: {bits}")
    if channels > 1:
        x = x[:x.size // channels * channels].reshape(-1, channels)
        acc = np.zeros(x.shape[0], np.float32)
        for c in range(channels):
            acc = acc + x[:, c]
        x = acc / np.float32(channels)
    return x.astype(np.float32), int(sample_rate)


def save_wav(samples, sample_rate: int, path) -> None:
    """16-bit PCM mono; sample -> (clamp(x, -1, 1) * 32767) truncated toward zero (`as i16`)."""
    import struct
    x = np.clip(np.asarray(samples, np.float32).ravel(), np.float32(-1.0), np.float32(1.0))
    pcm = np.trunc(x * np.float32(32767.0)).astype("<i2")
    data = pcm.tobytes()
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        fh.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16))
        fh.write(b"data" + struct.pack("<I", len(data)) + data)
