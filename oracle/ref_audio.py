"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the Paraformer mel/STFT frontend
(SURVEY.md 8a row a12).  Never imported by the product path.

Follows funasr-mlx/src/paraformer.rs line by line (this arithmetic is fully in-tree):
    MelFrontend::new (Hamming window)         :195-222
    hz_to_mel / mel_to_hz / create_mel_filterbank   :231-275
    MelFrontend::forward                      :278-367  (x32768, pre-emphasis 0.97, STFT power,
                                                         mel + ln(max(.,1e-10)), LFR(7,6), CMVN)
    compute_stft                              :374-411  (400-pt FFT per frame, no padding)
The reference's FFT is rustfft 6.2 in f32; this restatement evaluates the DFT with numpy's rfft in
float64 on the SAME f32 windowed frames, i.e. it is the exact value the reference's own validation
(funasr-mlx/examples/validate_correctness.rs:19-59, 284-287: FFT vs direct DFT, relative-L2 < 1e-5)
accepts both implementations against.  PINNED by that criterion and by the reference's generator
signals (validate_correctness.rs:441-487), reproduced in `signals()` below.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


class ParaformerFrontendConfig:
    """ParaformerConfig::default() frontend fields (paraformer.rs:110-145)."""
    sample_rate = 16000
    n_mels = 80
    n_fft = 400
    hop_length = 160
    lfr_m = 7
    lfr_n = 6


def hamming_window(n_fft: int) -> np.ndarray:
    i = np.arange(n_fft, dtype=F32)
    t = i / F32(n_fft - 1)
    return (F32(0.54) - F32(0.46) * np.cos(F32(2.0) * F32(np.pi) * t).astype(F32)).astype(F32)


def hz_to_mel(hz):
    return F32(2595.0) * np.log10(F32(1.0) + np.asarray(hz, F32) / F32(700.0)).astype(F32)


def mel_to_hz(mel):
    return F32(700.0) * (np.power(F32(10.0), np.asarray(mel, F32) / F32(2595.0)).astype(F32) - F32(1.0))


def create_mel_filterbank(n_fft: int, n_mels: int, sample_rate: float) -> np.ndarray:
    """paraformer.rs:239-275: triangular HTK filters, edges inclusive, no area normalisation. [n_mels, n_freqs]"""
    n_freqs = n_fft // 2 + 1
    mel_min, mel_max = hz_to_mel(0.0), hz_to_mel(sample_rate / 2.0)
    i = np.arange(n_mels + 2, dtype=F32)
    mel_points = mel_to_hz(mel_min + (mel_max - mel_min) * i / F32(n_mels + 1))
    fft_freqs = np.arange(n_freqs, dtype=F32) * F32(sample_rate) / F32(n_fft)
    fb = np.zeros((n_mels, n_freqs), F32)
    for m in range(n_mels):
        fl, fc, fr = mel_points[m], mel_points[m + 1], mel_points[m + 2]
        up = (fft_freqs >= fl) & (fft_freqs <= fc)
        dn = (fft_freqs > fc) & (fft_freqs <= fr)
        fb[m, up] = (fft_freqs[up] - fl) / (fc - fl)
        fb[m, dn] = (fr - fft_freqs[dn]) / (fr - fc)
    return fb


def preemphasis_scaled(audio: np.ndarray) -> np.ndarray:
    """x*32768 then y[0]=x[0], y[i]=x[i]-0.97*x[i-1] in f32 (paraformer.rs:289-300)."""
    x = (np.asarray(audio, F32) * F32(32768.0)).astype(F32)
    y = x.copy()
    y[1:] = (x[1:] - (F32(0.97) * x[:-1]).astype(F32)).astype(F32)
    return y


def stft_power(samples: np.ndarray, window: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """compute_stft (:374-411): frames (n - n_fft)/hop + 1, no centre padding; |X_k|^2 for k <= n_fft/2.
    Too-short input yields ONE all-zero frame (the `vec![0; n_freqs]` early return, :386-388)."""
    n_freqs = n_fft // 2 + 1
    n = len(samples)
    n_frames = (n - n_fft) // hop + 1 if n >= n_fft else 0
    if n_frames == 0:
        return np.zeros((1, n_freqs), np.float64)
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = (samples[idx] * window[None, :]).astype(F32)          # f32 product, as buffer[i] = s*w
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    return spec.real ** 2 + spec.imag ** 2


def lfr_indices(n_frames: int, lfr_m: int, lfr_n: int) -> np.ndarray:
    """:326-351: left-pad (m-1)/2 copies of frame 0, tail clamps to the last frame. [T', m]"""
    left = (lfr_m - 1) // 2
    t_out = (n_frames + left + lfr_n - 1) // lfr_n
    padded = np.arange(t_out)[:, None] * lfr_n + np.arange(lfr_m)[None, :]
    return np.clip(padded - left, 0, n_frames - 1)


def mel_frontend(audio: np.ndarray, addshift=None, rescale=None, cfg=ParaformerFrontendConfig):
    """MelFrontend::forward -> dict(power [T,201] f64, logmel [T,80] f32, feats [T', 560] f32)."""
    audio = np.asarray(audio, F32)
    if not np.all(np.isfinite(audio)):
        raise ValueError("Audio contains NaN or Inf values")          # :284-286
    window = hamming_window(cfg.n_fft)
    fb = create_mel_filterbank(cfg.n_fft, cfg.n_mels, float(cfg.sample_rate))
    power = stft_power(preemphasis_scaled(audio), window, cfg.n_fft, cfg.hop_length)
    mel = power @ fb.astype(np.float64).T
    logmel = np.log(np.maximum(mel, 1e-10)).astype(F32)
    idx = lfr_indices(logmel.shape[0], cfg.lfr_m, cfg.lfr_n)
    feats = logmel[idx].reshape(idx.shape[0], cfg.lfr_m * cfg.n_mels)
    if addshift is not None and rescale is not None:
        feats = ((feats + np.asarray(addshift, F32)[None, :]) * np.asarray(rescale, F32)[None, :]).astype(F32)
    return {"power": power, "logmel": logmel, "feats": feats.astype(F32)}


# ---- the reference's deterministic test signals (validate_correctness.rs:441-487) ----
def _t(n, sr):
    return np.arange(n, dtype=F32) / F32(sr)


def generate_sine(sr, duration, freq):
    n = int(duration * sr)
    return np.sin(F32(2.0) * F32(np.pi) * F32(freq) * np.arange(n, dtype=F32) / F32(sr)).astype(F32)


def generate_noise(sr, duration):
    n = int(duration * sr)
    out = np.empty(n, F32)
    seed = 12345
    for i in range(n):
        seed = (seed * 1103515245 + 12345) & 0xFFFFFFFFFFFFFFFF
        out[i] = F32(F32(seed >> 16) / F32(32768.0)) - F32(1.0)
    return out


def generate_mixed(sr, duration):
    t = _t(int(duration * sr), sr)
    tw = F32(2.0) * F32(np.pi)
    return (F32(0.5) * np.sin(tw * F32(200.0) * t) + F32(0.3) * np.sin(tw * F32(500.0) * t)
            + F32(0.2) * np.sin(tw * F32(1000.0) * t)).astype(F32)


def generate_speech_like(sr, duration):
    t = _t(int(duration * sr), sr)
    tw = F32(2.0) * F32(np.pi)
    f0 = F32(150.0)
    env = np.abs(np.sin(tw * F32(3.0) * t))
    return (env * (F32(0.4) * np.sin(tw * f0 * t) + F32(0.3) * np.sin(tw * F32(2.0) * f0 * t)
                   + F32(0.2) * np.sin(tw * F32(3.0) * f0 * t) + F32(0.1) * np.sin(tw * F32(4.0) * f0 * t))).astype(F32)


def signals(sr=16000, duration=1.0):
    """The four STFT validation cases of the reference plus its LCG noise generator."""
    return {
        "sine_440": generate_sine(sr, duration, 440.0),
        "mixed": generate_mixed(sr, duration),
        "speech_like": generate_speech_like(sr, duration),
        "noise_lcg": generate_noise(sr, min(duration, 1.0)),
    }


def stft_power_direct_dft_f32(samples, window, n_fft, hop):
    """The reference's OWN oracle: compute_stft_reference (validate_correctness.rs:19-59), O(N^2) DFT
    with f32 angles and f32 accumulation.  Used to pin this module by the reference's criterion."""
    n_freqs = n_fft // 2 + 1
    n_frames = (len(samples) - n_fft) // hop + 1 if len(samples) >= n_fft else 0
    if n_frames == 0:
        return np.zeros((1, n_freqs), F32)
    k = np.arange(n_freqs, dtype=F32)[:, None]
    n = np.arange(n_fft, dtype=F32)[None, :]
    ang = (F32(2.0) * F32(np.pi) * k * n / F32(n_fft)).astype(F32)
    c, s = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    out = np.empty((n_frames, n_freqs), F32)
    for f in range(n_frames):
        w = (samples[f * hop:f * hop + n_fft] * window).astype(F32)
        re = (c * w[None, :]).sum(axis=1, dtype=F32)
        im = -(s * w[None, :]).sum(axis=1, dtype=F32)
        out[f] = re * re + im * im
    return out


# --------------------------------------------------------------------------
# 8f rank 4 sibling frontend: WhisperFeatureExtractor-compatible log-mel of qwen3-asr-mlx/src/audio.rs
#   MelFrontend::new :41-66, compute_mel_spectrogram :68-128, Slaney scale :229-253, filterbank :260-319
# --------------------------------------------------------------------------

def hz_to_slaney_mel(freq):
    f_sp, min_log_hz = np.float32(200.0 / 3.0), np.float32(1000.0)
    min_log_mel = min_log_hz / f_sp
    logstep = np.float32(np.log(np.float32(6.4))) / np.float32(27.0)
    freq = np.float32(freq)
    return freq / f_sp if freq < min_log_hz else min_log_mel + np.float32(np.log(freq / min_log_hz)) / logstep


def slaney_mel_to_hz(mel):
    f_sp, min_log_hz = np.float32(200.0 / 3.0), np.float32(1000.0)
    min_log_mel = min_log_hz / f_sp
    logstep = np.float32(np.log(np.float32(6.4))) / np.float32(27.0)
    mel = np.float32(mel)
    return f_sp * mel if mel < min_log_mel else min_log_hz * np.float32(np.exp(logstep * (mel - min_log_mel)))


def whisper_mel_filterbank(sample_rate: int = 16000, n_fft: int = 400, n_mels: int = 128) -> np.ndarray:
    """[n_mels, n_freqs] float32, Slaney scale + Slaney (2 / bandwidth) normalisation."""
    n_freqs = n_fft // 2 + 1
    fmax = np.float32(sample_rate) / np.float32(2.0)
    mel_min, mel_max = hz_to_slaney_mel(0.0), hz_to_slaney_mel(fmax)
    ff = [slaney_mel_to_hz(mel_min + (mel_max - mel_min) * np.float32(i) / np.float32(n_mels + 1)) for i in range(n_mels + 2)]
    fb = np.zeros((n_mels, n_freqs), np.float32)
    for m in range(n_mels):
        lower, center, upper = ff[m], ff[m + 1], ff[m + 2]
        for k in range(n_freqs):
            freq = np.float32(k) * fmax / np.float32(n_freqs - 1)
            if lower <= freq <= center and center > lower:
                fb[m, k] = (freq - lower) / (center - lower)
            elif center < freq <= upper and upper > center:
                fb[m, k] = (upper - freq) / (upper - center)
        bw = upper - lower
        if bw > 0:
            fb[m] *= np.float32(2.0) / bw
    return fb


def whisper_log_mel(samples: np.ndarray, sample_rate: int = 16000, n_mels: int = 128, n_fft: int = 400, hop: int = 160) -> np.ndarray:
    """compute_mel_spectrogram (audio.rs:68-128): [n_mels, n_frames]; DFT in float64 (the reference uses rustfft in f32)."""
    x = np.asarray(samples, np.float32)
    if x.size == 0:
        raise ValueError("Audio samples are empty")
    if x.size < n_fft:
        raise ValueError("Audio too short")
    n_frames = 1 + (x.size - n_fft) // hop
    window = (np.float32(0.5) * (np.float32(1.0) - np.cos(np.float32(2.0) * np.float32(np.pi) * np.arange(n_fft, dtype=np.float32)
                                                          / np.float32(n_fft)).astype(np.float32))).astype(np.float32)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = (x[idx] * window[None, :]).astype(np.float32)
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2)
    mel = power @ whisper_mel_filterbank(sample_rate, n_fft, n_mels).astype(np.float64).T        # [frames, mels]
    logm = np.log10(np.maximum(mel, 1e-10))
    logm = np.maximum(logm, logm.max() - 8.0)
    return ((logm + 4.0) / 4.0).T.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# Fun-ASR-Nano / SenseVoice frontend: funasr-nano-mlx/src/audio.rs:44-157 (MelFrontend), :287-339 (filterbank), :345-412 (apply_lfr).
# All arithmetic is in-tree except the FFT (rustfft f32), evaluated here in float64 on the same f32 windowed frames -- the criterion the
# reference's own FFT-vs-DFT validation uses (see the module header).  The reference's tests for it check shapes and the two error
# cases only (:418-484); tests/test_sensevoice_oracle.py replays those.
# ---------------------------------------------------------------------------------------------------------------------
def sensevoice_mel_filterbank(sample_rate: int = 16000, n_fft: int = 400, n_mels: int = 80) -> np.ndarray:
    """create_mel_filterbank (:287-339): triangles between FFT BIN indices floor((n_fft + 1) * hz / sr); [n_mels, n_freqs] f32."""
    n_freqs = n_fft // 2 + 1
    f = np.float32
    hz2mel = lambda hz: f(2595.0) * np.log10(f(1.0) + f(hz) / f(700.0), dtype=np.float32)
    mel2hz = lambda mel: f(700.0) * (np.power(f(10.0), f(mel) / f(2595.0), dtype=np.float32) - f(1.0))
    lo, hi = hz2mel(0.0), hz2mel(f(sample_rate) / f(2.0))
    bins = []
    for i in range(n_mels + 2):
        hz = mel2hz(lo + (hi - lo) * f(i) / f(n_mels + 1))
        bins.append(int(np.floor(f(n_fft + 1) * hz / f(sample_rate))))
    fb = np.zeros((n_mels, n_freqs), np.float32)
    for m in range(n_mels):
        left, center, right = bins[m], bins[m + 1], bins[m + 2]
        for k in range(left, center):
            if k < n_freqs and center > left:
                fb[m, k] = f(k - left) / f(center - left)
        for k in range(center, right):
            if k < n_freqs and right > center:
                fb[m, k] = f(right - k) / f(right - center)
    return fb


def sensevoice_log_mel(samples, sample_rate: int = 16000, n_mels: int = 80, n_fft: int = 400, hop: int = 160, max_length: float = 30.0) -> np.ndarray:
    """MelFrontend::compute_mel_spectrogram (:93-157): [n_mels, n_frames] float32 (the reference adds a leading batch axis of 1)."""
    x = np.asarray(samples, np.float32).ravel()
    if x.size == 0:
        raise ValueError("Cannot compute mel spectrogram: audio samples are empty")                  # :95-99
    if x.size < hop:
        raise ValueError("Audio too short")                                                         # :105-110
    x = x[: int(np.float32(max_length) * np.float32(sample_rate))]                                   # :112-117
    n_frames = max(x.size // hop, 1)                                                                # :121
    window = (np.float32(0.5) * (np.float32(1.0) - np.cos(np.float32(2.0) * np.float32(np.pi) * np.arange(n_fft, dtype=np.float32)
                                                          / np.float32(n_fft - 1), dtype=np.float32))).astype(np.float32)   # :68-72
    padded = np.concatenate([x, np.zeros(n_fft, np.float32)])                                       # :131-134: zeros past the end
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = (padded[idx] * window[None, :]).astype(np.float32)
    spec = np.fft.fft(frames.astype(np.float64), axis=1)[:, : n_fft // 2 + 1]
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ sensevoice_mel_filterbank(sample_rate, n_fft, n_mels).astype(np.float64).T
    return np.log(np.maximum(mel, 1e-10)).T.astype(np.float32)


def apply_lfr(mel: np.ndarray, lfr_m: int = 7, lfr_n: int = 6) -> np.ndarray:
    """apply_lfr (:345-412): mel [n_mels, n_frames] -> [ceil(n_frames / lfr_n), lfr_m * n_mels]; frame t stacks source frames
    t * lfr_n + (j - lfr_m // 2), clamped to [0, n_frames - 1]."""
    n_mels, n_frames = mel.shape
    t_out = (n_frames + lfr_n - 1) // lfr_n
    out = np.zeros((t_out, lfr_m * n_mels), np.float32)
    for t in range(t_out):
        center = t * lfr_n
        for j in range(lfr_m):
            if j < lfr_m // 2:
                off = lfr_m // 2 - j
                src = 0 if off > center else center - off
            else:
                src = min(center + (j - lfr_m // 2), n_frames - 1)
            out[t, j * n_mels:(j + 1) * n_mels] = mel[:, src]
    return out


# WAV container (mlx-rs-core/src/audio.rs:46-163 load_wav, :285-326 save_wav)
def wav_bytes(samples, sample_rate: int, bits: int = 16, channels: int = 1, extra_chunk: bool = False) -> bytes:
    """Test helper: a RIFF/WAVE image (PCM 16/24 or float 32) with an optional unknown chunk before `data`."""
    import struct
    x = np.asarray(samples, np.float32).reshape(-1, channels)
    if bits == 16:
        body = (np.clip(x, -1, 1) * 32767.0).astype("<i2").tobytes()
        fmt_tag = 1
    elif bits == 24:
        v = (np.clip(x, -1, 1) * 8388607.0).astype("<i4").reshape(-1)
        body = b"".join(int(s).to_bytes(4, "little", signed=True)[:3] for s in v)
        fmt_tag = 1
    else:
        body = x.astype("<f4").tobytes()
        fmt_tag = 3
    fmt = struct.pack("<HHIIHH", fmt_tag, channels, sample_rate, sample_rate * channels * bits // 8, channels * bits // 8, bits)
    chunks = b"fmt " + struct.pack("<I", 16) + fmt
    if extra_chunk:
        chunks += b"LIST" + struct.pack("<I", 6) + b"abcdef"
    chunks += b"data" + struct.pack("<I", len(body)) + body
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def load_wav_bytes(buf: bytes):
    """load_wav (audio.rs:46-163): (mono float32 samples, sample_rate)."""
    import struct
    if buf[:4] != b"RIFF":
        raise ValueError("Not a RIFF file")
    if buf[8:12] != b"WAVE":
        raise ValueError("Not a WAVE file")
    pos, sr, bits, ch, data = 12, 0, 16, 1, b""
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        pos += 8
        if cid == b"fmt ":
            ch, sr = struct.unpack("<H", buf[pos + 2:pos + 4])[0], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
            bits = struct.unpack("<H", buf[pos + 14:pos + 16])[0]
        elif cid == b"data":
            data = buf[pos:pos + size]
            break
        pos += size
    if bits == 16:
        x = np.frombuffer(data[:len(data) // 2 * 2], "<i2").astype(np.float32) / np.float32(32768.0)
    elif bits == 24:
        raw = np.frombuffer(data[:len(data) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = (raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16))
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / np.float32(8388608.0)
    elif bits == 32:
        x = np.frombuffer(data[:len(data) // 4 * 4], "<f4").astype(np.float32)
    else:
        raise ValueError(f"Unsupported bits per sample: {bits}")
    if ch > 1:
        x = x[:x.size // ch * ch].reshape(-1, ch)
        acc = np.zeros(x.shape[0], np.float32)
        for c in range(ch):                       # iter().sum::<f32>() in channel order, then / channels
            acc = (acc + x[:, c]).astype(np.float32)
        x = (acc / np.float32(ch)).astype(np.float32)
    return x, sr
