"""Audio either side of the frontends: the WAV container of mlx-rs-core/src/audio.rs:46-163,
285-326 (host) and the WhisperFeatureExtractor-compatible log-mel of qwen3-asr-mlx/src/audio.rs:24-128 (GPU).
The oracle's Slaney filterbank is pinned on transformers' own `mel_filter_bank(norm="slaney", mel_scale="slaney")`,
the implementation the reference says it matches."""
import numpy as np
import pytest

from oracle import ref_audio as ra


def _audio_mod():
    import omx_import
    omx_import.load_package()
    from ominix_mlx_amd import audio
    return audio


def test_slaney_filterbank_matches_transformers():
    au = pytest.importorskip("transformers.audio_utils")
    hf = au.mel_filter_bank(num_frequency_bins=201, num_mel_filters=128, min_frequency=0.0, max_frequency=8000.0,
                            sampling_rate=16000, norm="slaney", mel_scale="slaney").T
    fb = ra.whisper_mel_filterbank(16000, 400, 128)
    assert fb.shape == (128, 201)
    assert np.abs(hf - fb).max() <= 1e-6 * max(1.0, np.abs(hf).max() / 0.04)
    assert abs(ra.hz_to_slaney_mel(0.0)) < 1e-6 and abs(ra.hz_to_slaney_mel(1000.0) - 15.0) < 1e-4   # 1000 Hz = 15 mel (Slaney)


@pytest.mark.parametrize("bits,channels,extra", [(16, 1, False), (16, 2, True), (24, 1, True), (32, 1, False), (32, 2, False)])
def test_wav_container_load(tmp_path, bits, channels, extra):
    audio = _audio_mod()
    g = np.random.default_rng(bits + channels)
    x = (g.uniform(-0.9, 0.9, size=(1500, channels))).astype(np.float32)
    buf = ra.wav_bytes(x, 22050, bits, channels, extra)
    path = tmp_path / "a.wav"
    path.write_bytes(buf)
    got, sr = audio.load_wav(path)
    want, sr2 = ra.load_wav_bytes(buf)
    assert sr == sr2 == 22050 and got.shape == (1500,)
    np.testing.assert_array_equal(got, want)
    tol = {16: 2.0 ** -14, 24: 2.0 ** -22, 32: 0.0}[bits]      # the helper writes x * (2^(b-1) - 1) truncated, the loader divides by 2^(b-1)
    np.testing.assert_allclose(got, x.mean(axis=1), atol=tol + 1e-7)


def test_wav_container_errors_and_save_round_trip(tmp_path):
    audio = _audio_mod()
    p = tmp_path / "bad.wav"
    p.write_bytes(b"RIFX" + b"\0" * 40)
    with pytest.raises(ValueError, match="Not a RIFF file"):
        audio.load_wav(p)
    p.write_bytes(b"RIFF\0\0\0\0WAVX" + b"\0" * 40)
    with pytest.raises(ValueError, match="Not a WAVE file"):
        audio.load_wav(p)
    buf = bytearray(ra.wav_bytes(np.zeros(10, np.float32), 8000, 16))
    buf[34:36] = (8).to_bytes(2, "little")
    p.write_bytes(bytes(buf))
    with pytest.raises(ValueError, match="Unsupported bits per sample: 8"):
        audio.load_wav(p)
    x = np.array([0.0, 0.5, -0.5, 1.5, -2.0, 0.25], np.float32)
    out = tmp_path / "o.wav"
    audio.save_wav(x, 16000, out)
    raw = out.read_bytes()
    assert raw[:4] == b"RIFF" and int.from_bytes(raw[4:8], "little") == 36 + 12 and raw[36:40] == b"data"
    y, sr = audio.load_wav(out)
    assert sr == 16000
    np.testing.assert_array_equal(y, np.trunc(np.clip(x, -1, 1) * np.float32(32767.0)).astype(np.float32) / np.float32(32768.0))


def test_whisper_oracle_shapes_and_errors():
    x = ra.generate_speech_like(16000, 0.5)
    m = ra.whisper_log_mel(x)
    assert m.shape == (128, 1 + (x.size - 400) // 160)
    assert m.max() - m.min() <= 2.0 + 1e-6                      # clip at max - 8, then / 4
    with pytest.raises(ValueError):
        ra.whisper_log_mel(np.zeros(0, np.float32))
    with pytest.raises(ValueError):
        ra.whisper_log_mel(np.zeros(399, np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sine_440", "mixed", "speech_like", "noise_lcg"])
def test_whisper_log_mel_matches_oracle(omx, kind):
    """Tolerance: log-mel abs < 1e-3 before the /4 of the Whisper normalisation, i.e. 2.5e-4 here --
    the device sums the 400-pt DFT and the filter products in float32, the oracle in float64."""
    from ominix_mlx_amd import audio
    sig = ra.signals(16000, 1.0)[kind]
    fe = audio.WhisperMelFrontend()
    got = fe.compute_mel_spectrogram(sig).numpy()
    want = ra.whisper_log_mel(sig)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2.5e-4


@pytest.mark.gpu
def test_whisper_log_mel_long_audio_and_errors(omx):
    from ominix_mlx_amd import audio
    fe = audio.WhisperMelFrontend()
    x = np.tile(ra.generate_speech_like(16000, 1.0), 30)                       # 30 s
    got = fe.compute_mel_spectrogram(x).numpy()
    assert got.shape == (128, 2998)
    assert np.abs(got - ra.whisper_log_mel(x)).max() <= 2.5e-4
    with pytest.raises(omx.OmxError, match="empty"):
        fe.compute_mel_spectrogram(np.zeros(0, np.float32))
    with pytest.raises(omx.OmxError, match="too short"):
        fe.compute_mel_spectrogram(np.zeros(399, np.float32))
