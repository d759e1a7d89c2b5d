"""oracle/ref_audio.py, Fun-ASR-Nano / SenseVoice frontend, on the CPU: the reference's own tests (funasr-nano-mlx/src/audio.rs:418-484 --
filterbank size, LFR dimensions, spectrogram shape, the two error cases) plus the structure of the restated pieces."""
import numpy as np
import pytest

from oracle import ref_audio as ra


def test_mel_filterbank():                       # audio.rs:418-421
    fb = ra.sensevoice_mel_filterbank(16000, 400, 80)
    assert fb.size == 80 * 201
    assert fb.min() >= 0.0 and fb.max() <= 1.0
    assert (fb.sum(axis=1) > 0).sum() >= 70      # low filters can be empty: two mel points in the same FFT bin


def test_lfr_dimensions():                        # audio.rs:424-436
    mel = (np.arange(8000, dtype=np.float32) * np.float32(0.001)).reshape(80, 100)
    lfr = ra.apply_lfr(mel, 7, 6)
    assert lfr.shape == (17, 560)
    # frame 0: three copies of source frame 0 (left clamp), then frames 0..3;  last frame: centre 96, right edge clamped to 99
    np.testing.assert_array_equal(lfr[0, :80], mel[:, 0])
    np.testing.assert_array_equal(lfr[0, 3 * 80:4 * 80], mel[:, 0])
    np.testing.assert_array_equal(lfr[0, 6 * 80:], mel[:, 3])
    np.testing.assert_array_equal(lfr[16, 6 * 80:], mel[:, 99])
    np.testing.assert_array_equal(lfr[16, 0:80], mel[:, 93])


def test_mel_spectrogram_basic():                 # audio.rs:446-463
    x = np.sin(2 * np.pi * 440.0 * np.arange(16000, dtype=np.float32) / 16000).astype(np.float32)
    mel = ra.sensevoice_log_mel(x)
    assert mel.shape == (80, 100)
    band = mel[:, 10:90].mean(axis=1)
    fb = ra.sensevoice_mel_filterbank()
    assert fb[int(np.argmax(band)), 11] > 0      # 440 Hz = FFT bin 11 of 400 at 16 kHz: the loudest band is a filter covering it
    assert np.isfinite(mel).all()


def test_errors_and_truncation():                 # audio.rs:466-484, :112-117
    with pytest.raises(ValueError, match="empty"):
        ra.sensevoice_log_mel(np.zeros(0, np.float32))
    with pytest.raises(ValueError, match="too short"):
        ra.sensevoice_log_mel(np.zeros(10, np.float32))
    x = np.random.default_rng(0).standard_normal(16000 * 31).astype(np.float32)
    assert ra.sensevoice_log_mel(x).shape == (80, 3000)                   # 30 s cap
    assert ra.sensevoice_log_mel(x[:200]).shape == (80, 1)                # shorter than a window: one zero-padded frame
