"""GPU parity of the audio surroundings: the windowed-sinc resampler (csrc/resample.hip vs oracle/ref_resample.py)
and the Fun-ASR-Nano / SenseVoice frontend + LFR (csrc/mel.hip vs oracle/ref_audio.py).

Tolerances: resampler -- the read positions are computed identically (host float64 recurrence), the 4 x 256-tap dot products are f32 in
a different summation order and the phase table uses libm cosf / sinf instead of numpy's: |diff| <= 2e-5 on unit-amplitude signals,
lengths EQUAL.  Frontend -- the criterion tests/test_gpu_audio.py holds the Paraformer frontend to: log-mel abs < 1e-3 on every bin within
50 dB of its frame's strongest bin and |d mel| <= 2e-5 x frame peak on all bins (an f32 DFT -- the reference's rustfft included --
carries ~1e-7 x peak of rounding noise, which is the whole content of the leakage-only bins of a pure tone: their LOG is not comparable)."""
import numpy as np
import pytest

from oracle import ref_audio as ra, ref_resample as rr

pytestmark = pytest.mark.gpu


def _check_logmel(got, ref):
    """got / ref: [n_mels, n_frames]."""
    got, ref = got.T, ref.T
    strong = ref >= ref.max(axis=1, keepdims=True) - np.log(1e5)
    assert np.abs(got - ref)[strong].max() < 1e-3
    peak = np.exp(ref.astype(np.float64)).max(axis=1, keepdims=True)
    assert (np.abs(np.exp(got.astype(np.float64)) - np.exp(ref.astype(np.float64))) <= 2e-5 * peak + 1e-9).all()


@pytest.mark.parametrize("src,dst,n", [(48000, 16000, 48000 * 3 + 17), (44100, 16000, 44100), (8000, 16000, 9000), (22050, 16000, 30011),
                                        (16000, 24000, 4096 * 3), (32000, 16000, 5000), (16000, 32000, 100), (48000, 16000, 300)])
def test_resample_matches_oracle(omx, src, dst, n):
    from ominix_mlx_amd import audio
    g = np.random.default_rng(n)
    t = np.arange(n) / src
    x = (0.6 * np.sin(2 * np.pi * 310.0 * t) + 0.3 * np.sin(2 * np.pi * 2900.0 * t) + 0.05 * g.standard_normal(n)).astype(np.float32)
    want = rr.resample(x, src, dst)
    got = audio.resample(x, src, dst)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5


def test_resample_reference_tests_and_identity(omx):
    from ominix_mlx_amd import audio
    x = np.sin(np.arange(100, dtype=np.float32) * np.float32(0.1))       # mlx-rs-core/src/audio.rs:705-710
    assert len(audio.resample(x, 16000, 32000)) > len(x)
    x = np.sin(np.arange(48000, dtype=np.float32) / np.float32(48000))   # funasr-qwen4b-mlx/src/audio.rs:699-706
    assert 15000 <= len(audio.resample(x, 48000, 16000)) <= 17000
    five = np.array([1, 2, 3, 4, 5], np.float32)
    np.testing.assert_array_equal(audio.resample(five, 16000, 16000), five)
    assert audio.resample(np.zeros(0, np.float32), 48000, 16000).size == 0


def test_resample_30s_full_size_properties(omx):
    """BASELINE config 4 size (30 s): 48 kHz -> 16 kHz of a band-limited sine, checked without the oracle through the resampler's
    defining property, plus device-to-device use."""
    from ominix_mlx_amd import audio
    from ominix_mlx_amd.ops import Tensor
    src, dst, n, f0 = 48000, 16000, 48000 * 30, 1000.0
    x = np.sin(2 * np.pi * f0 * np.arange(n) / src).astype(np.float32)
    y = audio.resample_device(Tensor.from_numpy(x, "f32"), src, dst).numpy().ravel()
    assert len(y) == 16000 * 30
    t = rr.output_time(np.arange(len(y)), dst / src)
    m = (t > 200) & (t < n - 400)
    assert np.abs(y[m] - np.sin(2 * np.pi * f0 * t[m] / src)).max() < 1e-4


@pytest.mark.parametrize("name", ["sine_440", "noise_lcg", "mixed", "speech_like"])
def test_sensevoice_mel_and_lfr_match_oracle(omx, name):
    from ominix_mlx_amd import audio
    x = np.asarray(ra.signals(16000, 1.0)[name], np.float32)
    fe = audio.SenseVoiceMelFrontend()
    assert fe.n_freqs == 201                                              # funasr-nano-mlx/src/audio.rs:439-443
    mel = fe.compute_mel_spectrogram(x)
    want = ra.sensevoice_log_mel(x)
    assert mel.shape == (1, 80, 100)
    got = mel.numpy()[0]
    _check_logmel(got, want)
    lfr = audio.apply_lfr(mel, 7, 6)
    assert lfr.shape == (1, 17, 560)
    np.testing.assert_array_equal(lfr.numpy()[0], ra.apply_lfr(got, 7, 6))      # pure data movement: exact


def test_sensevoice_edge_cases(omx):
    from ominix_mlx_amd import audio
    fe = audio.SenseVoiceMelFrontend()
    with pytest.raises(Exception, match="empty"):
        fe.compute_mel_spectrogram(np.zeros(0, np.float32))
    with pytest.raises(Exception, match="too short"):
        fe.compute_mel_spectrogram(np.zeros(10, np.float32))
    g = np.random.default_rng(1)
    short = g.standard_normal(200).astype(np.float32)                     # shorter than a window: one zero-padded frame
    _check_logmel(fe.compute_mel_spectrogram(short).numpy()[0], ra.sensevoice_log_mel(short))
    long = (0.1 * g.standard_normal(16000 * 31)).astype(np.float32)       # 30 s cap (audio.rs:112-117)
    mel = fe.compute_mel_spectrogram(long)
    assert mel.shape == (1, 80, 3000)
    _check_logmel(mel.numpy()[0], ra.sensevoice_log_mel(long))
    # the reference's LFR test (audio.rs:424-436): [1, 80, 100] -> [1, 17, 560]
    from ominix_mlx_amd.ops import Tensor
    m = (np.arange(8000, dtype=np.float32) * np.float32(0.001)).reshape(1, 80, 100)
    lfr = audio.apply_lfr(Tensor.from_numpy(m, "f32"), 7, 6)
    assert lfr.shape == (1, 17, 560)
    np.testing.assert_array_equal(lfr.numpy()[0], ra.apply_lfr(m[0], 7, 6))
