"""oracle/ref_resample.py on the CPU: the reference's own tests of `audio::resample` (length properties -- the only ones it has:
mlx-rs-core/src/audio.rs:705-710, step-audio2-mlx/src/audio.rs:675-683, funasr-qwen4b-mlx/src/audio.rs:692-706), its identity cases,
the driver's length rule, and the property any windowed-sinc resampler must have (rubato itself is not available: parity unpinned)."""
import numpy as np
import pytest

from oracle import ref_resample as rr


def test_reference_test_resample_upsample_small():
    x = np.sin(np.arange(100, dtype=np.float32) * np.float32(0.1))       # mlx-rs-core/src/audio.rs:705-710
    y = rr.resample(x, 16000, 32000)
    assert len(y) > len(x)
    assert len(y) == 140              # shorter than a filter: only the flush chunk produces output (idx -228 .. -158 in steps of 0.5)


def test_reference_test_resample_8000():
    x = np.sin(np.arange(8000, dtype=np.float32) * np.float32(0.1))      # step-audio2-mlx/src/audio.rs:675-683
    y = rr.resample(x, 16000, 32000)
    assert len(y) == 16000


def test_reference_test_resample_same_rate_and_empty():
    x = np.array([1.0, 2.0, 3.0, 4.0, 5.0], np.float32)                  # funasr-qwen4b-mlx/src/audio.rs:692-696
    np.testing.assert_array_equal(rr.resample(x, 16000, 16000), x)
    assert rr.resample(np.zeros(0, np.float32), 48000, 16000).size == 0  # audio.rs:179-181


def test_reference_test_resample_downsample():
    x = np.sin(np.arange(48000, dtype=np.float32) / np.float32(48000))   # funasr-qwen4b-mlx/src/audio.rs:699-706
    y = rr.resample(x, 48000, 16000)
    assert 15000 <= len(y) <= 17000 and len(y) == 16000


@pytest.mark.parametrize("src,dst,n", [(48000, 16000, 50000), (44100, 16000, 44100), (8000, 16000, 9000), (22050, 16000, 30011),
                                        (16000, 24000, 4096 * 3), (32000, 16000, 5000)])
def test_bandlimited_sine_is_reproduced(src, dst, n):
    """Interior of the output == the same sine sampled at the new rate, at the time grid the table layout implies; the output has the
    reference's length round(n * dst / src)."""
    f0 = 440.0
    x = np.sin(2 * np.pi * f0 * np.arange(n) / src).astype(np.float32)
    y = rr.resample(x, src, dst)
    ratio = dst / src
    assert len(y) == int(np.floor(n * ratio + 0.5))
    t = rr.output_time(np.arange(len(y)), ratio)
    ref = np.sin(2 * np.pi * f0 * t / src)
    m = (t > 200) & (t < n - 400)          # away from the start-up and from the tail the driver cuts (audio.rs:239-256)
    assert m.sum() > 1000
    assert np.abs(y[m] - ref[m]).max() < 1e-4


def test_linearity_and_stopband():
    g = np.random.default_rng(3)
    a, b = g.standard_normal(12000).astype(np.float32), g.standard_normal(12000).astype(np.float32)
    ya, yb, yab = rr.resample(a, 48000, 16000), rr.resample(b, 48000, 16000), rr.resample(a + 2 * b, 48000, 16000)
    np.testing.assert_allclose(yab, ya + 2 * yb, atol=2e-5)
    # a tone above the new Nyquist (12 kHz at 16 kHz output) is removed: BlackmanHarris2 stopband
    x = np.sin(2 * np.pi * 12000 * np.arange(24000) / 48000).astype(np.float32)
    y = rr.resample(x, 48000, 16000)
    assert np.abs(y[300:-300]).max() < 1e-4


def test_sinc_table_properties():
    t = rr.make_sincs(256, 256, 0.95)
    assert t.shape == (256, 256) and t.dtype == np.float32
    assert abs(float(t.sum()) / 256 - 1.0) < 1e-4          # every phase sums to ~1: DC gain 1
    assert np.argmax(t[255]) == 128                         # phase 255 holds y[256 p]: centred on tap 128
