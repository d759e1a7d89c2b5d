"""CPU: pins oracle/ref_audio.py by the reference's own criterion and fixtures
(funasr-mlx/examples/validate_correctness.rs: 4 deterministic signals + LCG noise; FFT-STFT vs
direct-DFT STFT must agree to relative-L2 < 1e-5, :284-287), plus the in-tree sanity tests
(mlx-rs-core/src/audio.rs:690-702 hz_to_mel; paraformer.rs:1550 560-dim CMVN)."""
import numpy as np
import pytest

from oracle import ref_audio as ra


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.sqrt(((a - b) ** 2).sum() / max((a ** 2).sum(), 1e-300))


@pytest.mark.parametrize("name", ["sine_440", "mixed", "speech_like", "noise_lcg"])
def test_stft_matches_reference_direct_dft_criterion(name):
    sig = ra.signals(16000, 0.25)[name]        # 0.25 s: the O(N^2) reference DFT stays fast
    cfg = ra.ParaformerFrontendConfig
    w = ra.hamming_window(cfg.n_fft)
    fast = ra.stft_power(sig, w, cfg.n_fft, cfg.hop_length)
    direct = ra.stft_power_direct_dft_f32(sig, w, cfg.n_fft, cfg.hop_length)
    assert fast.shape == direct.shape == ((len(sig) - 400) // 160 + 1, 201)
    assert rel_l2(direct, fast) < 1e-5


def test_lcg_noise_generator_first_values():
    """validate_correctness.rs:448-458: seed 12345, x = (seed >> 16)/32768 - 1 with u64 wrapping."""
    n = ra.generate_noise(16000, 0.001)
    s = (12345 * 1103515245 + 12345) & 0xFFFFFFFFFFFFFFFF
    assert n[0] == np.float32(np.float32(s >> 16) / np.float32(32768.0)) - np.float32(1.0)
    assert len(n) == 16


def test_mel_scale_and_filterbank_properties():
    assert ra.hz_to_mel(0.0) == 0.0
    assert abs(float(ra.hz_to_mel(1000.0)) - 1000.0) < 1.0               # audio.rs:697-702
    fb = ra.create_mel_filterbank(400, 80, 16000.0)
    assert fb.shape == (80, 201) and fb.min() >= 0 and fb.max() <= 1.0
    assert (fb.sum(axis=1) > 0).all()                                    # every filter covers a bin
    w = ra.hamming_window(400)
    assert abs(w[0] - 0.08) < 1e-6 and abs(w[-1] - 0.08) < 1e-5 and abs(w.max() - 1.0) < 1e-4


def test_frontend_shapes_30s_and_edge_cases():
    cfg = ra.ParaformerFrontendConfig
    out = ra.mel_frontend(np.zeros(480000, np.float32))
    assert out["logmel"].shape == (2998, 80) and out["feats"].shape == (501, 560)
    assert np.allclose(out["logmel"], np.log(1e-10))
    short = ra.mel_frontend(np.ones(100, np.float32))                     # < n_fft: one all-zero frame
    assert short["feats"].shape == (1, 560) and np.allclose(short["feats"], np.log(1e-10))
    idx = ra.lfr_indices(10, cfg.lfr_m, cfg.lfr_n)
    assert idx.shape == (3, 7) and idx[0].tolist() == [0, 0, 0, 0, 1, 2, 3] and idx[-1].tolist() == [9] * 7
    with pytest.raises(ValueError):
        ra.mel_frontend(np.array([0.0, np.nan], np.float32))
