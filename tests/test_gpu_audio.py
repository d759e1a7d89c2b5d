"""GPU parity of the mel/STFT frontend (a12) against oracle/ref_audio.py on the reference's
deterministic signals.  Tolerances: STFT power relative-L2 < 1e-5 (the reference's
own acceptance bound, validate_correctness.rs:284-287); log-mel / features abs < 1e-3 on every bin
whose energy is within 50 dB of the frame's strongest bin, and |d mel| <= 2e-5 x frame peak on all
bins (an f32 DFT -- the reference's rustfft included -- carries ~1e-7 x peak of rounding noise, which
is the whole content of the leakage-only bins of a pure tone, so their LOG is not comparable)."""
import numpy as np
import pytest

from oracle import ref_audio as ra
from test_audio_oracle import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def audio(omx):
    from ominix_mlx_amd import audio
    return audio


def _strong(ref_logmel):
    return ref_logmel >= ref_logmel.max(axis=1, keepdims=True) - np.log(1e5)


def _check_logmel(got, ref):
    strong = _strong(ref)
    assert np.abs(got - ref)[strong].max() < 1e-3
    peak = np.exp(ref.astype(np.float64)).max(axis=1, keepdims=True)
    assert (np.abs(np.exp(got.astype(np.float64)) - np.exp(ref.astype(np.float64))) <= 2e-5 * peak + 1e-9).all()


@pytest.mark.parametrize("name", ["sine_440", "mixed", "speech_like", "noise_lcg"])
def test_frontend_matches_oracle(audio, name):
    sig = ra.signals(16000, 1.0)[name]
    g = np.random.default_rng(3)
    addshift, rescale = (-8 + g.standard_normal(560) * 0.1).astype(np.float32), (0.1 + g.random(560) * 0.05).astype(np.float32)
    fe = audio.MelFrontend()
    fe.set_cmvn(addshift, rescale)
    feats, logmel, power = fe.forward(sig, return_intermediates=True)
    ref = ra.mel_frontend(sig, addshift, rescale)
    assert feats.shape == (1,) + ref["feats"].shape
    assert rel_l2(ref["power"], power.numpy()) < 1e-5
    _check_logmel(logmel.numpy(), ref["logmel"])
    idx = ra.lfr_indices(ref["logmel"].shape[0], 7, 6)
    strong = _strong(ref["logmel"])[idx].reshape(idx.shape[0], -1)
    d = np.abs(feats.numpy()[0] - ref["feats"]) / rescale[None, :]       # undo the CMVN scale for the bound
    assert d[strong].max() < 1e-3


def test_frontend_30s_shapes_and_no_cmvn(audio):
    sig = np.tile(ra.signals(16000, 1.0)["speech_like"], 30)
    fe = audio.MelFrontend()
    assert fe.frames(len(sig)) == (2998, 501)
    feats = fe.forward(sig).numpy()
    ref = ra.mel_frontend(sig)["feats"]
    assert feats.shape == (1, 501, 560)
    full = ra.mel_frontend(sig)
    strong = _strong(full["logmel"])[ra.lfr_indices(2998, 7, 6)].reshape(501, -1)
    assert np.abs(feats[0] - ref)[strong].max() < 1e-3


def test_frontend_edge_cases(audio, omx):
    fe = audio.MelFrontend()
    short = fe.forward(np.ones(100, np.float32)).numpy()
    assert short.shape == (1, 1, 560) and np.allclose(short, np.log(1e-10), atol=1e-5)
    exact = fe.forward(np.ones(400, np.float32)).numpy()          # exactly one frame
    np.testing.assert_allclose(exact[0], ra.mel_frontend(np.ones(400, np.float32))["feats"], atol=1e-3)
    with pytest.raises(omx.OmxError, match="NaN or Inf"):
        fe.forward(np.array([0.0, np.inf] * 300, np.float32))
    with pytest.raises(omx.OmxError, match="560"):
        fe.set_cmvn(np.zeros(80), np.ones(80))
