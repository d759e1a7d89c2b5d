"""CPU: properties of oracle/ref_paraformer.py that the reference itself checks
(examples/validate_correctness.rs part 2 / compare_cif_batch.rs: batched CIF == per-item CIF)."""
import numpy as np

from oracle import ref_paraformer as rp


def test_cif_batch_equals_single_and_conserves_mass():
    g = np.random.default_rng(0)
    B, T, H = 3, 120, 16
    hidden = g.standard_normal((B, T, H)).astype(np.float32)
    alphas = (g.random((B, T)) * 0.4).astype(np.float32)
    frames, counts = rp.cif_fire(hidden, alphas)
    for b in range(B):
        f1, c1 = rp.cif_fire(hidden[b:b + 1], alphas[b:b + 1])
        assert c1[0] == counts[b]
        np.testing.assert_array_equal(f1[0], frames[b, :counts[b]])
    # every fired frame integrates exactly one unit of alpha: count == floor(sum alpha) (+1 if the tail > 0.45)
    for b in range(B):
        s = float(alphas[b].astype(np.float64).sum())
        assert counts[b] in (int(np.floor(s)), int(np.floor(s)) + 1)


def test_cif_known_small_case():
    hidden = np.array([[[1.0], [2.0], [4.0]]], np.float32)
    alphas = np.array([[0.6, 0.6, 0.5]], np.float32)
    frames, counts = rp.cif_fire(hidden, alphas)
    # t0: 0.6 ; t1: fires with completion 0.4 -> 0.6*1 + 0.4*2 = 1.4, remainder 0.2*2 ; t2: 0.2+0.5=0.7 < 1 ; tail 0.7 > 0.45
    assert counts.tolist() == [2]
    np.testing.assert_allclose(frames[0, :, 0], [1.4, 0.2 * 2 + 0.5 * 4], rtol=1e-6)


def test_fsmn_is_centered_depthwise_conv():
    v = np.zeros((9, 2)); v[4, 0] = 1.0; v[4, 1] = 2.0
    w = np.zeros((2, 3)); w[0] = [1, 2, 3]; w[1] = [0, 1, 0]
    out = rp.fsmn(v, w)
    np.testing.assert_array_equal(out[:, 0], [0, 0, 0, 3, 2, 1, 0, 0, 0])      # correlation, not convolution (as Conv1d)
    np.testing.assert_array_equal(out[:, 1], v[:, 1])
