"""GPU parity of the Paraformer body pieces (a13) against oracle/ref_paraformer.py.
Tolerance: the reference path is float32; the MI355X build keeps activations in bf16, so the encoder
layer is compared at 2^-6 * max|ref|; CIF is float32 on both sides (1e-5 relative)."""
import numpy as np
import pytest

from oracle import ref_core as rc, ref_paraformer as rp

pytestmark = pytest.mark.gpu


def _weights(in_dim, dim, ffn, k, seed):
    g = np.random.default_rng(seed)
    r = lambda *s, sc=1.0: rc.bf16_round((g.standard_normal(s) * sc).astype(np.float32))
    return {"norm1_w": r(in_dim, sc=0.1) + 1, "norm1_b": r(in_dim, sc=0.1), "qkv_w": r(3 * dim, in_dim, sc=0.05), "qkv_b": r(3 * dim, sc=0.1),
            "out_w": r(dim, dim, sc=0.05), "out_b": r(dim, sc=0.1), "fsmn_w": r(dim, k, sc=0.2), "norm2_w": r(dim, sc=0.1) + 1,
            "norm2_b": r(dim, sc=0.1), "ffn_up_w": r(ffn, dim, sc=0.05), "ffn_up_b": r(ffn, sc=0.1), "ffn_down_w": r(dim, ffn, sc=0.05),
            "ffn_down_b": r(dim, sc=0.1)}


@pytest.mark.parametrize("T,in_dim", [(101, 512), (501, 512), (77, 560)])      # 560: the first layer (no attention residual)
def test_sanm_encoder_layer_matches_oracle(omx, T, in_dim):
    from ominix_mlx_amd import paraformer
    dim, ffn, heads, k = 512, 2048, 4, 11
    w = _weights(in_dim, dim, ffn, k, 3)
    for key in ("norm1_w", "norm2_w"):
        w[key] = rc.bf16_round(w[key])
    x = rc.bf16_round(np.random.default_rng(4).standard_normal((T, in_dim)).astype(np.float32))
    ref = rp.sanm_encoder_layer(x, w, heads)
    got = paraformer.SanmEncoderLayer(w, heads, k).forward(omx.ops.Tensor.from_numpy(x)).numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2.0 ** -6 * np.abs(ref).max()


def test_cif_fire_matches_oracle(omx):
    from ominix_mlx_amd import paraformer
    g = np.random.default_rng(5)
    B, T, H = 2, 501, 512
    hidden = g.standard_normal((B, T, H)).astype(np.float32)
    alphas = (g.random((B, T)) * 0.35).astype(np.float32)
    alphas[1, 300:] = 0.0                                             # ragged: the second item fires fewer tokens
    ref_frames, ref_counts = rp.cif_fire(hidden, alphas)
    T_ = omx.ops.Tensor
    frames, counts = paraformer.cif_fire(T_.from_numpy(hidden, "f32"), T_.from_numpy(alphas, "f32"))
    np.testing.assert_array_equal(counts, ref_counts)
    np.testing.assert_allclose(frames, ref_frames, rtol=1e-5, atol=1e-5)
