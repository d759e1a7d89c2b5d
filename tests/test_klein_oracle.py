"""CPU: sanity / KATs of oracle/ref_klein.py that the reference's own tests pin
(flux-klein-mlx/src/sampler.rs:390-407 Euler step; layers.rs:931-935 timestep_embedding shape)."""
import numpy as np

from oracle import ref_klein as rk


def test_euler_step_kat():
    """sampler.rs test_step: [1, 2] + (0.75 - 1.0) * [0.5, 0.5] = [0.875, 1.875]."""
    out = rk.euler_step(np.array([1.0, 2.0]), np.array([0.5, 0.5]), 1.0, 0.75)
    np.testing.assert_allclose(out, [0.875, 1.875], rtol=0, atol=0)


def test_timestep_embedding_shape_and_layout():
    e = rk.timestep_embedding(500.0, 256)
    assert e.shape == (1, 256)
    assert abs(e[0, 0] - np.cos(500.0)) < 1e-6 and abs(e[0, 128] - np.sin(500.0)) < 1e-6      # [cos | sin], freq_0 = 1


def test_rope_tables_and_rotation_is_norm_preserving():
    ids = np.concatenate([rk.create_txt_ids(5), rk.create_img_ids(3, 4)], 0)
    cos, sin = rk.compute_rope(ids)
    assert cos.shape == (17, 128)
    np.testing.assert_array_equal(cos[:, 0::2], cos[:, 1::2])                                  # duplicated pairs
    np.testing.assert_allclose(cos[:5, :96], 1.0)                                              # text: only the 4th axis moves
    x = np.random.default_rng(0).standard_normal((17, 2, 128))
    y = rk.apply_rope(x, cos, sin)
    np.testing.assert_allclose((y ** 2).sum(-1), (x ** 2).sum(-1), rtol=1e-6)


def test_forward_shapes_tiny():
    p = rk.KleinParams.tiny()
    w = rk.synth_weights(p)
    o = rk.KleinOracle(p, w)
    g = np.random.default_rng(1)
    ids = np.concatenate([rk.create_txt_ids(6), rk.create_img_ids(2, 3)], 0)
    cos, sin = rk.compute_rope(ids)
    out = o.forward_with_rope(g.standard_normal((6, 128)), g.standard_normal((6, 512)), 500.0, cos, sin)
    assert out.shape == (6, 128) and np.isfinite(out).all()
