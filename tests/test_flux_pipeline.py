"""The sampling loop around the FLUX.2-klein DiT.  CPU part: the oracle restatement
(oracle/ref_klein.py) against the reference's own tests (flux-klein-mlx/src/sampler.rs:352-407) and the host module
(ominix-mlx_amd/flux_pipeline.py, written independently) against the oracle -- scalar float32 arithmetic, so equality
is exact.  GPU part: prior noise, Euler update and a whole short denoising run against the oracle."""
import numpy as np
import pytest

from oracle import mlx_rng as rng
from oracle import ref_core as rc
from oracle import ref_klein as rk


def _host():
    import omx_import
    omx_import.load_package()
    from ominix_mlx_amd import flux_pipeline
    return flux_pipeline


def test_reference_sampler_kats_on_the_oracle():
    ts = rk.sampler_timesteps(4, True)                                   # sampler.rs:364-371
    assert len(ts) == 5 and abs(ts[0] - 1.0) < 1e-6 and abs(ts[4]) < 1e-6
    data = np.array([1.0, 2.0, 3.0, 4.0], np.float32).reshape(1, 2, 2)   # sampler.rs:374-387
    assert np.abs(rk.add_noise(data, np.zeros_like(data), np.array([0.0], np.float32)) - data).max() < 1e-6
    got = rk.sampler_step(np.array([[[1.0, 2.0]]], np.float32), np.array([[[0.5, 0.5]]], np.float32), 1.0, 0.75)
    assert np.abs(got - np.array([[[0.875, 1.875]]], np.float32)).max() < 1e-5   # sampler.rs:390-407


@pytest.mark.parametrize("seq,steps", [(1024, 4), (4096, 4), (4096, 28), (4608, 50), (256, 1)])
def test_official_schedule_properties_and_host_equals_oracle(seq, steps):
    fp = _host()
    want = rk.official_schedule(steps, seq)
    got = fp.official_schedule(steps, seq)
    assert [float(x) for x in got] == [float(x) for x in want]
    assert got[0] == 1.0 and got[-1] == 0.0 and all(a > b for a, b in zip(got, got[1:]))
    assert float(fp.compute_empirical_mu(seq, steps)) == float(rk.compute_empirical_mu(seq, steps))
    # the SNR shift spends more of the trajectory at high noise than the linear schedule does
    if steps >= 4:
        assert got[steps // 2] > 0.5


def test_host_sampler_equals_oracle_and_reference_defaults():
    fp = _host()
    assert (fp.FluxSamplerConfig.schnell().num_steps, fp.FluxSamplerConfig.schnell().is_schnell) == (4, True)   # sampler.rs:353-361
    assert (fp.FluxSamplerConfig.dev().num_steps, fp.FluxSamplerConfig.dev().is_schnell) == (50, False)
    for cfg, schnell in ((fp.FluxSamplerConfig.schnell(), True), (fp.FluxSamplerConfig.dev(), False)):
        got = fp.FluxSampler(cfg).timesteps(7)
        want = rk.sampler_timesteps(7, schnell, cfg.shift)
        assert [float(x) for x in got] == [float(x) for x in want]
    g = np.random.default_rng(0)
    x, n = g.standard_normal((2, 5, 3)).astype(np.float32), g.standard_normal((2, 5, 3)).astype(np.float32)
    t = np.array([0.25, 0.9], np.float32)
    np.testing.assert_array_equal(fp.FluxSampler.add_noise(x, n, t), rk.add_noise(x, n, t))
    np.testing.assert_array_equal(fp.FluxSampler.step(x, n, 0.8, 0.55), rk.sampler_step(x, n, 0.8, 0.55))


def test_unpack_latents_layout():
    fp = _host()
    ph, pw, z, p = 3, 5, 32, 2
    lat = np.arange(ph * pw * z * p * p, dtype=np.float32).reshape(ph * pw, z * p * p)
    got = fp.unpack_latents(lat, ph, pw)
    np.testing.assert_array_equal(got, rk.unpack_latents(lat, ph, pw))
    assert got.shape == (ph * p, pw * p, z)
    # element (patch y, x; channel c; in-patch dy, dx) lands at pixel (2y + dy, 2x + dx), channel c
    y, x, c, dy, dx = 2, 4, 7, 1, 0
    assert got[2 * y + dy, 2 * x + dx, c] == lat[y * pw + x, (c * p + dy) * p + dx]


def test_normal_kat_and_erfinv_accuracy():
    assert float(rng.normal((1,), rng.key(0))[0]) == pytest.approx(-0.20, abs=0.01)      # mlx-rs/src/random.rs:582-586
    from scipy.special import erfinv
    x = np.linspace(-0.999999, 0.999999, 100001).astype(np.float32)
    want = erfinv(x.astype(np.float64))
    ulp = np.abs(rng.erfinv32(x).astype(np.float64) - want) / np.spacing(np.abs(want).astype(np.float32))
    assert ulp.max() <= 3.0
    z = rng.normal((100000,), rng.key(1))
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02


# ---------------------------------------------------------------- GPU ----------------------------------------------------------------

@pytest.mark.gpu
def test_device_normal_matches_oracle(omx):
    n = 200003
    got = omx.ops.random_normal(omx.ops.random_key(4), (n,)).numpy()
    want = rng.normal((n,), rng.key(4))
    same = got.view(np.uint32) == want.view(np.uint32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert same.mean() >= 0.999 and ulp.max() <= 2      # fused multiply-adds on the device, double-rounded ones in numpy
    assert float(omx.ops.random_normal(omx.ops.random_key(0), (1,)).numpy()[0]) == pytest.approx(-0.20, abs=0.01)
    sc = omx.ops.random_normal(omx.ops.random_key(4), (1000,), loc=2.0, scale=0.5).numpy()
    np.testing.assert_allclose(sc, rng.normal((1000,), rng.key(4), 2.0, 0.5), atol=1e-6)


@pytest.mark.gpu
def test_euler_step_kernel(omx):
    from ominix_mlx_amd import klein   # binds omx_klein_euler_step
    T = omx.ops.Tensor
    g = np.random.default_rng(1)
    z = g.standard_normal(5000).astype(np.float32)
    v = rc.bf16_round(g.standard_normal(5000).astype(np.float32))
    zt, vt, z16 = T.from_numpy(z, "f32"), T.from_numpy(v, "bf16"), T((5000,), "bf16")
    dt = np.float32(0.37) - np.float32(0.81)
    omx.check(omx.lib.omx_klein_euler_step(zt.ptr, vt.ptr, float(dt), z16.ptr, 5000, None))
    want = rk.euler_step(z, v, np.float32(0.81), np.float32(0.37)).astype(np.float32)
    np.testing.assert_array_equal(zt.numpy(), want)
    np.testing.assert_array_equal(z16.numpy(), rc.bf16_round(want))


@pytest.mark.gpu
def test_denoise_loop_matches_oracle(omx):
    """generate_klein.rs:412-446 end to end on the tiny DiT: same prior (the device normal draw of key
    RandomState(seed).next()), 3 Euler steps of the official schedule.  Tolerance: the forward's own bound
    (tests/test_gpu_klein.py) times the sum of |dt| = 1 over the trajectory, plus the bf16 copy the DiT reads."""
    from ominix_mlx_amd import flux_pipeline, klein
    p = rk.KleinParams.tiny()
    weights = rk.synth_weights(p)
    oracle = rk.KleinOracle(p, weights)
    height = width = 16 * 6                      # 6 x 6 patches
    s_txt, steps, seed = 24, 3, 11
    txt = rc.bf16_round(np.random.default_rng(3).standard_normal((s_txt, p.txt_embed_dim)).astype(np.float32))
    m = klein.FluxKlein(p.in_channels, p.hidden_size, p.txt_embed_dim, p.num_heads, p.depth, p.depth_single, p.head_dim, p.mlp_hidden)
    m.synth_weights()
    seen = []
    got = flux_pipeline.denoise(m, omx.ops.Tensor.from_numpy(txt), height, width, steps, seed,
                                on_step=lambda i, a, b, ms: seen.append((i, a, b)))
    noise = rng.normal((36, p.in_channels), rng.RandomState(seed).next())
    want = rk.denoise(oracle, txt, 6, 6, steps, noise)
    assert [s[0] for s in seen] == [0, 1, 2] and seen[0][1] == 1.0 and seen[-1][2] == 0.0
    # per-step velocity error bound of the forward, accumulated with weights |dt_i| (sum = 1)
    v_scale = max(np.abs(oracle.forward_with_rope(noise, txt, 1000.0, *rk.compute_rope(
        np.concatenate([rk.create_txt_ids(s_txt), rk.create_img_ids(6, 6)], 0)))).max(), 1.0)
    bound = 2.0 ** -6 * v_scale * np.sqrt(p.depth + p.depth_single) * 2 + 2.0 ** -8 * np.abs(want).max()
    assert got.shape == want.shape and np.abs(got - want).max() <= bound
    np.testing.assert_array_equal(flux_pipeline.unpack_latents(got, 6, 6), rk.unpack_latents(got, 6, 6))
