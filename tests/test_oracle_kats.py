"""Pins the CPU oracle (oracle/ref_core.py, oracle/mlx_rng.py) against every known-answer
test the reference holds for the hot-path primitives.  Each test
cites the reference test it reproduces; the reference asserts with 2 % tolerance, the
oracle reproduces the asserted values to ~1e-6 relative."""
import numpy as np
import pytest

from oracle import mlx_rng as rng
from oracle import ref_core as rc

REL = 2e-6


def _stats(a):
    a = np.asarray(a, dtype=np.float64)
    return a.mean(), a.sum()


def _draw(seed):
    rng.seed(seed)
    return rng.uniform(0.0, 1.0, (2, 8, 16))


@pytest.mark.parametrize("seed,mean,total", [
    (71, 0.5082664489746094, 130.1162109375),        # nn/positional_encoding.rs:437-445
    (22, 0.5029706, 128.76047),                      # nn/activation.rs:1296-1304
    (853, 0.5143963, 131.68546),                     # nn/activation.rs:1161-1169
    (744, 0.50868857, 130.22427),                    # nn/linear.rs:229-238
])
def test_rng_reproduces_reference_input_statistics(seed, mean, total):
    m, s = _stats(_draw(seed))
    assert m == pytest.approx(mean, rel=REL)
    assert s == pytest.approx(total, rel=REL)


def test_rope_kat_seed71():
    """mlx-rs/src/fast.rs:232-250 and nn/positional_encoding.rs:432-462."""
    a = _draw(71)
    out = rc.rope(a, dims=8, traditional=False, base=10000.0, scale=1.0, offset=0)
    m, s = _stats(out)
    assert out.shape == (2, 8, 16) and out.dtype == np.float32
    assert m == pytest.approx(0.45625377, rel=REL)
    assert s == pytest.approx(116.800964, rel=REL)
    # the interleaved ("traditional") pairing is excluded by the same KAT
    m_trad, _ = _stats(rc.rope(a, 8, True, 10000.0, 1.0, 0))
    assert abs(m_trad - 0.45625377) > 1e-3


def test_rms_norm_kat_seed103():
    """mlx-rs/src/fast.rs:254-273 and nn/normalization.rs:702-733."""
    a = _draw(103)
    out = rc.rms_norm(a, np.ones(16, np.float32), 1e-5)
    m, s = _stats(out)
    assert m == pytest.approx(0.87293875, rel=REL)
    assert s == pytest.approx(223.47232, rel=REL)


def test_layer_norm_kat_seed635():
    """mlx-rs/src/fast.rs:277-298 and nn/normalization.rs:666-699."""
    a = _draw(635)
    out = rc.layer_norm(a, np.ones(16, np.float32), np.zeros(16, np.float32), 1e-5)[..., 0]
    m, s = _stats(out)
    assert out.shape == (2, 8)
    assert m == pytest.approx(0.29099038, rel=5e-6)
    assert s == pytest.approx(4.655846, rel=5e-6)


def test_silu_kat_seed22():
    """mlx-rs/src/nn/activation.rs:1291-1320."""
    out = rc.silu(_draw(22))
    m, s = _stats(out)
    assert m == pytest.approx(0.33197093, rel=REL)
    assert s == pytest.approx(84.98456, rel=REL)


def test_softmax_kat_seed853():
    """mlx-rs/src/nn/activation.rs:1156-1180: rows sum to 1."""
    out = rc.softmax(_draw(853), axis=-1)
    m, s = _stats(out)
    assert m == pytest.approx(0.0625, rel=1e-6)
    assert s == pytest.approx(16.0, rel=1e-6)


def test_linear_kat_seed744():
    """mlx-rs/src/nn/linear.rs:224-252: input, then Linear::new(16,5) draws
    W ~ U(-1/4, 1/4)[5,16], then b ~ U(-1/4, 1/4)[5]; y = x W^T + b."""
    a = _draw(744)
    k = np.sqrt(1.0 / 16.0)
    w = rng.uniform(-k, k, (5, 16))
    b = rng.uniform(-k, k, (5,))
    out = rc.linear(a, w, b)
    m, s = _stats(out)
    assert out.shape == (2, 8, 5)
    assert m == pytest.approx(0.10419309, rel=REL)
    assert s == pytest.approx(8.335447, rel=REL)


def test_matmul_kat_exact():
    """mlx-rs/src/ops/arithmetic.rs:1921-1937."""
    a = np.array([[1, 2], [3, 4]], np.float32)
    b = np.array([[-5, 37.5, 4], [7, 1, 0]], np.float32)
    np.testing.assert_array_equal(rc.matmul(a, b).ravel(), np.array([9, 39.5, 4, 13, 116.5, 12], np.float32))


@pytest.mark.parametrize("bits", [2, 4, 8])
def test_quantize_dequantize_bound(bits):
    """mlx-rs/src/ops/quantization.rs:289-305: arange(512) rows x128, group 128."""
    w = np.tile(np.arange(512, dtype=np.float32), (128, 1))
    q, s, b = rc.quantize(w, group_size=128, bits=bits)
    assert q.shape == (128, 512 * bits // 32) and s.shape == (128, 4) and b.shape == (128, 4)
    w_hat = rc.dequantize(q, s, b, group_size=128, bits=bits)
    assert np.abs(w - w_hat).max() <= 127.0 / (1 << bits) + 1e-4


def test_bf16_round_is_rne():
    x = np.array([1.0, 1.00390625, 1.01171875, -1.00390625, 3.4e38], np.float32)
    # 1+2^-8 is a tie -> even (1.0); 1+3*2^-8 is a tie -> even (1+2^-6... i.e. 1.015625)
    out = rc.bf16_round(x)
    assert out[0] == 1.0 and out[1] == 1.0 and out[2] == 1.015625 and out[3] == -1.0
    assert np.isinf(out[4])
    bits = rc.to_bf16_bits(np.array([1.0, -2.0], np.float32))
    assert bits.tolist() == [0x3F80, 0xC000]
    np.testing.assert_array_equal(rc.from_bf16_bits(bits), np.array([1.0, -2.0], np.float32))


# ---- a10 sampler: the keyed random KATs of mlx-rs/src/random.rs (asserted there to 0.01 / exactly) ----

def test_uniform_kats_key0():
    k = rng.key(0)
    assert float(rng.uniform(0.0, 10.0, (1,), k)[0]) == pytest.approx(4.18, abs=0.01)            # random.rs:549-553
    np.testing.assert_allclose(rng.uniform(0.0, 10.0, (3,), k), [9.65, 3.14, 6.33], atol=0.01)    # random.rs:556-562


def test_split_is_deterministic_and_distinct():      # random.rs:532-541
    k1, k2 = rng.split(rng.key(0), 2)
    assert tuple(k1) != tuple(k2)
    r1, r2 = rng.split(rng.key(0), 2)
    assert tuple(r1) == tuple(k1) and tuple(r2) == tuple(k2)
    assert [tuple(map(int, x)) for x in rng.split2(rng.key(0))] == [tuple(map(int, k1)), tuple(map(int, k2))]


def test_gumbel_kat_key0():                          # random.rs:690-694
    assert float(rng.gumbel((1,), rng.key(0))[0]) == pytest.approx(0.13, abs=0.01)


def test_categorical_kats_key0():
    logits = np.zeros((5, 20), np.float32)
    np.testing.assert_array_equal(rng.categorical(logits, rng.key(0)), [1, 1, 17, 17, 17])       # random.rs:697-707
    np.testing.assert_array_equal(rng.categorical(logits, rng.key(0), num_samples=2),
                                  [[16, 3], [14, 10], [17, 7], [6, 8], [12, 8]])                 # random.rs:710-719


def test_sampler_temperature_zero_is_argmax_and_nonzero_is_categorical():   # mlx-rs-core/src/sampler.rs:9-18
    logits = np.array([[0.1, 2.0, -1.0, 2.0]], np.float32)
    np.testing.assert_array_equal(rc.sample(logits, 0.0, None), [1])
    k = rng.key(5)
    want = rng.categorical((logits * np.float32(1.0 / 0.7)).astype(np.float32), k)
    np.testing.assert_array_equal(rc.sample(logits, 0.7, k), want)
    # a very cold temperature concentrates on the (first) maximum
    np.testing.assert_array_equal(rc.sample(np.array([[0.0, 5.0, 1.0]], np.float32), 1e-3, k), [1])


def test_fused_modulate_literal_metal_restatement_vs_the_mathematical_one():
    """VERDICT r1 (oracle fidelity): `rc.fused_modulate` is the float64 function; the Metal kernel (metal_kernels.rs:28-94)
    accumulates in T and indexes scale / shift by column only.  Both are restated: for f32 inputs and one batch element they
    agree to float32 accuracy; for bf16 the T-precision statistics move the result by whole bf16 steps (why the product kernel,
    with fp32 statistics, is held to the mathematical one); for B > 1 the literal kernel applies batch 0's modulation everywhere."""
    g = np.random.default_rng(5)
    x = (g.standard_normal((1, 6, 768)) * 2 + 0.3).astype(np.float32)
    sh, sc = (g.standard_normal((1, 768)) * 0.5).astype(np.float32), (g.standard_normal((1, 768)) * 0.5).astype(np.float32)
    ideal = rc.fused_modulate(x, sh, sc, 1e-6, "f32")
    lit = rc.fused_modulate_metal_literal(x, sh, sc, "f32")
    assert np.abs(lit - ideal).max() <= 2e-5 * np.abs(ideal).max()
    xb, shb, scb = rc.bf16_round(x), rc.bf16_round(sh), rc.bf16_round(sc)
    ideal_b = rc.fused_modulate(xb, shb, scb, 1e-6, "bf16")
    lit_b = rc.fused_modulate_metal_literal(xb, shb, scb, "bf16")
    rel = np.abs(lit_b - ideal_b).max() / np.abs(ideal_b).max()
    assert 2.0 ** -8 < rel < 0.2            # visibly different (statistics summed in bf16), same function
    x2 = np.concatenate([x, x[:, ::-1]], 0)
    sh2, sc2 = np.concatenate([sh, -sh], 0), np.concatenate([sc, sc * 0.5], 0)
    lit2 = rc.fused_modulate_metal_literal(x2, sh2, sc2, "f32")
    np.testing.assert_allclose(lit2[1], rc.fused_modulate(x2[1:2], sh2[:1], sc2[:1], 1e-6, "f32")[0], rtol=2e-5, atol=2e-5)   # batch 0's shift / scale
    assert np.abs(lit2[1] - rc.fused_modulate(x2, sh2, sc2, 1e-6, "f32")[1]).max() > 0.1                                   # not its own
