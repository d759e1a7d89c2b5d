"""GPU parity of the sampler's temperature branch (SURVEY.md 8a row a10, 8f rank 2 "non-greedy sampling"):
MLX's keyed generator and `categorical`, against oracle/mlx_rng.py, which tests/test_oracle_kats.py pins on the
reference's own keyed KATs (mlx-rs/src/random.rs:549-562, 690-719).

  * key / split / bits / uniform: integer and single-rounding float work -> bit-exact;
  * gumbel: two float32 logs, each the correctly rounded value in both implementations (double log, one
    rounding); the two double logs may differ in their last bit, which can move a float32 rounding in
    ~1e-7 of the cases -> >= 99.99 % bit-exact, the rest within 1 float32 ulp;
  * categorical / sampler / engine: token ids exact against the oracle drawing from the SAME logits.
"""
import numpy as np
import pytest

from oracle import mlx_rng as rng
from oracle import ref_core as rc
from oracle import ref_qwen3 as rq
from oracle import synth
from test_gpu_primitives import rand
from test_gpu_qwen3 import CONFIGS, _engine

pytestmark = pytest.mark.gpu


def _key(omx, seed):
    return omx.ops.random_key(seed)


def _host_key(t):
    k = t.numpy().ravel()
    return (np.uint32(k[0]), np.uint32(k[1]))


@pytest.mark.parametrize("seed", [0, 1, 0x123456789ABCDEF0])
def test_key_and_split_match_oracle(omx, seed):
    k = _key(omx, seed)
    np.testing.assert_array_equal(k.numpy(), np.array(rng.key(seed), np.uint32))
    for num in (2, 3, 7):
        got = omx.ops.random_split(k, num).numpy()
        want = np.array([[a, b] for a, b in rng.split(rng.key(seed), num)], np.uint32)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 100, 101, 65537, 1 << 20])
def test_bits_and_uniform_are_bit_exact(omx, n):
    k = _key(omx, 17)
    np.testing.assert_array_equal(omx.ops.random_bits(k, (n,)).numpy(), rng.bits(rng.key(17), n))
    got = omx.ops.random_uniform(k, (n,), -2.0, 3.0).numpy()
    want = rng.uniform(-2.0, 3.0, (n,), rng.key(17))
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (got >= -2.0).all() and (got < 3.0).all()


def test_reference_keyed_kats_on_device(omx):
    k = _key(omx, 0)
    assert float(omx.ops.random_uniform(k, (1,), 0.0, 10.0).numpy()[0]) == pytest.approx(4.18, abs=0.01)      # random.rs:549-553
    np.testing.assert_allclose(omx.ops.random_uniform(k, (3,), 0.0, 10.0).numpy(), [9.65, 3.14, 6.33], atol=0.01)  # :556-562
    assert float(omx.ops.random_gumbel(k, (1,)).numpy()[0]) == pytest.approx(0.13, abs=0.01)                   # :690-694
    logits = omx.ops.Tensor.from_numpy(np.zeros((5, 20), np.float32), "f32")
    np.testing.assert_array_equal(omx.ops.random_categorical(logits, k).numpy(), [1, 1, 17, 17, 17])           # :697-707
    np.testing.assert_array_equal(omx.ops.random_categorical(logits, k, num_samples=2).numpy(),
                                  [[16, 3], [14, 10], [17, 7], [6, 8], [12, 8]])                               # :710-719


def test_gumbel_matches_oracle(omx):
    n = 300001
    got = omx.ops.random_gumbel(_key(omx, 9), (n,)).numpy()
    want = rng.gumbel((n,), rng.key(9))
    same = got.view(np.uint32) == want.view(np.uint32)
    assert same.mean() >= 0.9999
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("rows,V,S", [(1, 151936, None), (4, 5000, None), (3, 777, 5), (1, 1, None), (2, 2, 3)])
def test_categorical_matches_oracle(omx, dtype, rows, V, S):
    x = rand((rows, V), 40 + V) * 3.0
    if dtype == "bf16":
        x = rc.bf16_round(x)
    t = omx.ops.Tensor.from_numpy(x, dtype)
    got = omx.ops.random_categorical(t, _key(omx, 5), num_samples=S).numpy()
    want = rng.categorical(x, rng.key(5), num_samples=S)
    np.testing.assert_array_equal(got, want)
    # the sampler's scaling folded into the same pass (sampler.rs:14)
    inv = np.float32(1.0) / np.float32(0.8)
    got = omx.ops.random_categorical(t, _key(omx, 6), num_samples=S, inv_temp=float(inv)).numpy()
    want = rng.categorical((x.astype(np.float32) * inv).astype(np.float32), rng.key(6), num_samples=S)
    np.testing.assert_array_equal(got, want)


def test_categorical_follows_the_distribution(omx):
    """20 000 draws from softmax([0, 1, 2, 3]): frequencies within 4 sigma of the probabilities."""
    logits = np.array([0.0, 1.0, 2.0, 3.0], np.float32)
    p = np.exp(logits) / np.exp(logits).sum()
    n = 20000
    got = omx.ops.random_categorical(omx.ops.Tensor.from_numpy(logits[None, :], "f32"), _key(omx, 123), num_samples=n).numpy().ravel()
    freq = np.bincount(got, minlength=4) / n
    assert (np.abs(freq - p) <= 4 * np.sqrt(p * (1 - p) / n)).all()


# ---- through the mlx-c handle ABI, the way mlx-rs-core's DefaultSampler calls it ----

def test_default_sampler_through_handle_abi(omx):
    from ominix_mlx_amd import core, mlx_c as mx
    x = rc.bf16_round(rand((1, 4096), 77) * 4.0)
    logits = mx.Array.from_numpy(x)
    s = core.DefaultSampler()
    np.testing.assert_array_equal(s.sample(logits, 0.0).numpy(), rc.sample(x, 0.0, None))
    for temp, seed in [(0.7, 0), (1.0, 3), (1.5, 99)]:
        got = s.sample(logits, temp, mx.random_key(seed)).numpy()
        np.testing.assert_array_equal(got, rc.sample(x, temp, rng.key(seed)))
    # key = None: the global RandomState (random.rs:21-68); same seed, same stream of draws (random.rs:507-518)
    core.seed(3)
    a = [int(s.sample(logits, 0.9).numpy()[0]) for _ in range(4)]
    core.seed(3)
    b = [int(s.sample(logits, 0.9).numpy()[0]) for _ in range(4)]
    state = rng.RandomState(3)
    want = [int(rc.sample(x, 0.9, state.next())[0]) for _ in range(4)]
    assert a == b == want


def test_handle_abi_random_ops_and_errors(omx):
    from ominix_mlx_amd import OmxError, mlx_c as mx
    k = mx.random_key(0)
    k1, k2 = mx.random_split(k, 2)
    want = rng.split(rng.key(0), 2)
    np.testing.assert_array_equal(k1.numpy(), np.array(want[0], np.uint32))
    np.testing.assert_array_equal(k2.numpy(), np.array(want[1], np.uint32))
    np.testing.assert_array_equal(mx.random_bits([7], k).numpy(), rng.bits(rng.key(0), 7))
    np.testing.assert_array_equal(mx.random_uniform(0.0, 10.0, [3], k).numpy(), rng.uniform(0.0, 10.0, (3,), rng.key(0)))
    g = mx.random_gumbel([2, 5], k).numpy()
    assert np.abs(g - rng.gumbel((2, 5), rng.key(0))).max() <= 1e-6
    z = mx.Array.from_numpy(np.zeros((5, 20), np.float32), mx.FLOAT32)
    np.testing.assert_array_equal(mx.random_categorical(z, -1, None, k).numpy(), [1, 1, 17, 17, 17])
    np.testing.assert_array_equal(mx.random_categorical(z, -1, 2, k).numpy(), [[16, 3], [14, 10], [17, 7], [6, 8], [12, 8]])
    # library-global sequence (key handle empty): seeding makes it repeatable
    mx.random_seed(11)
    a = mx.random_uniform(0.0, 1.0, [4]).numpy()
    mx.random_seed(11)
    b = mx.random_uniform(0.0, 1.0, [4]).numpy()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, rng.uniform(0.0, 1.0, (4,), rng.RandomState(11).next()))
    with pytest.raises(OmxError):
        mx.random_categorical(z, 0, None, k)                  # only the last axis
    with pytest.raises(OmxError):
        mx.random_bits([4], mx.Array.from_numpy(np.zeros(3, np.uint32), mx.UINT32))   # a key is 2 words


# ---- the fused engine: Generate with temp != 0 (qwen3-mlx/src/model.rs:733-741, 785, 815) ----

@pytest.mark.parametrize("name", ["gqa2_d64", "gqa4_d128"])
@pytest.mark.parametrize("temp,seed", [(0.8, 0), (1.3, 42)])
def test_engine_temperature_sampling_draws_what_the_oracle_draws_from_the_same_logits(omx, name, temp, seed):
    cfg = CONFIGS[name]
    prompt = synth.prompt_ids(32, cfg.vocab_size)
    m = _engine(omx, cfg)
    m.set_sampler(temp, seed)
    state = rng.RandomState(seed)
    toks = [m.prefill(prompt)]
    logits = [m.last_logits()]
    for _ in range(12):
        toks.append(int(m.decode(1)[0]))
        logits.append(m.last_logits())
    assert m.decode_path() == "graph"
    want = [int(rc.sample(l[None, :], temp, state.next())[0]) for l in logits]
    assert toks == want
    # not the greedy stream, and a different seed gives a different stream
    greedy = [int(np.argmax(l)) for l in logits]
    assert toks != greedy
    m2 = _engine(omx, cfg)
    m2.set_sampler(temp, seed + 1)
    other = [m2.prefill(prompt)] + [int(t) for t in m2.decode(12)]
    assert other != toks
    # the same seed replays, in one decode call
    m3 = _engine(omx, cfg)
    m3.set_sampler(temp, seed)
    again = [m3.prefill(prompt)] + [int(t) for t in m3.decode(12)]
    assert again == toks


def test_engine_sampler_end_to_end_against_the_oracle_model(omx):
    """Whole `Generate` with temp != 0 against the oracle MODEL (its own logits): equal while the draw's winner is
    separated by more than the engine-vs-oracle logit tolerance -- checked per step, stop at the first unsafe one."""
    cfg = CONFIGS["gqa4_d128"]
    temp, seed = 0.9, 7
    prompt = synth.prompt_ids(32, cfg.vocab_size)
    oracle = rq.Qwen3Oracle(cfg, rq.synth_weights(cfg))
    ref_tokens, ref_logits = oracle.generate(prompt, 8, return_logits=True, temp=temp, seed=seed)
    m = _engine(omx, cfg)
    m.set_sampler(temp, seed)
    got = [m.prefill(prompt)] + [int(t) for t in m.decode(7)]
    state = rng.RandomState(seed)
    checked = 0
    for i in range(8):
        k = state.next()
        noisy = (ref_logits[i].astype(np.float32) * np.float32(1.0 / temp)).astype(np.float32) + rng.gumbel((1, cfg.vocab_size), k)[0]
        top2 = np.partition(noisy, -2)[-2:]
        if top2[1] - top2[0] < 0.1:      # a near-tie of the noisy scores: the two bf16 models may legitimately differ
            break
        assert got[i] == int(ref_tokens[i])
        checked += 1
    assert checked >= 3


def test_engine_sampler_switches_back_to_greedy(omx):
    cfg = CONFIGS["gqa4_d128"]
    prompt = synth.prompt_ids(24, cfg.vocab_size)
    plain = _engine(omx, cfg)
    want = [plain.prefill(prompt)] + [int(t) for t in plain.decode(6)]
    m = _engine(omx, cfg)
    m.set_sampler(1.0, 1)
    m.prefill(prompt)
    m.decode(3)
    m.reset()
    m.set_sampler(0.0)
    got = [m.prefill(prompt)] + [int(t) for t in m.decode(6)]
    assert got == want


def test_quantized_engine_temperature_sampling(omx):
    from ominix_mlx_amd import engine
    cfg = CONFIGS["gqa4_d128"]
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                     vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                     tie_word_embeddings=cfg.tie_word_embeddings, rope_scaling=cfg.rope_scaling, max_context=256,
                     quantization={"bits": 4, "group_size": 64})
    m.synth_weights()
    m.set_sampler(0.7, 5)
    state = rng.RandomState(5)
    prompt = synth.prompt_ids(16, cfg.vocab_size)
    toks, logits = [m.prefill(prompt)], [m.last_logits()]
    for _ in range(6):
        toks.append(int(m.decode(1)[0]))
        logits.append(m.last_logits())
    assert toks == [int(rc.sample(l[None, :], 0.7, state.next())[0]) for l in logits]
