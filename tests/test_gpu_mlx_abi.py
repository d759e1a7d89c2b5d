"""GPU: the mlx-c compatible handle surface (include/omx_mlx_c.h), driven the way mlx-rs /
mlx-rs-core drive it (one C call per Rust op, Guarded-style status handling), against the oracle.
Covers SURVEY.md 8a rows a2 (KVCache / ConcatKeyValueCache), a1-a5 through handles, a10."""
import numpy as np
import pytest

from oracle import ref_core as rc, ref_qwen3 as rq
from test_gpu_primitives import assert_bf16_close, rand

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mx(omx):
    from ominix_mlx_amd import mlx_c
    return mlx_c


@pytest.fixture(scope="module")
def core(omx):
    from ominix_mlx_amd import core
    return core


def test_array_lifecycle_and_views(mx):
    a = rand((2, 3, 8), 1)
    t = mx.Array.from_numpy(a, mx.FLOAT32)
    assert t.shape == (2, 3, 8) and t.strides == (24, 8, 1) and t.dtype == mx.FLOAT32
    np.testing.assert_array_equal(t.numpy(), a)
    tt = mx.transpose_axes(t, [0, 2, 1])
    assert tt.shape == (2, 8, 3) and tt.strides == (24, 1, 8)         # a view, like MLX
    np.testing.assert_array_equal(tt.numpy(), a.transpose(0, 2, 1))
    np.testing.assert_array_equal(mx.reshape(tt, [2, -1]).numpy(), a.transpose(0, 2, 1).reshape(2, -1))
    sl = mx.slice(t, [0, 1, 2], [2, 3, 8], [1, 1, 2])
    np.testing.assert_array_equal(sl.numpy(), a[:, 1:3, 2:8:2])
    np.testing.assert_array_equal(mx.expand_dims(t, 1).numpy(), a[:, None])
    s = mx.Array(mx.lib.mlx_array_new_float32(2.5))
    assert s.shape == () and s.item() == 2.5
    np.testing.assert_allclose(mx.multiply(t, s).numpy(), a * 2.5, rtol=1e-6)


def test_error_convention_status_and_message(mx, omx):
    a = mx.Array.from_numpy(np.zeros((2, 3)), mx.FLOAT32)
    b = mx.Array.from_numpy(np.zeros((4, 5)), mx.FLOAT32)
    with pytest.raises(omx.OmxError, match="broadcast"):
        mx.add(a, b)
    with pytest.raises(omx.OmxError, match="Invalid mask mode"):
        q = mx.Array.from_numpy(np.zeros((1, 1, 1, 64)))
        mx.scaled_dot_product_attention(q, q, q, 1.0, "banana")
    s = mx.lib.mlx_default_cpu_stream_new()      # no CPU backend: empty handle + message
    assert not s.ctx and b"no CPU backend" in mx.lib.omx_last_error()
    mx.lib.omx_clear_error()


def test_fast_ops_through_handles(mx):
    x = rc.bf16_round(rand((2, 5, 256), 2))
    w = rc.bf16_round(1 + 0.1 * rand((256,), 3))
    X, W = mx.Array.from_numpy(x), mx.Array.from_numpy(w)
    assert_bf16_close(mx.rms_norm(X, W, 1e-6).numpy(), rc.rms_norm(x, w, 1e-6, "bf16"), 1)
    assert_bf16_close(mx.rms_norm(X, None, 1e-6).numpy(), rc.rms_norm(x, None, 1e-6, "bf16"), 1)
    assert_bf16_close(mx.layer_norm(X, W, None, 1e-5).numpy(), rc.layer_norm(x, w, None, 1e-5, "bf16"), 1, atol=1e-3)
    q = rc.bf16_round(rand((1, 4, 7, 64), 4))
    assert_bf16_close(mx.rope(mx.Array.from_numpy(q), 64, False, 1e6, 1.0, 11).numpy(),
                      rc.rope(q, 64, False, 1e6, 1.0, 11, "bf16"), 1, atol=1e-6)
    up, gate = rc.bf16_round(rand((3, 512), 5)), rc.bf16_round(rand((3, 512), 6) * 4)
    assert_bf16_close(mx.fused_swiglu(mx.Array.from_numpy(up), mx.Array.from_numpy(gate)).numpy(),
                      rc.fused_swiglu(up, gate, "bf16"), 1)


def test_linear_as_mlx_rs_does_it(mx):
    """nn::Linear::forward = matmul(x, w.t()) / addmm(bias, x, w.t()) -- linear.rs:87-92."""
    x = rc.bf16_round(rand((1, 1, 1024), 7))
    w = rc.bf16_round(rand((512, 1024), 8) * 0.05)
    b = rc.bf16_round(rand((512,), 9))
    X, W, Bv = mx.Array.from_numpy(x), mx.Array.from_numpy(w), mx.Array.from_numpy(b)
    Wt = mx.transpose(W)
    assert Wt.strides == (1, 1024)
    assert_bf16_close(mx.matmul(X, Wt).numpy(), rc.linear(x, w, None, "bf16"), 1, atol=1e-5)
    xm = rc.bf16_round(rand((2, 9, 1024), 10))
    assert_bf16_close(mx.addmm(Bv, mx.Array.from_numpy(xm), Wt).numpy(), rc.linear(xm, w, b, "bf16"), 1, atol=1e-4)
    # a non-transposed right operand is materialised, not rejected
    wk = rc.bf16_round(rand((1024, 64), 11) * 0.05)
    assert_bf16_close(mx.matmul(mx.Array.from_numpy(xm), mx.Array.from_numpy(wk)).numpy(), rc.matmul(xm, wk, "bf16"), 1, atol=1e-4)


@pytest.mark.parametrize("cls", ["KVCache", "ConcatKeyValueCache"])
def test_kv_cache_semantics_match_cache_rs(mx, core, cls):
    """cache.rs:66-84 / 134-194 through mlx_zeros / mlx_concatenate_axis / mlx_slice_update / mlx_slice:
    same offsets, capacities, returned views and contents as the oracle for prefill + decode +
    a growth across the 256-token step + reset."""
    g_cache = getattr(core, cls)()
    o_cache = getattr(rc, cls)()
    B, Hkv, D = 1, 2, 64
    seed = 100
    for n_new in [200, 1, 1, 60, 1, 300, 1]:
        seed += 1
        k = rc.bf16_round(rand((B, Hkv, n_new, D), seed))
        v = rc.bf16_round(rand((B, Hkv, n_new, D), seed + 50))
        gk, gv = g_cache.update_and_fetch(mx.Array.from_numpy(k), mx.Array.from_numpy(v))
        ok, ov = o_cache.update_and_fetch(k, v)
        assert g_cache.offset() == o_cache.offset()
        assert gk.shape == ok.shape and gv.shape == ov.shape
        np.testing.assert_array_equal(gk.numpy(), ok)
        np.testing.assert_array_equal(gv.numpy(), ov)
        if cls == "KVCache":
            assert g_cache.keys.shape[2] == o_cache.capacity()       # same step-256 growth (incl. the trim at :165-172)
    g_cache.reset(); o_cache.reset()
    assert g_cache.offset() == o_cache.offset()


@pytest.mark.parametrize("cls", ["KVCache", "ConcatKeyValueCache"])
def test_kv_cache_trim(mx, core, cls):
    """`trim(n)` -- the operation speculative.rs:165-169 notes the reference's `KeyValueCache` trait lacks -- on the handle route:
    after trimming, offsets, returned views and contents equal a cache that never saw the trimmed positions, across the step-256
    growth boundary and for a trim larger than the cache."""
    B, Hkv, D = 1, 2, 64
    g, o, plain = getattr(core, cls)(), getattr(rc, cls)(), getattr(rc, cls)()
    chunks = [rc.bf16_round(rand((B, Hkv, n, D), 300 + i)) for i, n in enumerate([250, 5, 4, 3, 20])]
    def feed(cache, c, wrap):
        return cache.update_and_fetch(wrap(c), wrap(c * 0.5))
    for c in chunks[:3]:                       # 259 positions: past the first 256-step
        feed(g, c, mx.Array.from_numpy); feed(o, c, np.asarray)
    for c in chunks[:2]:
        feed(plain, c, np.asarray)
    assert g.trim(4) == 4 and o.trim(4) == 4   # forget the third chunk
    assert g.offset() == o.offset() == plain.offset() == 255
    for c in chunks[3:]:
        gk, gv = feed(g, c, mx.Array.from_numpy)
        ok, ov = feed(o, c, np.asarray)
        pk, pv = feed(plain, c, np.asarray)
        np.testing.assert_array_equal(gk.numpy(), pk); np.testing.assert_array_equal(gv.numpy(), pv)
        np.testing.assert_array_equal(ok, pk)
    n_before = g.offset()
    assert n_before == plain.offset() and g.trim(10_000) == n_before and g.offset() == 0      # more than is cached: everything goes


def test_masks_and_sdpa_like_the_callers(mx, core):
    """create_attention_mask(h, cache, Some(true)) then SDPA with the array mask (model.rs:401, utils.rs)."""
    B, H, Hkv, T, D, off = 1, 4, 2, 19, 64, 7
    cache = core.KVCache()
    k0 = rc.bf16_round(rand((B, Hkv, off, D), 20)); v0 = rc.bf16_round(rand((B, Hkv, off, D), 21))
    cache.update_and_fetch(mx.Array.from_numpy(k0), mx.Array.from_numpy(v0))
    h = mx.Array.from_numpy(np.zeros((B, T, 8)))
    mask = core.create_attention_mask(h, [cache], True)
    np.testing.assert_array_equal(mask.numpy(), rc.create_causal_mask(T, off))
    assert core.create_attention_mask(mx.Array.from_numpy(np.zeros((B, 1, 8))), [cache], True) is None
    assert core.create_attention_mask(h, [cache], None) == "causal"
    q = rc.bf16_round(rand((B, H, T, D), 22)); k1 = rc.bf16_round(rand((B, Hkv, T, D), 23)); v1 = rc.bf16_round(rand((B, Hkv, T, D), 24))
    K, V = cache.update_and_fetch(mx.Array.from_numpy(k1), mx.Array.from_numpy(v1))
    assert K.strides[1] == 256 * D                       # the strided view of the step-256 buffer, no copy
    got = core.scaled_dot_product_attention(mx.Array.from_numpy(q), K, V, None, D ** -0.5, mask).numpy()
    ref = rc.scaled_dot_product_attention(q, np.concatenate([k0, k1], 2), np.concatenate([v0, v1], 2), D ** -0.5,
                                          rc.create_causal_mask(T, off), "bf16")
    assert_bf16_close(got, ref, 2, atol=4e-3 * np.abs(ref).max())


def test_qwen3_block_replayed_call_by_call(mx, core):
    """TransformerBlock::forward (qwen3-mlx/src/model.rs:161-215, 263-267, 321-332) issued through the
    handle ABI exactly as the Rust does, for a 9-token prefill followed by 3 decode steps, plus the
    greedy sampler; compared with the oracle block."""
    cfg = rq.Qwen3Config(512, 1, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False)
    w = rq.synth_weights(cfg)
    oracle = rq.Qwen3Oracle(cfg, w)
    W = {k: mx.Array.from_numpy(v) for k, v in w.items()}
    p = "model.layers.0."
    rope = core.initialize_rope(cfg.head_dim, cfg.rope_theta, False, None)

    def linear(x, name):
        return mx.matmul(x, mx.transpose(W[name]))

    def block(x, mask, cache):
        B, L, _ = x.shape
        xn = mx.rms_norm(x, W[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        q, k, v = (linear(xn, p + f"self_attn.{n}_proj.weight") for n in "qkv")
        q = mx.rms_norm(mx.transpose_axes(mx.reshape(q, [B, L, cfg.num_attention_heads, -1]), [0, 2, 1, 3]),
                        W[p + "self_attn.q_norm.weight"], cfg.rms_norm_eps)
        k = mx.rms_norm(mx.transpose_axes(mx.reshape(k, [B, L, cfg.num_key_value_heads, -1]), [0, 2, 1, 3]),
                        W[p + "self_attn.k_norm.weight"], cfg.rms_norm_eps)
        v = mx.transpose_axes(mx.reshape(v, [B, L, cfg.num_key_value_heads, -1]), [0, 2, 1, 3])
        q = core.apply_rope(rope, q, cache.offset())
        k = core.apply_rope(rope, k, cache.offset())
        k, v = cache.update_and_fetch(k, v)
        m = mask if mask is not None else ("causal" if L > 1 else None)
        o = core.scaled_dot_product_attention(q, k, v, None, cfg.head_dim ** -0.5, m)
        o = mx.reshape(mx.transpose_axes(o, [0, 2, 1, 3]), [B, L, -1])
        h = mx.add(x, linear(o, p + "self_attn.o_proj.weight"))
        hn = mx.rms_norm(h, W[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        g = linear(hn, p + "mlp.gate_proj.weight")
        act = mx.multiply(mx.multiply(g, mx.sigmoid(g)), linear(hn, p + "mlp.up_proj.weight"))   # nn::silu(g) * up
        return mx.add(h, linear(act, p + "mlp.down_proj.weight"))

    g = np.random.default_rng(5)
    xs = rc.bf16_round(g.standard_normal((1, 12, cfg.hidden_size)).astype(np.float32))
    gcache, ocache = core.KVCache(), rc.KVCache()
    chunks = [(0, 9), (9, 10), (10, 11), (11, 12)]
    for a, b in chunks:
        x = xs[:, a:b]
        X = mx.Array.from_numpy(x)
        mask = core.create_attention_mask(X, [gcache], True)
        got = block(X, mask, gcache).numpy()
        omask = rc.create_attention_mask(b - a, ocache.offset(), None, True)
        ref = oracle.block(0, x, omask if isinstance(omask, np.ndarray) else None, ocache)
        assert_bf16_close(got, ref, 2, atol=2.0 ** -7 * np.abs(ref).max())
    # greedy sampler over a logits row (sampler.rs:9-12)
    logits = rc.bf16_round(rand((1, 2048), 30))
    tok = core.DefaultSampler().sample(mx.Array.from_numpy(logits), 0.0)
    assert tok.dtype == mx.UINT32 and tok.numpy().tolist() == rc.sample_greedy(logits).tolist()


def test_slice_update_donation_contract(mx):
    """mlx_slice_update used to mutate a solely-owned `src` in place unconditionally.  Contract now
    (include/omx_mlx_c.h): a sole owner's buffer moves to the result and `src` becomes unreadable (an error, never a
    silently changed value); with a second reference alive the update copies and `src` keeps its contents."""
    base = rc.bf16_round(rand((1, 2, 8, 16), 5))
    upd = rc.bf16_round(rand((1, 2, 3, 16), 6))
    want = base.copy(); want[:, :, 2:5] = upd
    # (a) a second reference keeps src intact: functional semantics
    src = mx.Array.from_numpy(base)
    keep = mx.Array.op(lambda res, h: (mx.lib.mlx_array_set(res, h)), src.h)
    out = mx.slice_update(src, mx.Array.from_numpy(upd), [0, 0, 2, 0], [1, 2, 5, 16])
    np.testing.assert_array_equal(out.numpy(), want)
    np.testing.assert_array_equal(src.numpy(), base)
    np.testing.assert_array_equal(keep.numpy(), base)
    # (b) sole owner: donated -- the result is right, reading the old handle is a loud error, freeing it is fine
    src2 = mx.Array.from_numpy(base)
    out2 = mx.slice_update(src2, mx.Array.from_numpy(upd), [0, 0, 2, 0], [1, 2, 5, 16])
    np.testing.assert_array_equal(out2.numpy(), want)
    with pytest.raises(Exception, match="donated"):
        mx.astype(src2, mx.FLOAT32)
    del src2
    np.testing.assert_array_equal(out2.numpy(), want)
