"""GPU: the mlx-c compatible handle surface (include/omx_mlx_c.h), driven the way mlx-rs /
mlx-rs-core drive it (one C call per Rust op, Guarded-style status handling), against the oracle.
Covers SURVEY.md 8a rows a2 (KVCache / ConcatKeyValueCache), a1-a5 through handles, a10."""
import numpy as np
import pytest

from oracle import ref_core as rc, ref_qwen3 as rq, synth
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
    # custom frequencies instead of a base (fast.rs:15-46 `freqs`); giving both is MLX core's error
    freqs = (1.0 + 40.0 * np.random.default_rng(8).random(32)).astype(np.float32)
    assert_bf16_close(mx.rope(mx.Array.from_numpy(q), 64, False, None, 1.0, 11, mx.Array.from_numpy(freqs, mx.FLOAT32)).numpy(),
                      rc.rope(q, 64, False, None, 1.0, 11, "bf16", freqs=freqs), 1, atol=1e-6)
    with pytest.raises(Exception):
        mx.rope(mx.Array.from_numpy(q), 64, False, 1e6, 1.0, 11, mx.Array.from_numpy(freqs, mx.FLOAT32))
    # a float16 array read back through mlx_array_data_float16 (as_slice::<f16>)
    h = rand((3, 5), 21).astype(np.float16)
    np.testing.assert_array_equal(mx.Array.from_numpy(h, mx.FLOAT16).numpy(), h)
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


def _scalar(mx, v):
    return mx.Array(mx.lib.mlx_array_new_float32(float(v)))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_batched_matmul_broadcasts_like_mlx(mx, dt):
    """mlx_matmul on N-D operands (ops.h:598-602): a[..., M, K] @ b[..., K, N] with broadcast batch dimensions, b given as a
    transposed VIEW (k_t of the explicit attention: consumed in place) or as plain row-major [K, N] (attn @ v), 1-D operands."""
    code = mx.FLOAT32 if dt == "f32" else mx.BFLOAT16
    rnd = (lambda a: a.astype(np.float32)) if dt == "f32" else rc.bf16_round
    tol = dict(rtol=2e-5, atol=2e-5) if dt == "f32" else dict(rtol=2e-2, atol=2e-2)
    a = rnd(rand((2, 3, 5, 16), 30)); b = rnd(rand((2, 3, 16, 7), 31))
    A, B = mx.Array.from_numpy(a, code), mx.Array.from_numpy(b, code)
    np.testing.assert_allclose(mx.matmul(A, B).numpy(), a @ b, **tol)
    # k_t: a transposed view of [.., N, K] -- no copy of b
    k = rnd(rand((2, 3, 9, 16), 32))
    Kt = mx.transpose_axes(mx.Array.from_numpy(k, code), [0, 1, 3, 2])
    assert Kt.strides[-2] == 1
    np.testing.assert_allclose(mx.matmul(A, Kt).numpy(), a @ k.transpose(0, 1, 3, 2), **tol)
    # broadcast: [2, 1, 5, 16] @ [3, 16, 7] -> [2, 3, 5, 7]
    a1 = rnd(rand((2, 1, 5, 16), 33)); b1 = rnd(rand((3, 16, 7), 34))
    got = mx.matmul(mx.Array.from_numpy(a1, code), mx.Array.from_numpy(b1, code))
    assert got.shape == (2, 3, 5, 7)
    np.testing.assert_allclose(got.numpy(), a1 @ b1, **tol)
    # 1-D operands drop their dimension
    v = rnd(rand((16,), 35))
    got = mx.matmul(mx.Array.from_numpy(v, code), B)
    assert got.shape == (2, 3, 7)
    np.testing.assert_allclose(got.numpy(), v @ b, **tol)
    with pytest.raises(Exception, match="inner dimensions"):
        mx.matmul(A, mx.Array.from_numpy(rnd(rand((2, 3, 15, 7), 36)), code))


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("mask", ["none", "causal", "bool", "additive"])
def test_sdpa_in_f32_and_f16(mx, dt, mask):
    """mlx_fast_scaled_dot_product_attention in the dtypes the reference tests it in (mlx-rs/src/fast.rs:303-331): float32 and float16
    operands, f32 softmax / accumulation, GQA, Tq != Tk, every mask form -- against the oracle evaluated on the same (rounded) inputs."""
    B, H, Hkv, Tq, Tk, D = 2, 4, 2, 7, 19, 64
    np_dt = np.float32 if dt == "f32" else np.float16
    code = mx.FLOAT32 if dt == "f32" else mx.FLOAT16
    q = rand((B, H, Tq, D), 40).astype(np_dt); k = rand((B, Hkv, Tk, D), 41).astype(np_dt); v = rand((B, Hkv, Tk, D), 42).astype(np_dt)
    Q, K, V = (mx.Array.from_numpy(t, code) for t in (q, k, v))
    m_np, m_arg = None, None
    if mask == "causal":
        m_np, m_arg = rc.create_causal_mask(Tq, Tk - Tq), "causal"
    elif mask == "bool":
        m_np = np.random.default_rng(5).random((Tq, Tk)) > 0.3
        m_np[:, 0] = True
        m_arg = mx.Array.from_numpy(m_np, mx.BOOL)
    elif mask == "additive":
        m_np = (0.5 * rand((Tq, Tk), 43)).astype(np_dt)
        m_arg = mx.Array.from_numpy(m_np, code)
    out = mx.scaled_dot_product_attention(Q, K, V, D ** -0.5, m_arg)
    assert out.dtype == code and out.shape == (B, H, Tq, D)
    got = (out if dt == "f32" else mx.astype(out, mx.FLOAT32)).numpy().astype(np.float64)     # (f16 -> f32 is exact)
    ref = rc.scaled_dot_product_attention(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), D ** -0.5,
                                          None if m_np is None else (m_np if m_np.dtype == np.bool_ else m_np.astype(np.float64)), "f32")
    tol = 2e-5 if dt == "f32" else 2e-3          # f16: one rounding of the output (11-bit mantissa)
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * np.abs(ref).max())


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("mask", ["none", "causal"])
def test_sdpa_float16_flash_kernel(mx, D, mask, monkeypatch):
    """float16 SDPA without a mask array runs on the flash kernel's float16 instantiation (round 4; attn_prefill.hip F16): 300 queries
    over 428 keys (ragged 64-key tiles, causal bottom-right aligned), GQA -- against the float64 oracle on the same float16 inputs, and
    against the explicit f32 form (OMX_SDPA_F16_EXPLICIT=1) which it replaces."""
    B, H, Hkv, Tq, Tk = 1, 4, 2, 300, 428
    q = rand((B, H, Tq, D), 60).astype(np.float16); k = rand((B, Hkv, Tk, D), 61).astype(np.float16); v = rand((B, Hkv, Tk, D), 62).astype(np.float16)
    Q, K, V = (mx.Array.from_numpy(t, mx.FLOAT16) for t in (q, k, v))
    m_np, m_arg = (rc.create_causal_mask(Tq, Tk - Tq), "causal") if mask == "causal" else (None, None)
    outs = {}
    for form in ("flash", "explicit"):
        if form == "explicit":
            monkeypatch.setenv("OMX_SDPA_F16_EXPLICIT", "1")
        out = mx.scaled_dot_product_attention(Q, K, V, D ** -0.5, m_arg)
        assert out.dtype == mx.FLOAT16 and out.shape == (B, H, Tq, D)
        outs[form] = mx.astype(out, mx.FLOAT32).numpy().astype(np.float64)
    ref = rc.scaled_dot_product_attention(q.astype(np.float64), k.astype(np.float64), v.astype(np.float64), D ** -0.5, m_np, "f32")
    for form in outs:
        np.testing.assert_allclose(outs[form], ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    # P is rounded to float16 ahead of the second product: the two forms differ by a fraction of the output's own rounding step
    assert np.abs(outs["flash"] - outs["explicit"]).max() <= 2.0 ** -10 * np.abs(ref).max()


def test_sanm_attention_replayed_op_by_op(mx):
    """SanmAttention::forward (funasr-mlx/src/paraformer.rs:496-532) issued through the handle ABI exactly as the Rust does -- fused
    qkv Linear, index / reshape / transpose views, q.matmul(k_t) * scale, softmax_axis, attn.matmul(v), the FSMN depthwise Conv1d over
    the v projection, out_proj -- in float32 against the float64 oracle (oracle/ref_paraformer.py:sanm_attention)."""
    from oracle import ref_paraformer as rp
    T, heads, D, ks = 37, 4, 32, 11
    dim = heads * D
    p = {"qkv_w": 0.1 * rand((3 * dim, dim), 50), "qkv_b": 0.1 * rand((3 * dim,), 51), "out_w": 0.1 * rand((dim, dim), 52),
         "out_b": 0.1 * rand((dim,), 53), "fsmn_w": 0.2 * rand((dim, ks), 54)}
    x = rand((T, dim), 55)
    ref = rp.sanm_attention(x.astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()}, heads)
    f32 = lambda a: mx.Array.from_numpy(np.asarray(a, np.float32), mx.FLOAT32)
    X = f32(x[None])
    qkv = mx.addmm(f32(p["qkv_b"]), X, mx.transpose(f32(p["qkv_w"])))                       # nn::Linear with bias (linear.rs:87-90)
    parts = [mx.slice(qkv, [0, 0, i * dim], [1, T, (i + 1) * dim]) for i in range(3)]
    q, k, v = (mx.transpose_axes(mx.reshape(t, [1, T, heads, D]), [0, 2, 1, 3]) for t in parts)
    k_t = mx.transpose_axes(k, [0, 1, 3, 2])
    scores = mx.multiply(mx.matmul(q, k_t), _scalar(mx, D ** -0.5))
    attn = mx.matmul(mx.softmax_axis(scores, -1), v)
    attn = mx.reshape(mx.transpose_axes(attn, [0, 2, 1, 3]), [1, T, dim])
    # Conv1d weight of MLX: [C_out, k, C_in / groups] (groups = dim: depthwise)
    conv = mx.conv1d(parts[2], f32(p["fsmn_w"][:, :, None]), 1, ks // 2, 1, dim)
    fsmn_out = mx.add(conv, parts[2])
    out = mx.add(mx.addmm(f32(p["out_b"]), attn, mx.transpose(f32(p["out_w"]))), fsmn_out)
    np.testing.assert_allclose(out.numpy()[0], ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_cross_attention_replayed_op_by_op(mx):
    """ParaformerDecoderLayer::cross_attention (paraformer.rs:981-1017): Tq != Tk, k / v sliced from one fused projection of the
    encoder output; float32, op by op through the handle ABI."""
    Tq, Tk, heads, D = 9, 41, 4, 32
    dim = heads * D
    wq, bq = 0.1 * rand((dim, dim), 60), 0.1 * rand((dim,), 61)
    wkv, bkv = 0.1 * rand((2 * dim, dim), 62), 0.1 * rand((2 * dim,), 63)
    wo, bo = 0.1 * rand((dim, dim), 64), 0.1 * rand((dim,), 65)
    x, enc = rand((Tq, dim), 66), rand((Tk, dim), 67)
    f64 = lambda a: np.asarray(a, np.float64)
    qr = (f64(x) @ f64(wq).T + bq).reshape(Tq, heads, D).transpose(1, 0, 2)
    kvr = f64(enc) @ f64(wkv).T + bkv
    kr = kvr[:, :dim].reshape(Tk, heads, D).transpose(1, 0, 2); vr = kvr[:, dim:].reshape(Tk, heads, D).transpose(1, 0, 2)
    s = qr @ kr.transpose(0, 2, 1) * D ** -0.5
    pr = np.exp(s - s.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
    ref = (pr @ vr).transpose(1, 0, 2).reshape(Tq, dim) @ f64(wo).T + bo
    f32 = lambda a: mx.Array.from_numpy(np.asarray(a, np.float32), mx.FLOAT32)
    q = mx.addmm(f32(bq), f32(x[None]), mx.transpose(f32(wq)))
    kv = mx.addmm(f32(bkv), f32(enc[None]), mx.transpose(f32(wkv)))
    k, v = mx.slice(kv, [0, 0, 0], [1, Tk, dim]), mx.slice(kv, [0, 0, dim], [1, Tk, 2 * dim])
    q = mx.transpose_axes(mx.reshape(q, [1, Tq, heads, D]), [0, 2, 1, 3])
    k = mx.transpose_axes(mx.reshape(k, [1, Tk, heads, D]), [0, 2, 1, 3])
    v = mx.transpose_axes(mx.reshape(v, [1, Tk, heads, D]), [0, 2, 1, 3])
    scores = mx.multiply(mx.matmul(q, mx.transpose_axes(k, [0, 1, 3, 2])), _scalar(mx, D ** -0.5))
    out = mx.matmul(mx.softmax_axis(scores, -1), v)
    out = mx.reshape(mx.transpose_axes(out, [0, 2, 1, 3]), [1, Tq, dim])
    out = mx.addmm(f32(bo), out, mx.transpose(f32(wo)))
    np.testing.assert_allclose(out.numpy()[0], ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_klein_single_block_attention_replayed_op_by_op(mx):
    """The attention of KleinSingleBlock::forward (flux-klein-mlx/src/klein_model.rs:650-660): bf16 q / k / v [B, S, H, D] transposed
    to [B, H, S, D], ops::matmul(q, k^T) / sqrt(D), softmax_axis, ops::matmul(attn, v), transpose back -- batched 4-D x 4-D bf16
    matmuls through the handle ABI, against the oracle's SDPA on the same bf16 inputs."""
    B, S, H, D = 1, 80, 4, 128
    q, k, v = (rc.bf16_round(rand((B, S, H, D), 70 + i)) for i in range(3))
    Q, K, V = (mx.transpose_axes(mx.Array.from_numpy(t), [0, 2, 1, 3]) for t in (q, k, v))
    attn = mx.matmul(Q, mx.transpose_axes(K, [0, 1, 3, 2]))
    assert attn.dtype == mx.BFLOAT16
    attn = mx.softmax_axis(mx.divide(attn, _scalar(mx, np.sqrt(D))), -1)
    assert attn.dtype == mx.FLOAT32              # bf16 scores / f32 scalar ARRAY: MLX promotes, the softmax and attn @ v run in f32
    out = mx.matmul(attn, V)
    assert out.dtype == mx.FLOAT32
    out = mx.reshape(mx.transpose_axes(out, [0, 2, 1, 3]), [B, S, H * D])
    # oracle on the same roundings: bf16 scores (one rounding of q k^T), then everything in f32
    qt, kt, vt = (t.transpose(0, 2, 1, 3).astype(np.float64) for t in (q, k, v))
    sc = rc.bf16_round(qt @ kt.transpose(0, 1, 3, 2)).astype(np.float64) / np.sqrt(D)
    pr = np.exp(sc - sc.max(-1, keepdims=True)); pr /= pr.sum(-1, keepdims=True)
    ref = (pr @ vt).transpose(0, 2, 1, 3).reshape(B, S, H * D)
    np.testing.assert_allclose(out.numpy(), ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())


def test_reductions_on_any_axis_and_general_addmm(mx):
    """The conscious omissions of round 2 closed (mlx-c ops.h: argmax_axis / softmax_axis take any axis, addmm any alpha / beta and a
    broadcastable c): checked against numpy in float32."""
    a = rand((3, 5, 7), 90).astype(np.float32)
    A = mx.Array.from_numpy(a, mx.FLOAT32)
    for ax in (0, 1, 2, -2):
        np.testing.assert_array_equal(mx.argmax_axis(A, ax).numpy(), a.argmax(axis=ax))
        assert mx.argmax_axis(A, ax, True).shape == tuple(np.expand_dims(a.argmax(axis=ax), ax % 3).shape)
        e = np.exp(a - a.max(axis=ax, keepdims=True))
        got = mx.softmax_axis(A, ax)
        assert got.shape == a.shape
        np.testing.assert_allclose(got.numpy(), e / e.sum(axis=ax, keepdims=True), rtol=2e-6, atol=2e-7)
    x = rand((4, 6, 16), 91).astype(np.float32); w = rand((16, 9), 92).astype(np.float32); c = rand((6, 9), 93).astype(np.float32)
    got = mx.addmm(mx.Array.from_numpy(c, mx.FLOAT32), mx.Array.from_numpy(x, mx.FLOAT32), mx.Array.from_numpy(w, mx.FLOAT32), 0.5, -2.0)
    np.testing.assert_allclose(got.numpy(), -2.0 * c + 0.5 * (x @ w), rtol=2e-5, atol=2e-5)
    # nn::Linear's form keeps its fused route (bias in the GEMM epilogue)
    b = rand((9,), 94).astype(np.float32)
    got = mx.addmm(mx.Array.from_numpy(b, mx.FLOAT32), mx.Array.from_numpy(x, mx.FLOAT32), mx.Array.from_numpy(w, mx.FLOAT32))
    np.testing.assert_allclose(got.numpy(), b + x @ w, rtol=2e-5, atol=2e-5)


def test_drop_in_route_replayed_natively_matches_the_engine(omx):
    """csrc/per_op_route.hip (round 5): qwen3-mlx's Model::forward + Generate::next replayed call for call through the mlx-c handle ABI by
    native code (what an unmodified crate does), on the ENGINE's own weights (omx_qwen3_get_weight + omx_mlx_array_from_device: borrowed,
    no copy).  The fused engine keeps the per-op arithmetic and rounding points, so the two routes emit the same greedy tokens until a
    near-tie of the flat i.i.d. logits flips on the summation order of a GEMV (measured: the first 106 of 231) -- asserted on the first 64;
    a 300-token prompt (the KVCache's second 256-step is allocated in the prompt, the concatenating growth during decode at 512) and 230
    new tokens; ~38 mlx_* calls per layer and token."""
    from ominix_mlx_amd import engine
    cfg = dict(hidden_size=512, num_hidden_layers=3, intermediate_size=1536, num_attention_heads=8, num_key_value_heads=4, head_dim=64,
               vocab_size=2048, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
    m = engine.Model(max_context=1024, **cfg)
    m.synth_weights()
    prompt = synth.prompt_ids(300, cfg["vocab_size"])
    want = [int(m.prefill(prompt))] + [int(t) for t in m.decode(230)]
    got = m.per_op_route(prompt, 230)
    assert [int(t) for t in got["tokens"]][:64] == want[:64]
    assert len(got["tokens"]) == 231 and all(0 <= int(t) < cfg["vocab_size"] for t in got["tokens"])
    assert 35 * 3 <= got["calls_per_token"] <= 45 * 3 + 12, got["calls_per_token"]
    assert got["ms_per_token"] > 0 and got["prefill_ms"] > 0
    # the engine is untouched by the replay (own cache, own graph): it continues where it was
    more = [int(t) for t in m.decode(4)]
    m.reset()
    again = [int(m.prefill(prompt))] + [int(t) for t in m.decode(234)]
    assert again[:231] == want and again[231:] == more
    m.close()
