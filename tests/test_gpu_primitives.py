"""GPU parity tests (through the C ABI) for the hot-path primitives, SURVEY.md section 8a rows
a1, a3, a4, a5, a8, a9, a10.  The first block replays the reference's own seeded tests on the
MI355X; the rest compares against the CPU oracle on seeded inputs, including the edge cases the
reference's layouts imply (ragged dims, offsets, GQA groups, strided KV views, masks).

Tolerances (stated per SURVEY.md 8c "Consequence"):
  f32 kernels ........ |d| <= 2e-5 * max(1,|ref|)   (fp32 accumulation order only)
  bf16 kernels ....... <= 1 bf16 ulp of the oracle value for single-rounding ops
                       (<= 2 ulp where the op holds an intermediate in bf16)
"""
import numpy as np
import pytest

from oracle import mlx_rng as rng
from oracle import ref_core as rc

pytestmark = pytest.mark.gpu


def ulp_bf16(ref):
    ref = np.abs(np.asarray(ref, dtype=np.float64))
    e = np.floor(np.log2(np.maximum(ref, 2.0 ** -126)))
    return 2.0 ** (e - 7)


def assert_bf16_close(got, ref, ulps=1.0, atol=0.0):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    tol = ulps * ulp_bf16(ref) + atol
    bad = err > tol
    assert not bad.any(), f"{bad.sum()} of {bad.size} elements off by > {ulps} bf16 ulp; worst err {err.max():.3e} (ref {ref.ravel()[err.argmax()]:.5g})"


def assert_f32_close(got, ref, rtol=2e-5):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    assert (err <= rtol * np.maximum(1.0, np.abs(ref))).all(), f"worst err {err.max():.3e}"


def rand(shape, seed, lo=-1.0, hi=1.0):
    g = np.random.default_rng(seed)
    return g.uniform(lo, hi, size=shape).astype(np.float32)


# ----------------------------------------------------------------------------------------
# the reference's own seeded tests, run on the GPU path
# ----------------------------------------------------------------------------------------

def _draw(seed):
    rng.seed(seed)
    return rng.uniform(0.0, 1.0, (2, 8, 16))


def test_ref_test_rope_on_gpu(omx):
    """mlx-rs/src/fast.rs:232-250."""
    T = omx.ops.Tensor
    a = _draw(71)
    out = omx.ops.rope(T.from_numpy(a, "f32"), 8, False, 10000.0, 1.0, 0).numpy()
    assert out.shape == (2, 8, 16) and out.dtype == np.float32
    assert out.mean(dtype=np.float64) == pytest.approx(0.45625377, abs=1e-6)
    assert out.sum(dtype=np.float64) == pytest.approx(116.800964, abs=3e-4)


def test_ref_test_rms_norm_on_gpu(omx):
    """mlx-rs/src/fast.rs:254-273."""
    T = omx.ops.Tensor
    out = omx.ops.rms_norm(T.from_numpy(_draw(103), "f32"), T.from_numpy(np.ones(16), "f32"), 1e-5).numpy()
    assert out.mean(dtype=np.float64) == pytest.approx(0.87293875, abs=1e-6)
    assert out.sum(dtype=np.float64) == pytest.approx(223.47232, abs=3e-4)


def test_ref_test_layer_norm_affine_on_gpu(omx):
    """mlx-rs/src/fast.rs:277-298."""
    T = omx.ops.Tensor
    out = omx.ops.layer_norm(T.from_numpy(_draw(635), "f32"), T.from_numpy(np.ones(16), "f32"),
                             T.from_numpy(np.zeros(16), "f32"), 1e-5).numpy()[..., 0]
    assert out.mean(dtype=np.float64) == pytest.approx(0.29099038, abs=2e-6)
    assert out.sum(dtype=np.float64) == pytest.approx(4.655846, abs=3e-5)


# ----------------------------------------------------------------------------------------
# a4 norms
# ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("rows,dim", [(1, 4096), (7, 128), (33, 1000), (5, 3), (4, 3072), (0, 64)])
@pytest.mark.parametrize("dt", ["bf16", "f32", "f16"])
def test_rms_norm_parity(omx, rows, dim, dt):
    T = omx.ops.Tensor
    x = rc.rnd(rand((rows, dim), 1) * 3, dt)
    w = rc.rnd(1 + 0.1 * rand((dim,), 2), dt)
    got = omx.ops.rms_norm(T.from_numpy(x, dt), T.from_numpy(w, dt), 1e-6).numpy()
    ref = rc.rms_norm(x, w, 1e-6, dt)
    if dt == "f32":
        assert_f32_close(got, ref)
    elif dt == "bf16":
        assert_bf16_close(got, ref, 1)
    else:
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-6)
    # weight == None is legal at the boundary (fast.h:166 "may be null")
    if rows:
        got = omx.ops.rms_norm(T.from_numpy(x, dt), None, 1e-6).numpy()
        ref = rc.rms_norm(x, None, 1e-6, dt)
        (assert_f32_close if dt == "f32" else (lambda g, r: np.testing.assert_allclose(g, r, rtol=8e-3, atol=1e-6)))(got, ref)


@pytest.mark.parametrize("rows,dim", [(3, 3072), (9, 512), (2, 37)])
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_layer_norm_parity(omx, rows, dim, dt):
    T = omx.ops.Tensor
    x = rc.rnd(rand((rows, dim), 3) * 2 + 0.5, dt)
    w = rc.rnd(1 + 0.1 * rand((dim,), 4), dt)
    b = rc.rnd(0.1 * rand((dim,), 5), dt)
    for ww, bb in ((w, b), (None, None), (w, None)):
        got = omx.ops.layer_norm(T.from_numpy(x, dt), None if ww is None else T.from_numpy(ww, dt),
                                 None if bb is None else T.from_numpy(bb, dt), 1e-5).numpy()
        ref = rc.layer_norm(x, ww, bb, 1e-5, dt)
        if dt == "f32":
            assert_f32_close(got, ref, 5e-5)
        else:
            assert_bf16_close(got, ref, 1, atol=1e-3)   # near-zero outputs: absolute floor from (x-mu) cancellation


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_fused_modulate_parity(omx, dt):
    """mlx-rs-core/src/metal_kernels.rs:260-339."""
    T = omx.ops.Tensor
    B, S, H = 2, 37, 3072
    x = rc.rnd(rand((B, S, H), 6) * 2, dt)
    shift = rc.rnd(0.2 * rand((B, H), 7), dt)
    scale = rc.rnd(0.2 * rand((B, H), 8), dt)
    got = omx.ops.fused_modulate(T.from_numpy(x, dt), T.from_numpy(shift, dt), T.from_numpy(scale, dt), 1e-6).numpy()
    ref = rc.fused_modulate(x, shift, scale, 1e-6, dt)
    if dt == "f32":
        assert_f32_close(got, ref, 5e-5)
    else:
        assert_bf16_close(got, ref, 1, atol=2e-3)


# ----------------------------------------------------------------------------------------
# a3 RoPE
# ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape,dims,trad,base,scale,offset", [
    ((1, 32, 1, 128), 128, False, 1e6, 1.0, 2047),      # Qwen3-8B decode q at ctx 2047
    ((1, 8, 17, 128), 128, False, 1e6, 1.0, 5),         # prefill chunk with offset
    ((2, 4, 9, 64), 32, False, 1e4, 0.25, 100),         # partial rotary + linear scaling (utils.rs:70-85)
    ((1, 2, 6, 64), 64, True, 1e4, 1.0, 0),             # traditional (interleaved) pairing
    ((3, 5, 16), 8, False, 1e4, 1.0, 3),                # 3-D input as in the reference test
])
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_rope_parity(omx, shape, dims, trad, base, scale, offset, dt):
    T = omx.ops.Tensor
    x = rc.rnd(rand(shape, 9), dt)
    got = omx.ops.rope(T.from_numpy(x, dt), dims, trad, base, scale, offset).numpy()
    ref = rc.rope(x, dims, trad, base, scale, offset, dt)
    if dt == "f32":
        assert_f32_close(got, ref, 1e-5)
    else:
        assert_bf16_close(got, ref, 1, atol=1e-6)


@pytest.mark.parametrize("trad", [False, True])
@pytest.mark.parametrize("dt", ["bf16", "f32", "f16"])
def test_rope_with_custom_freqs(omx, trad, dt):
    """fast::rope with `freqs` instead of a base (fast.rs:15-46): the default frequencies base^(2i/dims) handed over as `freqs` give the
    base form's result; arbitrary ones match the oracle; giving both or neither is an error like in MLX core."""
    T = omx.ops.Tensor
    x = rc.rnd(rand((2, 3, 9, 64), 19), dt)
    dims, half = 48, 24
    default = (10000.0 ** (np.arange(half, dtype=np.float64) / half)).astype(np.float32)
    tol = (lambda g, r: assert_f32_close(g, r, 2e-5)) if dt == "f32" else (lambda g, r: assert_bf16_close(g, r, 1, atol=1e-6)) if dt == "bf16" \
        else (lambda g, r: np.testing.assert_allclose(g, r, rtol=2.0 ** -10, atol=1e-6))
    got = omx.ops.rope(T.from_numpy(x, dt), dims, trad, None, 0.5, 7, freqs=T.from_numpy(default, "f32")).numpy()
    tol(got, rc.rope(x, dims, trad, 10000.0, 0.5, 7, dt))
    freqs = (1.0 + 50.0 * np.random.default_rng(5).random(half)).astype(np.float32)
    got = omx.ops.rope(T.from_numpy(x, dt), dims, trad, None, 1.0, 3, freqs=T.from_numpy(freqs, "f32")).numpy()
    tol(got, rc.rope(x, dims, trad, None, 1.0, 3, dt, freqs=freqs))
    with pytest.raises(omx.OmxError):
        omx.ops.rope(T.from_numpy(x, dt), dims, trad, 1e4, 1.0, 0, freqs=T.from_numpy(freqs, "f32"))
    with pytest.raises(omx.OmxError):
        omx.ops.rope(T.from_numpy(x, dt), dims, trad, None, 1.0, 0)


def test_rope_rejects_bad_dims(omx):
    T = omx.ops.Tensor
    with pytest.raises(omx.OmxError):
        omx.ops.rope(T.from_numpy(np.zeros((1, 2, 8)), "f32"), 16, False, 1e4, 1.0, 0)


# ----------------------------------------------------------------------------------------
# a8 fused_swiglu, add, gather, argmax
# ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [14336, 9216 * 3, 1001, 1])
@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_fused_swiglu_parity(omx, n, dt):
    T = omx.ops.Tensor
    up = rc.rnd(rand((n,), 10) * 4, dt)
    gate = rc.rnd(rand((n,), 11) * 6, dt)
    got = omx.ops.fused_swiglu(T.from_numpy(up, dt), T.from_numpy(gate, dt)).numpy()
    ref = rc.fused_swiglu(up, gate, dt)
    if dt == "f32":
        assert_f32_close(got, ref, 1e-5)
    else:
        assert_bf16_close(got, ref, 1)


def test_add_and_take_rows(omx):
    T = omx.ops.Tensor
    a = rc.bf16_round(rand((5, 300), 12))
    b = rc.bf16_round(rand((5, 300), 13))
    np.testing.assert_array_equal(omx.ops.add(T.from_numpy(a), T.from_numpy(b)).numpy(), rc.add(a, b, "bf16"))
    table = rc.bf16_round(rand((50, 256), 14))
    ids = np.array([[3, 49, 0], [7, 7, 21]], dtype=np.uint32)
    got = omx.ops.take_rows(T.from_numpy(table), T.from_numpy(ids, "u32")).numpy()
    np.testing.assert_array_equal(got, table[ids])


@pytest.mark.parametrize("n", [151936, 1000, 5])
def test_argmax_first_index_on_ties(omx, n):
    """sampler.rs:9-12; tie-break = first index."""
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((3, n), 15))
    x[0, n // 2] = 9.0
    x[1, [1, n - 1]] = 7.0          # tie -> index 1
    x[2, :] = -3.0                  # all equal -> index 0
    got = omx.ops.argmax(T.from_numpy(x)).numpy()
    np.testing.assert_array_equal(got, rc.sample_greedy(x))
    assert got.tolist() == [n // 2, 1, 0]


def test_fill_uniform_matches_numpy_twin(omx):
    from oracle import synth
    for dt in ("bf16", "f32"):
        got = omx.ops.fill_uniform((3, 1000), 1234, 0.05, 1.0, dt).numpy()
        ref = synth.uniform_pm((3, 1000), 1234, 0.05, 1.0, dt)
        np.testing.assert_array_equal(got, ref)


# ----------------------------------------------------------------------------------------
# a5 Linear at decode (M small): GEMV family
# ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("N,K", [(4096, 4096), (1024, 4096), (4096, 12288), (3072, 1024), (1000, 512),
                                 (257, 2048), (96, 1536), (64, 3072), (40, 6144), (8, 8192), (16, 14336)])
def test_linear_decode_parity(omx, N, K):
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((1, K), 16))
    w = rc.bf16_round(rand((N, K), 17) * 0.05)
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w)).numpy()
    ref = rc.linear(x, w, None, "bf16")
    assert got.shape == (1, N)
    assert_bf16_close(got, ref, 1, atol=1e-5)   # near-zero dot products: fp32 accumulation-order noise


def test_linear_small_batch(omx):
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((3, 1024), 18))
    w = rc.bf16_round(rand((512, 1024), 19) * 0.05)
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w)).numpy()
    assert_bf16_close(got, rc.linear(x, w, None, "bf16"), 1)


def test_linear_shape_error(omx):
    T = omx.ops.Tensor
    with pytest.raises(omx.OmxError):
        omx.ops.linear(T.from_numpy(np.zeros((1, 100))), T.from_numpy(np.zeros((8, 512))))


# ----------------------------------------------------------------------------------------
# a1 SDPA, decode shape (Tq == 1)
# ----------------------------------------------------------------------------------------

@pytest.mark.parametrize("B,H,Hkv,Tk,D", [
    (1, 32, 8, 2304, 128),     # Qwen3-8B decode at the end of the bench window
    (1, 32, 8, 1, 128),        # first token
    (1, 16, 8, 129, 128),      # Qwen3-0.6B, ragged length
    (2, 4, 4, 77, 64),         # MHA, D=64, batch 2
    (1, 28, 4, 300, 128),      # Qwen2.5-7B grouping (7 q heads per kv head)
    (1, 14, 2, 63, 64),        # Qwen2-0.5B
])
def test_sdpa_decode_parity(omx, B, H, Hkv, Tk, D):
    T = omx.ops.Tensor
    q = rc.bf16_round(rand((B, H, 1, D), 20) * 2)
    k = rc.bf16_round(rand((B, Hkv, Tk, D), 21) * 2)
    v = rc.bf16_round(rand((B, Hkv, Tk, D), 22))
    scale = 1.0 / np.sqrt(D)
    got = omx.ops.scaled_dot_product_attention(T.from_numpy(q), T.from_numpy(k), T.from_numpy(v), scale).numpy()
    ref = rc.scaled_dot_product_attention(q, k, v, scale, None, "bf16")
    assert_bf16_close(got, ref, 1, atol=2e-3 * np.abs(ref).max())


def test_sdpa_decode_strided_kv_view_and_masks(omx):
    """K/V handed over as the [..,:offset,:] view of the step-256 buffer (cache.rs:190-193)."""
    T = omx.ops.Tensor
    B, H, Hkv, cap, off, D = 1, 8, 2, 512, 300, 128
    q = rc.bf16_round(rand((B, H, 1, D), 23))
    kbuf = rc.bf16_round(rand((B, Hkv, cap, D), 24))
    vbuf = rc.bf16_round(rand((B, Hkv, cap, D), 25))
    scale = D ** -0.5
    tk, tv = T.from_numpy(kbuf), T.from_numpy(vbuf)
    for mask_np, mask_t in [
        (None, None),
        ("causal", "causal"),
    ]:
        got = omx.ops.scaled_dot_product_attention(T.from_numpy(q), T.from_numpy(kbuf[:, :, :off]).view((B, Hkv, off, D)),
                                                   T.from_numpy(vbuf[:, :, :off]).view((B, Hkv, off, D)), scale, mask_t).numpy()
        ref = rc.scaled_dot_product_attention(q, kbuf[:, :, :off], vbuf[:, :, :off], scale, mask_np, "bf16")
        assert_bf16_close(got, ref, 1, atol=2e-3 * np.abs(ref).max())
    # strided view: same buffers, Tk = off, strides of the full-capacity buffer
    kview = T((B, Hkv, off, D), "bf16", ptr=tk.ptr, owner=tk)
    vview = T((B, Hkv, off, D), "bf16", ptr=tv.ptr, owner=tv)
    got = omx.ops.scaled_dot_product_attention(T.from_numpy(q), kview, vview, scale, None,
                                               kv_strides=(Hkv * cap * D, cap * D)).numpy()
    ref = rc.scaled_dot_product_attention(q, kbuf[:, :, :off], vbuf[:, :, :off], scale, None, "bf16")
    assert_bf16_close(got, ref, 1, atol=2e-3 * np.abs(ref).max())
    # bool mask (keep-where-true) and additive mask of shape [1, Tk]
    keep = (np.arange(off) % 3 != 0)[None, :]
    got = omx.ops.scaled_dot_product_attention(T.from_numpy(q), kview, vview, scale, T.from_numpy(keep, "bool"),
                                               kv_strides=(Hkv * cap * D, cap * D)).numpy()
    ref = rc.scaled_dot_product_attention(q, kbuf[:, :, :off], vbuf[:, :, :off], scale, keep, "bf16")
    assert_bf16_close(got, ref, 1, atol=2e-3 * np.abs(ref).max())
    addm = rc.bf16_round(rand((1, off), 26) * 3)
    got = omx.ops.scaled_dot_product_attention(T.from_numpy(q), kview, vview, scale, T.from_numpy(addm, "bf16"),
                                               kv_strides=(Hkv * cap * D, cap * D)).numpy()
    ref = rc.scaled_dot_product_attention(q, kbuf[:, :, :off], vbuf[:, :, :off], scale, addm, "bf16")
    assert_bf16_close(got, ref, 1, atol=2e-3 * np.abs(ref).max())


def test_sdpa_softmax_spike_forces_rescale(omx):
    """One key far above the rest in a late tile: exercises the running-max rescale branch of the
    online softmax (cdna_hip_programming.md rule 26)."""
    T = omx.ops.Tensor
    B, H, Hkv, Tk, D = 1, 4, 1, 700, 128
    q = rc.bf16_round(rand((B, H, 1, D), 27))
    k = rc.bf16_round(rand((B, Hkv, Tk, D), 28) * 0.1)
    v = rc.bf16_round(rand((B, Hkv, Tk, D), 29))
    k[0, 0, 555] = rc.bf16_round(q[0, 1, 0] * 8)     # huge score for head 1 at token 555
    scale = D ** -0.5
    got = omx.ops.scaled_dot_product_attention(T.from_numpy(q), T.from_numpy(k), T.from_numpy(v), scale).numpy()
    ref = rc.scaled_dot_product_attention(q, k, v, scale, None, "bf16")
    assert_bf16_close(got, ref, 1, atol=2e-3 * np.abs(ref).max())


def test_sdpa_rejects_bad_group(omx):
    T = omx.ops.Tensor
    z = np.zeros((1, 6, 1, 64))
    with pytest.raises(omx.OmxError):
        omx.ops.scaled_dot_product_attention(T.from_numpy(z), T.from_numpy(np.zeros((1, 4, 8, 64))),
                                             T.from_numpy(np.zeros((1, 4, 8, 64))), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [3584, 18944, 896, 8960, 520, 768, 8])
@pytest.mark.parametrize("M,N", [(1, 1000), (3, 37)])
def test_linear_decode_shapes_without_a_tuned_gemv(omx, M, N, K):
    """Contraction widths with no tuned GEMV instantiation (Qwen2.5-7B: 3584 / 18944, Qwen2.5-0.5B: 896, ...) go through the
    generic streaming kernel; same tolerance as every bf16 Linear: one rounding of an fp32-accumulated dot product."""
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((M, K), 900 + K))
    w = rc.bf16_round(rand((N, K), 901 + K) * 0.05)
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w)).numpy()
    want = rc.linear(x, w, None, "bf16")
    noise = 4 * 2.0 ** -9 * np.sqrt(((x.astype(np.float64) ** 2) @ (w.astype(np.float64) ** 2).T)) * 2.0 ** -8
    assert_bf16_close(got, want, 1, atol=float(noise.max()) + 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("M,n_plain,half,K", [
    (4608, 0, 3072, 256),        # FLUX double-block mlp_in shape class (no plain columns)
    (2304, 768, 2560, 128),      # single-block to_qkv_mlp shape class: q/k/v columns + [gate | up]
    (300, 256, 10244, 64),       # ragged rows, half not a multiple of the 128-column tile
    (8, 0, 128, 64),             # small problems run as one grid of the 64 x 64 ring kernel (32 gate + 32 up columns per tile)
    (77, 64, 100, 128),          # ... ragged rows, a plain segment, half not a multiple of 32
    (512, 0, 3072, 256),         # ... the DiT txt stream's mlp_in shape class
    (5, 512, 1024, 512),         # M <= 8 and >= 2^20 weights: ONE weight-streaming launch (gemv_rows.hip, segmented mode)
    (8, 0, 4096, 256),           # ... all eight rows, gate / up only
    (1, 64, 1024, 1024),         # ... a single row, a short plain segment in front
    (3, 0, 2052, 512),           # ... 4104 virtual rows: the last block's trailing waves own no rows
])
def test_linear_swiglu_matches_linear_then_fused_swiglu(omx, M, n_plain, half, K):
    """omx_linear_swiglu is an in-epilogue form of nn::Linear + fused_swiglu (klein_model.rs:489-493, 905-916):
    bit-identical to the two-launch composition, and the composition is what the oracle is compared with."""
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((M, K), 1200 + M))
    w = rc.bf16_round(rand((n_plain + 2 * half, K), 1201 + K) * 0.2)
    xt, wt = T.from_numpy(x), T.from_numpy(w)
    plain, act = omx.ops.linear_swiglu(xt, wt, n_plain)
    full = omx.ops.linear(xt, wt).numpy()
    gate = np.ascontiguousarray(full[:, n_plain:n_plain + half])
    up = np.ascontiguousarray(full[:, n_plain + half:])
    want = omx.ops.fused_swiglu(T.from_numpy(up), T.from_numpy(gate)).numpy()
    assert np.array_equal(act.numpy(), want)
    if n_plain:
        assert np.array_equal(plain.numpy(), full[:, :n_plain])
    # and against the oracle's own composition
    ref = rc.fused_swiglu(rc.bf16_round(up.astype(np.float32)), rc.bf16_round(gate.astype(np.float32)), "bf16")
    assert_bf16_close(act.numpy(), ref, 1, atol=1e-6)


@pytest.mark.gpu
def test_linear_swiglu_rejects_unsupported_widths(omx):
    T = omx.ops.Tensor
    with pytest.raises(omx.OmxError):   # contraction width must be a multiple of 64
        omx.ops.linear_swiglu(T.from_numpy(np.zeros((8, 72), np.float32)), T.from_numpy(np.zeros((256, 72), np.float32)), 0)
    with pytest.raises(omx.OmxError):   # [gate | up] must split evenly
        omx.ops.linear_swiglu(T.from_numpy(np.zeros((8, 64), np.float32)), T.from_numpy(np.zeros((255, 64), np.float32)), 0)
