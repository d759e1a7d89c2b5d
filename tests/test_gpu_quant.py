"""GPU parity of MLX affine group quantisation against oracle/ref_core.py
(`quantize`, `dequantize`, `quantized_matmul`; the oracle's quantiser is pinned by the reference's own
bound KAT, mlx-rs/src/ops/quantization.rs:289-305, in tests/test_oracle_kats.py).

  * quantize:   packed words, scales and biases EQUAL to the oracle's except where an element sits within
                float rounding of a bin edge (f32 on the device, f64 in the oracle): <= 0.1 % of the
                4/8-bit fields may differ, and then by exactly one level;
  * dequantize: bit-exact (one fma per element, one bf16 rounding);
  * quantized_matmul / gather_qmm: fp32 accumulation in a different order than the oracle's f64 dot, one
                bf16 rounding of the result: |d| <= 1 bf16 ulp + 4 * 2^-9 * sqrt(sum x_i^2 w_i^2) noise floor.
"""
import numpy as np
import pytest

from oracle import ref_core as rc
from test_gpu_primitives import assert_bf16_close, rand

pytestmark = pytest.mark.gpu


def _unpack(packed, bits):
    per = 32 // bits
    shifts = np.arange(per, dtype=np.uint32) * np.uint32(bits)
    return ((packed[..., None] >> shifts) & np.uint32((1 << bits) - 1)).reshape(*packed.shape[:-1], -1)


@pytest.mark.parametrize("bits,group", [(4, 64), (4, 32), (4, 128), (8, 64)])
def test_quantize_matches_oracle(omx, bits, group):
    T = omx.ops.Tensor
    w = rc.bf16_round(rand((96, 512), 300 + bits + group) * 0.1)
    q, s, b = omx.ops.quantize(T.from_numpy(w), group, bits)
    rq, rs, rb = rc.quantize(w, group, bits)
    assert q.shape == rq.shape and s.shape == rs.shape
    np.testing.assert_array_equal(s.numpy(), rc.bf16_round(rs))
    np.testing.assert_array_equal(b.numpy(), rc.bf16_round(rb))
    got, want = _unpack(q.numpy(), bits).astype(np.int64), _unpack(rq, bits).astype(np.int64)
    diff = got != want
    assert diff.mean() <= 1e-3 and (np.abs(got - want)[diff] == 1).all()
    # the reference's own bound (quantization.rs:289-305): max |x - x_hat| <= range / 2^bits per group
    w_hat = omx.ops.dequantize(q, s, b, group, bits).numpy()
    g = w.reshape(96, -1, group)
    rng_ = np.repeat(g.max(-1) - g.min(-1), group, axis=-1)
    assert (np.abs(w_hat - w) <= rng_ / (1 << bits) + np.abs(w) * 2.0 ** -7 + 1e-6).all()


@pytest.mark.parametrize("bits", [2, 4, 8])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_reference_quantize_dequantize_kat_on_device(omx, bits, dtype):
    """mlx-rs/src/ops/quantization.rs:289-305 run through the device kernels, all three widths of its loop: ones[128,1] * arange(512),
    group 128, shapes [128, 512/el_per_int], [128, 4], [128, 4]; in float32 (the KAT's dtype) max |x - x_hat| <= 127 / 2^bits exactly as
    the reference asserts it, and the packed words / scales / biases equal the oracle's; the 16-bit dtypes add their rounding of x."""
    T = omx.ops.Tensor
    x = np.tile(np.arange(512, dtype=np.float32), (128, 1))
    if dtype != "f32":
        x = rc.rnd(x, dtype).astype(np.float32)
    q, s, b = omx.ops.quantize(T.from_numpy(x, dtype), 128, bits)
    assert q.shape == (128, 512 * bits // 32) and s.shape == (128, 4) and b.shape == (128, 4)
    x_hat = omx.ops.dequantize(q, s, b, 128, bits).numpy()
    slack = 0.0 if dtype == "f32" else 512 * (2.0 ** -8 if dtype == "bf16" else 2.0 ** -11)
    assert np.abs(x - x_hat).max() <= 127.0 / (1 << bits) + slack
    if dtype == "f32":
        rq, rs, rb = rc.quantize(x, 128, bits)
        np.testing.assert_array_equal(q.numpy(), rq)
        np.testing.assert_array_equal(s.numpy(), rs)
        np.testing.assert_array_equal(b.numpy(), rb)
        # (the device rounds scale * q and the sum separately in float32, the oracle once from float64: an ulp apart)
        np.testing.assert_allclose(x_hat, rc.dequantize(rq, rs, rb, 128, bits, "f32"), rtol=3e-7, atol=1e-4)


@pytest.mark.parametrize("group", [32, 64, 128])
def test_two_bit_quantize_matches_oracle(omx, group):
    """2-bit affine quantisation (16 elements per word) on random weights: words equal to the oracle's up to ties at k + 0.5."""
    T = omx.ops.Tensor
    w = rc.bf16_round(rand((64, 512), 77 + group) * 0.3)
    q, s, b = omx.ops.quantize(T.from_numpy(w), group, 2)
    rq, rs, rb = rc.quantize(w, group, 2)
    np.testing.assert_array_equal(s.numpy(), rc.bf16_round(rs))
    np.testing.assert_array_equal(b.numpy(), rc.bf16_round(rb))
    got, want = _unpack(q.numpy(), 2).astype(np.int64), _unpack(rq, 2).astype(np.int64)
    diff = got != want
    assert diff.mean() <= 1e-3 and (np.abs(got - want)[diff] == 1).all()
    np.testing.assert_array_equal(omx.ops.dequantize(T.from_numpy(rq, "u32"), T.from_numpy(rc.bf16_round(rs)), T.from_numpy(rc.bf16_round(rb)), group, 2).numpy(),
                                  rc.dequantize(rq, rc.bf16_round(rs), rc.bf16_round(rb), group, 2, "bf16"))


@pytest.mark.parametrize("bits,group", [(4, 64), (8, 64), (4, 128)])
def test_dequantize_is_exact(omx, bits, group):
    T = omx.ops.Tensor
    w = rand((40, 768), 320 + bits) * 0.2
    rq, rs, rb = rc.quantize(w, group, bits)
    rs, rb = rc.bf16_round(rs), rc.bf16_round(rb)
    got = omx.ops.dequantize(T.from_numpy(rq, "u32"), T.from_numpy(rs), T.from_numpy(rb), group, bits).numpy()
    np.testing.assert_array_equal(got, rc.dequantize(rq, rs, rb, group, bits, "bf16"))


def _qmm_case(seed, M, N, K, bits, group):
    w = rand((N, K), seed) * 0.05
    rq, rs, rb = rc.quantize(w, group, bits)
    rs, rb = rc.bf16_round(rs), rc.bf16_round(rb)
    x = rc.bf16_round(rand((M, K), seed + 1))
    return x, rq, rs, rb


@pytest.mark.parametrize("M,N,K,bits,group", [
    (1, 1536, 4096, 4, 64),      # decode, 4 words per lane
    (3, 512, 1024, 4, 64),       # K = 1024: 2 words per lane
    (2, 256, 512, 4, 64),        # K = 512: 1 word per lane
    (1, 640, 2048, 8, 64),       # 8-bit
    (5, 384, 2048, 4, 32),       # group 32
    (1, 300, 4096, 4, 128),      # ragged N, group 128
    (70, 512, 1024, 4, 64),      # prefill: dequantise + MFMA GEMM
])
def test_quantized_matmul_matches_oracle(omx, M, N, K, bits, group):
    T = omx.ops.Tensor
    x, rq, rs, rb = _qmm_case(340 + M + bits, M, N, K, bits, group)
    got = omx.ops.quantized_matmul(T.from_numpy(x), T.from_numpy(rq, "u32"), T.from_numpy(rs), T.from_numpy(rb), group, bits).numpy()
    ref = rc.quantized_matmul(x, rq, rs, rb, group, bits, "bf16")
    w = rc.dequantize(rq, rs, rb, group, bits, "f32")
    # fp32 accumulation order (GEMV) / weights rounded to bf16 before the MFMA GEMM (M > 16, as MLX's qmm does):
    # 4 sigma of independent 2^-9 relative perturbations of the K products
    noise = 4 * 2.0 ** -9 * np.sqrt((x.astype(np.float64) ** 2) @ (w.astype(np.float64) ** 2).T)
    assert got.shape == ref.shape
    assert (np.abs(got - ref) <= np.abs(ref) * 2.0 ** -7 + noise + 1e-6).all()


def test_gather_qmm_selects_the_expert_per_row(omx):
    """QuantizedSwitchLinear::apply (mixtral-mlx/src/model.rs:195-201): x [n, 1, K] against rhs_indices [n, k]."""
    T = omx.ops.Tensor
    E, N, K, n, k = 4, 256, 1024, 3, 2
    ws = [rc.quantize(rand((N, K), 360 + e) * 0.05, 64, 4) for e in range(E)]
    rq = np.stack([w[0] for w in ws]); rs = rc.bf16_round(np.stack([w[1] for w in ws])); rb = rc.bf16_round(np.stack([w[2] for w in ws]))
    x = rc.bf16_round(rand((n, K), 370))
    inds = np.array([[0, 3], [2, 2], [1, 0]], np.uint32)
    got = omx.ops.gather_qmm(T.from_numpy(x), T.from_numpy(rq, "u32"), T.from_numpy(rs), T.from_numpy(rb),
                             T.from_numpy(inds.reshape(-1), "u32"), x_div=k).numpy().reshape(n, k, N)
    for t in range(n):
        for j in range(k):
            e = int(inds[t, j])
            ref = rc.quantized_matmul(x[t:t + 1], rq[e], rs[e], rb[e], 64, 4, "bf16")[0]
            assert_bf16_close(got[t, j], ref, 1, atol=2.0 ** -8 * np.abs(ref).max())


def test_mlx_c_quantized_entry_points(omx):
    """The handle-based ABI (include/omx_mlx_c.h): mlx_quantize -> vector of three arrays, mlx_dequantize,
    mlx_quantized_matmul with optional group_size / bits (defaults 64 / 4, quantized.rs:330-333), errors."""
    from ominix_mlx_amd import mlx_c as mx
    w = rc.bf16_round(rand((128, 512), 380) * 0.1)
    x = rc.bf16_round(rand((2, 512), 381))
    wq, s, b = mx.quantize(mx.Array.from_numpy(w))
    assert wq.shape == (128, 64) and s.shape == (128, 8) and wq.dtype == mx.UINT32
    rq, rs, rb = rc.quantize(w, 64, 4)
    np.testing.assert_array_equal(s.numpy(), rc.bf16_round(rs))
    w_hat = mx.dequantize(wq, s, b).numpy()
    np.testing.assert_array_equal(w_hat, rc.dequantize(wq.numpy(), s.numpy(), b.numpy(), 64, 4, "bf16"))
    y = mx.quantized_matmul(mx.Array.from_numpy(x), wq, s, b).numpy()
    ref = rc.quantized_matmul(x, wq.numpy(), s.numpy(), b.numpy(), 64, 4, "bf16")
    assert_bf16_close(y, ref, 1, atol=2.0 ** -8 * np.abs(ref).max())
    with pytest.raises(omx.OmxError, match="transpose"):
        mx.quantized_matmul(mx.Array.from_numpy(x), wq, s, b, transpose=False)
    with pytest.raises(omx.OmxError, match="divisible"):
        mx.quantize(mx.Array.from_numpy(rand((4, 100), 1)))


@pytest.mark.parametrize("M,bits,group,K", [(1, 4, 64, 1024), (5, 4, 32, 1024), (3, 8, 64, 1024), (40, 4, 64, 1024), (2, 4, 64, 14336)])
def test_quantized_matmul_in_float16(omx, M, bits, group, K):
    """A float16 MLX checkpoint (the reference's only Mixtral format is 4-bit, and MLX community 4-bit checkpoints are float16) runs
    in float16 END TO END in MLX (nn/quantized.rs:361-385): x, scales, biases and the result are float16, the accumulation float32.
    quant.hip Act16<true> on the packed GEMV rows (nibbles -> exact float16 values, v_dot2_f32_f16), dequantise + f32 matrix cores
    for M > 16.  Against the oracle in float16 on the SAME float16 values: one float16 rounding of the result + the f32 accumulation
    noise floor."""
    N = 256
    T = omx.ops.Tensor
    w = rc.rnd(rand((N, K), 700 + bits + group) * 0.1, "f16")
    rq_, rs, rb = rc.quantize(w, group, bits)
    s16, b16 = rs.astype(np.float16), rb.astype(np.float16)
    x = rc.rnd(rand((M, K), 701), "f16")
    got_t = omx.ops.quantized_matmul(T.from_numpy(x, "f16"), T.from_numpy(rq_, "u32"), T.from_numpy(s16, "f16"), T.from_numpy(b16, "f16"), group, bits)
    assert got_t.dtype == omx.ops.dtype_code("f16")
    got = got_t.numpy()
    ref = rc.quantized_matmul(x, rq_, s16.astype(np.float32), b16.astype(np.float32), group, bits, "f16")
    wd = rc.dequantize(rq_, s16.astype(np.float32), b16.astype(np.float32), group, bits, "f32")
    noise = 4 * 2.0 ** -12 * np.sqrt((x.astype(np.float64) ** 2) @ (wd.astype(np.float64) ** 2).T)
    assert (np.abs(got.astype(np.float64) - ref) <= 2.0 ** -10 * np.abs(ref) + noise + 1e-6).all()
    # a bfloat16 activation on the float16 checkpoint: MLX promotes with result_type(bf16, f16) = float32 (the round-4 cast to
    # the scales' dtype returned float16 and overflowed above 65 504).  The result is float32, also for an activation no float16 can hold.
    xb = rc.bf16_round(x)
    xb[0, :4] = rc.bf16_round(np.array([1.0e5, -2.0e5, 7.0e4, 9.0e4], np.float32))   # |x| > 65 504: not float16 values
    mixed = omx.ops.quantized_matmul(T.from_numpy(xb), T.from_numpy(rq_, "u32"), T.from_numpy(s16, "f16"), T.from_numpy(b16, "f16"), group, bits)
    assert mixed.dtype == omx.ops.dtype_code("f32")
    wd16 = rc.dequantize(rq_, s16.astype(np.float32), b16.astype(np.float32), group, bits, "f32").astype(np.float16).astype(np.float64)
    ref32 = xb.astype(np.float64) @ wd16.T
    got32 = mixed.numpy().astype(np.float64)
    assert np.isfinite(got32).all()
    noise32 = 4 * 2.0 ** -22 * np.sqrt((xb.astype(np.float64) ** 2) @ (wd16 ** 2).T)
    assert (np.abs(got32 - ref32) <= 2.0 ** -20 * np.abs(ref32) + noise32 + 1e-6).all()
    # and the dequantised matrix itself (dtype of the scales, as MLX): one fma per element from the exact float16 values, one rounding
    dq = omx.ops.dequantize(T.from_numpy(rq_, "u32"), T.from_numpy(s16, "f16"), T.from_numpy(b16, "f16"), group, bits).numpy()
    np.testing.assert_array_equal(dq, rc.dequantize(rq_, s16.astype(np.float32), b16.astype(np.float32), group, bits, "f32").astype(np.float16).astype(np.float32))


# ---- round 6: the decode step's fused packed-GEMV forms at the real widths, on BOTH kernels -- the VALU kernel (quant.hip) and the
#      matrix-core kernel (qgemv_mfma.hip: K = 4096 / 12288, group 64) -- each against the oracle, through a debug entry point that
#      launches exactly what engine.hip launches (prologue: RMSNorm; epilogues: store, residual, SwiGLU in both roundings, logits +
#      greedy argmax key, unrounded f32 row sums) ----
import ctypes

PRO_NONE, PRO_RMSNORM = 0, 1
EPI_STORE, EPI_RESIDUAL, EPI_SWIGLU, EPI_ARGMAX, EPI_F32 = 0, 1, 2, 3, 4


def _bind_debug(omx):
    lib = omx.lib
    vp = ctypes.c_void_p
    lib.omx_debug_qgemv.restype = ctypes.c_int
    lib.omx_debug_qgemv.argtypes = [vp] * 12 + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_int, vp]
    lib.omx_debug_qgemv_mfma.restype = None
    lib.omx_debug_qgemv_mfma.argtypes = [ctypes.c_int]
    lib.omx_debug_qgemv_grid.restype = ctypes.c_int
    lib.omx_debug_qgemv_grid.argtypes = [ctypes.c_int]
    return lib


def test_gemv_epilogue_codes_match_the_header():
    import os, re
    src = open(os.path.join(os.path.dirname(__file__), "..", "ominix-mlx_amd", "csrc", "gemv.hpp")).read()
    for name, val in (("PRO_NONE", 0), ("PRO_RMSNORM", 1), ("EPI_STORE", 0), ("EPI_RESIDUAL", 1), ("EPI_SWIGLU", 2), ("EPI_ARGMAX", 3), ("EPI_F32", 4)):
        m = re.search(r"\b%s\s*=\s*(\d+)" % name, src)
        assert m and int(m.group(1)) == val, name


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("N,K,pro,epi,single,stack", [
    (6144, 4096, PRO_RMSNORM, EPI_STORE, 0, 4096),      # q | k+v rows stacked below: two members
    (4096, 4096, PRO_NONE, EPI_RESIDUAL, 0, 0),         # o
    (1536, 4096, PRO_RMSNORM, EPI_SWIGLU, 0, 0),        # gate / up, nn::silu(g) * u roundings
    (1536, 4096, PRO_NONE, EPI_SWIGLU, 1, 0),           # ... fused_swiglu's single rounding
    (4096, 12288, PRO_NONE, EPI_RESIDUAL, 0, 0),        # down
    (1040, 12288, PRO_NONE, EPI_F32, 0, 0),             # a tensor-parallel K slice's f32 row sums
    (1000, 4096, PRO_NONE, EPI_STORE, 0, 0),            # ragged N: the last 16-row block is half empty
    (20000, 4096, PRO_RMSNORM, EPI_ARGMAX, 0, 0),       # logits + argmax partials, blocks looping over several row blocks
    (2048, 1024, PRO_RMSNORM, EPI_STORE, 0, 1024),      # the other block sizes (K / 1024 waves): Qwen3-0.6B's hidden size, q | k stacked
    (1024, 2048, PRO_NONE, EPI_RESIDUAL, 0, 0),         # ... its o projection; a TP 2 shard's K slice of Qwen3-8B's
    (1536, 3072, PRO_NONE, EPI_RESIDUAL, 0, 0),         # ... its down projection
    (304, 6144, PRO_NONE, EPI_F32, 0, 0),               # the TP 2 shard's down slice (f32 partial rows)
    (768, 1024, PRO_RMSNORM, EPI_SWIGLU, 0, 0),         # one-wave blocks with the gate / up pair
    (640, 8192, PRO_NONE, EPI_STORE, 0, 0),
    (40000, 1024, PRO_RMSNORM, EPI_ARGMAX, 0, 0),       # a vocabulary matrix at hidden 1024: streaming one-wave blocks
])
def test_fused_packed_gemv_forms_match_oracle(omx, mfma, N, K, pro, epi, single, stack):
    lib = _bind_debug(omx)
    T = omx.ops.Tensor
    seed = 900 + N % 97 + epi
    x = rc.bf16_round(rand((1, K), seed))
    nw = rc.bf16_round(1.0 + 0.1 * rand((K,), seed + 1))
    resid = rc.bf16_round(rand((N,), seed + 2))
    mats = []
    for j in range(2 if epi == EPI_SWIGLU else 1):
        w = rand((N, K), seed + 3 + j) * 0.05
        q_, s_, b_ = rc.quantize(w, 64, 4)
        mats.append((q_, rc.bf16_round(s_), rc.bf16_round(b_)))
    xin = rc.rms_norm(x, nw, 1e-6, "bf16") if pro == PRO_RMSNORM else x
    ys = [rc.quantized_matmul(xin, m[0], m[1], m[2], 64, 4, "f32")[0].astype(np.float64) for m in mats]
    wd = [rc.dequantize(m[0], m[1], m[2], 64, 4, "f32").astype(np.float64) for m in mats]
    noise = [4 * 2.0 ** -9 * np.sqrt((xin.astype(np.float64) ** 2) @ (w_ ** 2).T)[0] for w_ in wd]

    dev = [tuple(T.from_numpy(a, "u32" if a.dtype == np.uint32 else "bf16") for a in m) for m in mats]
    if stack:   # the same matrix handed over as two row-stacked members: [0, stack) and [stack, N)
        q_, s_, b_ = mats[0]
        dev = [tuple(T.from_numpy(np.ascontiguousarray(a[:stack]), "u32" if a.dtype == np.uint32 else "bf16") for a in (q_, s_, b_)),
               tuple(T.from_numpy(np.ascontiguousarray(a[stack:]), "u32" if a.dtype == np.uint32 else "bf16") for a in (q_, s_, b_))]
    xd, nwd, rd = T.from_numpy(x), T.from_numpy(nw), T.from_numpy(resid)
    out = T.from_numpy(np.zeros((N,), np.float32))          # bf16
    out32 = T.from_numpy(np.zeros((N,), np.float32), "f32")
    nslot = lib.omx_debug_qgemv_grid(N)
    slots = T.from_numpy(np.zeros((2 * nslot,), np.uint32), "u32")
    lib.omx_debug_qgemv_mfma(mfma)
    try:
        second = dev[1] if len(dev) > 1 else (None, None, None)
        omx.check(lib.omx_debug_qgemv(out.ptr, out32.ptr, slots.ptr, xd.ptr, nwd.ptr, rd.ptr, dev[0][0].ptr, dev[0][1].ptr, dev[0][2].ptr,
                                      second[0].ptr if second[0] else None, second[1].ptr if second[1] else None,
                                      second[2].ptr if second[2] else None, stack, N, K, 64, 4, pro, epi, 1e-6, single, None))
        omx.check(omx.lib.omx_synchronize(None))
    finally:
        lib.omx_debug_qgemv_mfma(-1)
    got16 = out.numpy().astype(np.float64)
    ulp = 2.0 ** -7
    if epi == EPI_F32:
        got = out32.numpy().astype(np.float64)
        assert (np.abs(got - ys[0]) <= noise[0] + 1e-6).all()
    elif epi == EPI_STORE:
        assert (np.abs(got16 - ys[0]) <= np.abs(ys[0]) * ulp + noise[0] + 1e-6).all()
    elif epi == EPI_RESIDUAL:
        ref = resid.astype(np.float64) + ys[0]
        assert (np.abs(got16 - ref) <= (np.abs(ref) + np.abs(ys[0])) * ulp + noise[0] + 1e-6).all()
    elif epi == EPI_SWIGLU:
        g, u = ys
        sg = 1.0 / (1.0 + np.exp(-g))
        ref = g * sg * u
        # d(silu)/dg <= 1.1: the gate's noise times |u|, the up row's times |silu(g)|, three (one) bf16 roundings on top
        tol = np.abs(ref) * 4 * ulp + 1.1 * (noise[0] + np.abs(g) * ulp) * np.abs(u) + (noise[1] + np.abs(u) * ulp) * np.abs(g * sg) + 1e-6
        assert (np.abs(got16 - ref) <= tol).all()
    else:
        assert (np.abs(got16 - ys[0]) <= np.abs(ys[0]) * ulp + noise[0] + 1e-6).all()
        keys = slots.numpy().view(np.uint64)[:nslot]
        best = int(keys.max())
        idx = (~best) & 0xFFFFFFFF
        assert 0 <= idx < N
        # the key carries the device's own bf16 logits: its winner must be the first maximum of what the launch stored
        assert idx == int(np.argmax(got16))


def test_matrix_core_packed_gemv_agrees_with_the_valu_kernel_on_tokens(omx):
    """The two kernels round differently (MFMA accumulation order), so they are compared through what the engine reads: logits within
    the bf16 bound of each other and the same greedy winner on a peaked row."""
    lib = _bind_debug(omx)
    T = omx.ops.Tensor
    N, K = 4096, 4096
    w = rand((N, K), 77) * 0.05
    x = rc.bf16_round(rand((1, K), 78))
    w[1234] = x[0] * 0.2                                   # one row aligned with x: a clear winner
    q_, s_, b_ = rc.quantize(w, 64, 4)
    s_, b_ = rc.bf16_round(s_), rc.bf16_round(b_)
    nw = rc.bf16_round(np.ones((K,), np.float32))
    outs = {}
    for mode in (0, 1):
        out = T.from_numpy(np.zeros((N,), np.float32))
        slots = T.from_numpy(np.zeros((2 * lib.omx_debug_qgemv_grid(N),), np.uint32), "u32")
        lib.omx_debug_qgemv_mfma(mode)
        try:
            dq, ds, db = T.from_numpy(q_, "u32"), T.from_numpy(s_), T.from_numpy(b_)
            xd, nwd = T.from_numpy(x), T.from_numpy(nw)          # (held: a temporary would be freed before the launch reads it)
            omx.check(lib.omx_debug_qgemv(out.ptr, None, slots.ptr, xd.ptr, nwd.ptr, None, dq.ptr, ds.ptr, db.ptr,
                                          None, None, None, 0, N, K, 64, 4, PRO_RMSNORM, EPI_ARGMAX, 1e-6, 0, None))
            omx.check(omx.lib.omx_synchronize(None))
        finally:
            lib.omx_debug_qgemv_mfma(-1)
        outs[mode] = (out.numpy().astype(np.float64), (~int(slots.numpy().view(np.uint64).max())) & 0xFFFFFFFF)
    assert outs[0][1] == outs[1][1] == 1234
    assert (np.abs(outs[0][0] - outs[1][0]) <= np.maximum(np.abs(outs[0][0]), 1.0) * 2.0 ** -6).all()
