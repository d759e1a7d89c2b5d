"""GPU parity for the matrix-core side of the path: nn::Linear at M > 4 (a5 prefill / DiT) and
SDPA with Tq > 1 (a1 prefill: explicit bool mask as every on-path caller passes it, "causal",
additive, none = FLUX joint attention).  Oracle: oracle/ref_core.py.

Tolerances: GEMM -- <= 1 bf16 ulp of the oracle value + fp32-accumulation noise floor.
SDPA -- P is rounded to bf16 before the second product (flash attention), so
|d| <= 2 bf16 ulp + 4e-3 * max|ref| is allowed."""
import numpy as np
import pytest

from oracle import ref_core as rc
from test_gpu_primitives import assert_bf16_close, rand

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [
    (2048, 4096, 4096),      # Qwen3-8B prefill q_proj
    (128, 1024, 4096),       # k_proj for a 128-token prompt
    (77, 200, 512),          # ragged M and N tails
    (5, 128, 64),            # smallest GEMM-path M
    (501, 512, 560),         # Paraformer first layer: K not a multiple of 64 -> generic kernel
    (64, 72, 2048),
    (300, 3072, 9216),       # Klein MLP-out shape class
])
def test_linear_gemm_parity(omx, M, N, K):
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((M, K), 31))
    w = rc.bf16_round(rand((N, K), 32) * 0.05)
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w)).numpy()
    ref = rc.linear(x, w, None, "bf16")
    assert got.shape == (M, N)
    assert_bf16_close(got, ref, 1, atol=2e-5 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K", [
    (2048, 4096, 4096),      # prefill of 2048 tokens: O projection -- 8 x 16 tiles, two K halves of 32 steps
    (2048, 4096, 12288),     # ... down projection
    (1900, 3000, 4224),      # ragged tiles (8 x 12 = 96), an odd number of K steps (66 = 33 + 33)
    (2048, 2048, 8320),      # 64 tiles, 130 K steps (65 + 65: odd halves)
])
def test_linear_gemm_256_tile_kernel_split_k(omx, monkeypatch, M, N, K):
    """64 .. 128 tiles of 256^2 would leave half of the CUs idle: every tile's K range goes to TWO blocks, the f32 partial tiles meet in
    the tile's last block (fixed summation order) which runs the epilogue.  Against the oracle, and against the unsplit kernel within
    the rounding of one extra f32 addition; deterministic run to run; bias and residual go through the same epilogue."""
    T = omx.ops.Tensor
    monkeypatch.setenv("OMX_GEMM_KSPLIT", "1")      # (by default only 100+ tiles with K >= 8192 split: where it is faster)
    x = rc.bf16_round(rand((M, K), 141))
    w = rc.bf16_round(rand((N, K), 142) * 0.05)
    b = rc.bf16_round(rand((N,), 143))
    xt, wt, bt = T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)
    got = omx.ops.linear(xt, wt, bt).numpy()
    again = omx.ops.linear(xt, wt, bt).numpy()
    np.testing.assert_array_equal(got, again)
    ref = rc.linear(x, w, b, "bf16")
    assert_bf16_close(got, ref, 1, atol=2e-5 * np.sqrt(K) + 1e-4)
    monkeypatch.setenv("OMX_GEMM_KSPLIT", "0")
    plain = omx.ops.linear(xt, wt, bt).numpy()
    assert_bf16_close(got, plain, 1, atol=1e-5 * np.sqrt(K))
    assert (got != plain).mean() < 0.2          # (and mostly the very same bf16 values)


@pytest.mark.parametrize("M,N,K", [
    (256, 256, 64),          # one tile, one K step (prologue only)
    (300, 520, 128),         # ragged M and N tails, two K steps
    (512, 768, 192),         # odd number of K steps: both LDS buffers, tail without re-staging
    (1000, 1100, 1024),      # several tiles, long K loop
])
def test_linear_gemm_256_tile_kernel(omx, monkeypatch, M, N, K):
    """The 256 x 256 x 64 eight-wave kernel (phased staging, staggered wave groups) forced on small shapes;
    by default it serves GEMMs with >= 160 such tiles (DiT, long prefill)."""
    T = omx.ops.Tensor
    monkeypatch.setenv("OMX_GEMM_TILE", "256")
    x = rc.bf16_round(rand((M, K), 41))
    w = rc.bf16_round(rand((N, K), 42) * 0.05)
    b = rc.bf16_round(rand((N,), 43))
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    assert_bf16_close(got, rc.linear(x, w, b, "bf16"), 1, atol=2e-5 * np.sqrt(K) + 1e-4)
    monkeypatch.setenv("OMX_GEMM_MFMA", "32")
    got32 = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    monkeypatch.setenv("OMX_GEMM_TILE", "128")
    same = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    np.testing.assert_array_equal(got32, same)      # 32x32x16 variant: same MFMA, same k order as the 128^2 kernel
    monkeypatch.delenv("OMX_GEMM_MFMA")
    # transpose-detecting: identity activations against an asymmetric weight
    monkeypatch.setenv("OMX_GEMM_TILE", "256")
    eye = np.eye(K, dtype=np.float32)[: min(M, K)]
    wa = rc.bf16_round((np.arange(N * K).reshape(N, K) % 251 - 125).astype(np.float32) / 64)
    np.testing.assert_array_equal(omx.ops.linear(T.from_numpy(eye), T.from_numpy(wa)).numpy(), wa.T[: min(M, K)])


@pytest.mark.parametrize("M,N,K", [
    (256, 256, 128),         # one tile, two K steps: prologue + the two closing steps, no loop trip
    (300, 520, 256),         # ragged M and N tails, one loop trip
    (1000, 1100, 1024),      # several tiles, long K loop
    (2048, 4096, 4096),      # a prompt's O projection
])
@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("tile_rows", [256, 128])
def test_linear_gemm_four_wave_kernel(omx, monkeypatch, M, N, K, f16, tile_rows):
    """Round 5: the 256^2 tile on FOUR waves of 128 x 128 (csrc/gemm.hip gemm_nt_w4_kernel, the K loop generated by tools/gen_gemm5_asm.py:
    8 x 8 accumulators of 16x16x32 = 256 AGPRs per wave, both operands by LDS-DMA, three barriers per 64 k) -- against the oracle, and
    bit-identical to the eight-wave kernel (same instruction, same k order per output element); ragged edges and the bias epilogue included."""
    T = omx.ops.Tensor
    if tile_rows == 256:
        monkeypatch.setenv("OMX_GEMM_TILE", "256")
    else:       # the 128 x 256 tile (a wave owns 64 x 128): what the O / down projections of a 2 048-token prompt run on
        monkeypatch.setenv("OMX_GEMM_ROWS128", "1")
    dt = "f16" if f16 else "bf16"
    rnd = (lambda a: a.astype(np.float16).astype(np.float32)) if f16 else rc.bf16_round
    x = rnd(rand((M, K), 41))
    w = rnd(rand((N, K), 42) * 0.05)
    b = rnd(rand((N,), 43))
    xt, wt, bt = T.from_numpy(x, dt), T.from_numpy(w, dt), T.from_numpy(b, dt)
    monkeypatch.setenv("OMX_GEMM_W4", "1")
    got = omx.ops.linear(xt, wt, bt).numpy().astype(np.float32)
    monkeypatch.setenv("OMX_GEMM_W4", "0")
    if not f16:
        assert_bf16_close(got, rc.linear(x, w, b, "bf16"), 1, atol=2e-5 * np.sqrt(K) + 1e-4)
    else:
        ref = (x.astype(np.float64) @ w.astype(np.float64).T + b).astype(np.float32)
        assert np.abs(got - ref).max() <= 2.0 ** -10 * np.abs(ref).max() + 2e-5 * np.sqrt(K)
    eight = omx.ops.linear(xt, wt, bt).numpy().astype(np.float32)
    np.testing.assert_array_equal(got, eight)       # the same instruction over the same k order per output element


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_linear_gemm_four_wave_kernel_random_shapes(omx, monkeypatch, seed):
    """Random shapes (ragged M and N, K any multiple of 128, bias / no bias, both tile heights): the four-wave kernel must equal the eight-wave
    kernel bit for bit -- edge tiles take the checked epilogue, interior tiles the parked row stores, and a launch mixes both."""
    T = omx.ops.Tensor
    g = np.random.default_rng(100 + seed)
    for case in range(6):
        M = int(g.integers(1, 1200))
        N = int(g.integers(1, 300)) * 4
        K = int(g.integers(1, 12)) * 128
        rows128 = bool(case & 1)
        x = rc.bf16_round(g.standard_normal((M, K)).astype(np.float32))
        w = rc.bf16_round((g.standard_normal((N, K)) * 0.05).astype(np.float32))
        b = rc.bf16_round(g.standard_normal(N).astype(np.float32)) if case % 3 else None
        xt, wt, bt = T.from_numpy(x), T.from_numpy(w), (T.from_numpy(b) if b is not None else None)
        if rows128:
            monkeypatch.setenv("OMX_GEMM_ROWS128", "1")
            monkeypatch.delenv("OMX_GEMM_TILE", raising=False)
        else:
            monkeypatch.setenv("OMX_GEMM_TILE", "256")
            monkeypatch.delenv("OMX_GEMM_ROWS128", raising=False)
        monkeypatch.setenv("OMX_GEMM_W4", "1")
        got = omx.ops.linear(xt, wt, bt).numpy()
        monkeypatch.setenv("OMX_GEMM_W4", "0")
        want = omx.ops.linear(xt, wt, bt).numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"M={M} N={N} K={K} rows128={rows128} bias={b is not None}")


@pytest.mark.parametrize("M,N,K", [
    (128, 256, 64),          # one tile, one K step (prologue only)
    (300, 520, 128),         # ragged M and N tails, two K steps
    (384, 768, 192),         # odd number of K steps
    (1000, 1100, 1024),      # several tiles, long K loop
    (2048, 4096, 4096),      # the O projection of a 2 048-token prompt: the shape the default picks this tile for (256 blocks)
])
def test_linear_gemm_128_row_tile_is_bit_identical_to_the_256_tile(omx, monkeypatch, M, N, K):
    """Round 4: the eight-wave kernel with a 128 x 256 tile (csrc/gemm.hip TMR = 128: 64-row X pieces, one chunk per thread and
    stage, counted waits of 6 instead of 8) for grids of 80 .. 128 tiles of 256^2 -- half the chip.  Every output element is the same
    chain of 16x16x32 MFMAs over k, so it equals the 256-row tile bit for bit (bias and residual epilogues included), and the oracle
    within the usual bound."""
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((M, K), 61))
    w = rc.bf16_round(rand((N, K), 62) * 0.05)
    b = rc.bf16_round(rand((N,), 63))
    monkeypatch.setenv("OMX_GEMM_ROWS128", "1")
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    again = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    monkeypatch.setenv("OMX_GEMM_ROWS128", "0")
    monkeypatch.setenv("OMX_GEMM_TILE", "256")
    want = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(again, want)
    if M * N <= 1 << 21:
        assert_bf16_close(got, rc.linear(x, w, b, "bf16"), 1, atol=2e-5 * np.sqrt(K) + 1e-4)
    monkeypatch.setenv("OMX_GEMM_ROWS128", "1")
    monkeypatch.delenv("OMX_GEMM_TILE")
    eye = np.eye(K, dtype=np.float32)[: min(M, K)]
    wa = rc.bf16_round((np.arange(N * K).reshape(N, K) % 251 - 125).astype(np.float32) / 64)
    np.testing.assert_array_equal(omx.ops.linear(T.from_numpy(eye), T.from_numpy(wa)).numpy(), wa.T[: min(M, K)])


@pytest.mark.parametrize("M,N,K", [(300, 520, 128), (128, 256, 64), (1000, 1100, 1024), (2048, 4096, 512)])
def test_linear_gemm_in_float16(omx, M, N, K):
    """Round 4: the eight-wave kernel with float16 operands (v_mfma_f32_16x16x32_f16 on the same LDS fragments; bias added and the
    result rounded in float16) -- what a float16 checkpoint's prompt pass multiplies with.  Exact float16 products accumulated in f32:
    against numpy's f32 matmul of the same float16 values, one float16 rounding apart; transpose-detecting identity check."""
    T = omx.ops.Tensor
    x = rand((M, K), 71).astype(np.float16)
    w = (rand((N, K), 72) * 0.05).astype(np.float16)
    b = rand((N,), 73).astype(np.float16)
    got = omx.ops.linear(T.from_numpy(x, "f16"), T.from_numpy(w, "f16"), T.from_numpy(b, "f16")).numpy().astype(np.float32)
    ref = (x.astype(np.float32) @ w.astype(np.float32).T + b.astype(np.float32)).astype(np.float16).astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=2.0 ** -10, atol=2e-5 * np.sqrt(K))
    assert np.mean(got == ref) > 0.98          # (f32 summation order: a few results land on the other side of a float16 rounding)
    eye = np.eye(K, dtype=np.float16)[: min(M, K)]
    wa = ((np.arange(N * K).reshape(N, K) % 251 - 125).astype(np.float32) / 64).astype(np.float16)
    np.testing.assert_array_equal(omx.ops.linear(T.from_numpy(eye, "f16"), T.from_numpy(wa, "f16")).numpy(), wa.T[: min(M, K)])


@pytest.mark.parametrize("M,N,K", [
    (64, 64, 64),            # one tile, one K step: nothing in flight behind it
    (501, 512, 512),         # Paraformer attention projections: 8-stage ring holds the whole contraction
    (216, 512, 2048),        # Paraformer decoder FFN: ring wraps four times, ragged rows
    (130, 1100, 448),        # 7 K steps = ring depth - 1 of the 8-stage variant, ragged N
    (512, 2560, 640),        # 320 blocks: the 4-stage variant (two blocks per CU), 10 K steps
    (77, 1284, 192),         # 4-stage variant with fewer K steps than stages; N % 64 = 4
])
def test_linear_gemm_skinny_ring_kernel(omx, monkeypatch, M, N, K):
    """The 64 x 64 tile kernel with the deep LDS-DMA ring (default for GEMMs whose 128^2 grid has <= 128 tiles), forced here;
    bias through the fused epilogue, then the transpose-detecting identity check, then agreement with the 128^2 kernel to
    one bf16 ulp (different MFMA shape, same products)."""
    T = omx.ops.Tensor
    monkeypatch.setenv("OMX_GEMM_TILE", "64")
    x = rc.bf16_round(rand((M, K), 51))
    w = rc.bf16_round(rand((N, K), 52) * 0.05)
    b = rc.bf16_round(rand((N,), 53))
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    ref = rc.linear(x, w, b, "bf16")
    assert_bf16_close(got, ref, 1, atol=2e-5 * np.sqrt(K) + 1e-4)
    eye = np.eye(K, dtype=np.float32)[: min(M, K)]
    wa = rc.bf16_round((np.arange(N * K).reshape(N, K) % 251 - 125).astype(np.float32) / 64)
    np.testing.assert_array_equal(omx.ops.linear(T.from_numpy(eye), T.from_numpy(wa)).numpy(), wa.T[: min(M, K)])
    monkeypatch.setenv("OMX_GEMM_TILE", "128")
    other = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    assert_bf16_close(got, other, 1, atol=2e-5 * np.sqrt(K) + 1e-4)


@pytest.mark.parametrize("M,N,K", [
    (128, 512, 4096),        # 16 tiles x 64 K steps -> 4 splits
    (100, 4096, 12288),      # down projection of a ~100-token prompt: 128 tiles x 192 steps -> 4 splits, ragged rows
    (40, 1000, 3072),        # ragged columns, 3 splits of 16 steps
])
def test_linear_gemm_ring_kernel_split_k(omx, monkeypatch, M, N, K):
    """Few tiles and a long contraction: the ring kernel splits K over gridDim.y and the last split to arrive sums the f32
    partial tiles in split order.  Same tolerance as every GEMM; the result does not depend on which split arrives last
    (repeated launches are EQUAL, which also checks that the arrival counters return to zero)."""
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((M, K), 61))
    w = rc.bf16_round(rand((N, K), 62) * 0.03)
    b = rc.bf16_round(rand((N,), 63))
    xt, wt, bt = T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)
    ref = rc.linear(x, w, b, "bf16")
    runs = [omx.ops.linear(xt, wt, bt).numpy() for _ in range(4)]
    assert_bf16_close(runs[0], ref, 1, atol=2e-5 * np.sqrt(K) + 1e-4)
    for r in runs[1:]:
        np.testing.assert_array_equal(r, runs[0])
    monkeypatch.setenv("OMX_GEMM_SPLITK", "0")
    one = omx.ops.linear(xt, wt, bt).numpy()
    assert_bf16_close(runs[0], one, 1, atol=2e-5 * np.sqrt(K) + 1e-4)


def test_linear_gemm_bias_is_fused_addmm(omx):
    """nn/linear.rs:88-90: addmm rounds once."""
    T = omx.ops.Tensor
    M, N, K = 65, 256, 512
    x = rc.bf16_round(rand((M, K), 33))
    w = rc.bf16_round(rand((N, K), 34) * 0.05)
    b = rc.bf16_round(rand((N,), 35))
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    assert_bf16_close(got, rc.linear(x, w, b, "bf16"), 1, atol=1e-4)


def test_linear_gemm_transpose_detecting(omx):
    """A = I against an asymmetric W catches swapped C/D row/col mappings (cdna guide rule 16)."""
    T = omx.ops.Tensor
    K = 128
    x = np.eye(K, dtype=np.float32)
    w = rc.bf16_round((np.arange(K * K).reshape(K, K) % 251 - 125).astype(np.float32) / 64)
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w)).numpy()
    np.testing.assert_array_equal(got, w.T)


def _sdpa(omx, q, k, v, scale, mask_np):
    T = omx.ops.Tensor
    if mask_np is None or isinstance(mask_np, str):
        m = mask_np
    elif mask_np.dtype == np.bool_:
        m = T.from_numpy(mask_np, "bool")
    else:
        m = T.from_numpy(mask_np, "bf16")
    return omx.ops.scaled_dot_product_attention(T.from_numpy(q), T.from_numpy(k), T.from_numpy(v), scale, m).numpy()


def _check(got, ref):
    assert_bf16_close(got, ref, 2, atol=4e-3 * np.abs(ref).max())


@pytest.mark.parametrize("B,H,Hkv,Tq,Tk,D", [
    (1, 8, 8, 1024, 1024, 128),      # 8 heads: the XCD-major unit walk, 4 units of 256 rows per head
    (1, 3, 3, 1100, 1280, 128),      # Sq != Sk, ragged last query block, head count not a multiple of 8
    (2, 8, 2, 1024, 512, 128),       # batch 2, GQA, the shortest key length the kernel takes (8 tiles)
    (1, 16, 16, 1536, 768, 128),     # more units than the walk's first round on a 256-CU part only with 16 heads x 6 blocks: two units per workgroup on small parts
])
@pytest.mark.parametrize("thr", ["8", "0"])
def test_sdpa_four_wave_persistent_kernel(omx, monkeypatch, B, H, Hkv, Tq, Tk, D, thr):
    """csrc/attn_flash4.hip (round 5, OMX_ATTN_W4=1): 4 waves x 64 query rows, 32x32x16 MFMAs, the per-unit body as generated assembly
    (tools/gen_flash4_asm.py), K / V as one continuous LDS-DMA stream over the units a workgroup walks, deferred rescale (threshold 8, and
    0 = the textbook online softmax on the same code).  On the oracle within the SDPA bound and within a few bf16 ulps of the 8-wave kernel."""
    q = rc.bf16_round(rand((B, H, Tq, D), 71)); k = rc.bf16_round(rand((B, Hkv, Tk, D), 72)); v = rc.bf16_round(rand((B, Hkv, Tk, D), 73))
    scale = D ** -0.5
    monkeypatch.setenv("OMX_ATTN_W4", "0")
    old = _sdpa(omx, q, k, v, scale, None)
    monkeypatch.setenv("OMX_ATTN_W4", "1")
    monkeypatch.setenv("OMX_ATTN_W4_THR", thr)
    new = _sdpa(omx, q, k, v, scale, None)
    ref = rc.scaled_dot_product_attention(q, k, v, scale, None, "bf16")
    _check(new, ref)
    assert_bf16_close(new, old, 4, atol=4e-3 * np.abs(ref).max())


def test_sdpa_four_wave_kernel_spike_forces_the_deferred_rescale(omx, monkeypatch):
    """cdna_hip_programming.md T13 / rule 26: bounded random data never takes the deferred-rescale branch after the first tiles; one key row
    spiked against one query row at a late tile does, in both query blocks of a wave and for a single row only.  Threshold 8 == threshold 0
    to bf16 rounding, both on the float64 oracle."""
    B, H, T, D = 1, 8, 1024, 128
    q = rc.bf16_round(rand((B, H, T, D), 74)); k = rc.bf16_round(rand((B, H, T, D), 75) * 0.1); v = rc.bf16_round(rand((B, H, T, D), 76))
    k[0, 1, 700] = rc.bf16_round(q[0, 1, 230] * 6)      # query 230 (wave 3, block 1 of unit 0), key tile 10
    k[0, 1, 900] = rc.bf16_round(q[0, 1, 5] * 9)        # query 5 (wave 0, block 0), key tile 14: a second rescale of the same unit
    k[0, 5, 1000] = rc.bf16_round(q[0, 5, 600] * 7)     # another head, the unit's last tile group
    scale = D ** -0.5
    ref = rc.scaled_dot_product_attention(q, k, v, scale, None, "bf16")
    monkeypatch.setenv("OMX_ATTN_W4", "1")
    outs = {}
    for thr in ("8", "0"):
        monkeypatch.setenv("OMX_ATTN_W4_THR", thr)
        outs[thr] = _sdpa(omx, q, k, v, scale, None)
        _check(outs[thr], ref)
    assert_bf16_close(outs["8"], outs["0"], 4, atol=4e-3 * np.abs(ref).max())


@pytest.mark.parametrize("B,H,Hkv,Tq,Tk,D", [
    (1, 8, 2, 128, 128, 128),     # GQA prefill
    (1, 4, 4, 200, 200, 64),      # ragged, D = 64
    (2, 2, 1, 65, 65, 128),       # batch 2, one-past-a-tile
    (1, 4, 2, 33, 161, 128),      # chunked prefill: Tq < Tk (offset 128)
])
def test_sdpa_prefill_bool_mask_as_reference_callers_pass_it(omx, B, H, Hkv, Tq, Tk, D):
    """create_attention_mask(.., Some(true)) -> bool [Tq, offset+Tq] (utils.rs:134-153)."""
    q = rc.bf16_round(rand((B, H, Tq, D), 41))
    k = rc.bf16_round(rand((B, Hkv, Tk, D), 42))
    v = rc.bf16_round(rand((B, Hkv, Tk, D), 43))
    mask = rc.create_causal_mask(Tq, Tk - Tq)
    scale = D ** -0.5
    _check(_sdpa(omx, q, k, v, scale, mask), rc.scaled_dot_product_attention(q, k, v, scale, mask, "bf16"))
    # "causal" mode (bottom-right aligned) must agree with the explicit mask
    _check(_sdpa(omx, q, k, v, scale, "causal"), rc.scaled_dot_product_attention(q, k, v, scale, "causal", "bf16"))


def test_sdpa_prefill_sliding_window_mask(omx):
    """window_size branch of create_causal_mask (utils.rs:147-150)."""
    B, H, Hkv, T, D = 1, 2, 2, 150, 64
    q = rc.bf16_round(rand((B, H, T, D), 44)); k = rc.bf16_round(rand((B, Hkv, T, D), 45)); v = rc.bf16_round(rand((B, Hkv, T, D), 46))
    mask = rc.create_causal_mask(T, 0, 32)
    _check(_sdpa(omx, q, k, v, 0.125, mask), rc.scaled_dot_product_attention(q, k, v, 0.125, mask, "bf16"))


def test_sdpa_joint_attention_no_mask_and_additive(omx):
    """FLUX.2-klein joint attention shape class: non-causal, Sq != Sk (img queries over [txt,img] keys)."""
    B, H, Sq, Sk, D = 1, 3, 192, 256, 128
    q = rc.bf16_round(rand((B, H, Sq, D), 47)); k = rc.bf16_round(rand((B, H, Sk, D), 48)); v = rc.bf16_round(rand((B, H, Sk, D), 49))
    scale = D ** -0.5
    _check(_sdpa(omx, q, k, v, scale, None), rc.scaled_dot_product_attention(q, k, v, scale, None, "bf16"))
    addm = rc.bf16_round(rand((Sq, Sk), 50) * 2)
    _check(_sdpa(omx, q, k, v, scale, addm), rc.scaled_dot_product_attention(q, k, v, scale, addm, "bf16"))


def test_sdpa_prefill_spike_forces_rescale(omx):
    B, H, T, D = 1, 2, 256, 128
    q = rc.bf16_round(rand((B, H, T, D), 51)); k = rc.bf16_round(rand((B, H, T, D), 52) * 0.1); v = rc.bf16_round(rand((B, H, T, D), 53))
    k[0, 1, 200] = rc.bf16_round(q[0, 1, 230] * 6)   # late-tile spike for query 230 of head 1
    scale = D ** -0.5
    _check(_sdpa(omx, q, k, v, scale, "causal"), rc.scaled_dot_product_attention(q, k, v, scale, "causal", "bf16"))


@pytest.mark.parametrize("B,H,Hkv,Tq,Tk,D,mode", [
    (1, 8, 2, 1100, 1100, 128, "causal"),     # ragged last block and last tile, GQA
    (1, 8, 8, 700, 700, 128, None),           # joint attention, 8 heads: the XCD-major block order
    (2, 3, 3, 300, 428, 128, "causal"),       # chunked prefill (Tq < Tk), batch 2, head count not a multiple of 8
    (1, 4, 2, 520, 520, 64, "causal"),        # head_dim 64
    (1, 2, 2, 260, 260, 128, "bool"),         # explicit bool mask (sliding window)
    (1, 2, 1, 257, 300, 64, "additive"),
    (1, 2, 2, 40, 40, 128, "causal"),         # shorter than one tile
])
def test_sdpa_two_phase_kernel(omx, monkeypatch, B, H, Hkv, Tq, Tk, D, mode):
    """csrc/attn_prefill.hip attn_prefill_pp_kernel (8 waves in two groups one phase apart) walks the same 64-key tiles in the same order
    with the same arithmetic as the single-phase kernel: outputs bit-equal to OMX_ATTN_PP=0, and on the oracle within the SDPA bound."""
    q = rc.bf16_round(rand((B, H, Tq, D), 61)); k = rc.bf16_round(rand((B, Hkv, Tk, D), 62)); v = rc.bf16_round(rand((B, Hkv, Tk, D), 63))
    if mode == "bool":
        mask = rc.create_causal_mask(Tq, Tk - Tq, 100)
    elif mode == "additive":
        mask = rc.bf16_round(rand((Tq, Tk), 64) * 2)
    else:
        mask = mode
    scale = D ** -0.5
    outs = {}
    for pp in ("0", "1"):
        monkeypatch.setenv("OMX_ATTN_PP", pp)
        outs[pp] = _sdpa(omx, q, k, v, scale, mask)
    np.testing.assert_array_equal(outs["0"], outs["1"])
    ref = rc.scaled_dot_product_attention(q, k, v, scale, mask, "bf16")
    _check(outs["1"], ref)
    _check(outs["0"], ref)


@pytest.mark.parametrize("B,H,Hkv,Tq,Tk,D,mode", [
    (1, 8, 2, 1100, 1100, 128, "causal"),     # ragged last block and last tile, GQA
    (1, 8, 8, 700, 700, 128, None),           # joint attention, 8 heads: the XCD-major block order
    (2, 3, 3, 300, 428, 128, "causal"),       # chunked prefill (Tq < Tk), batch 2, head count not a multiple of 8
    (1, 2, 2, 260, 260, 128, "bool"),         # explicit bool mask (sliding window)
    (1, 2, 1, 257, 300, 128, "additive"),
    (1, 2, 2, 40, 40, 128, "causal"),         # shorter than one tile
])
def test_sdpa_two_phase_kernel_on_32x32_mfmas(omx, monkeypatch, B, H, Hkv, Tq, Tk, D, mode):
    """csrc/attn_prefill.hip attn_prefill_pp32_kernel (round 4, OMX_ATTN_PP32=1): the two-phase block with a wave's 32 query rows as one
    32-wide MFMA column block (32x32x16 MFMAs: half the MFMA issues for the same flops).  Another summation order inside the MFMAs, so
    not bit-equal to the 16x16x32 kernels: on the oracle within the SDPA bound, and within a bf16 ulp of the largest value of them."""
    from conftest import needs_experiments
    needs_experiments(omx)
    q = rc.bf16_round(rand((B, H, Tq, D), 61)); k = rc.bf16_round(rand((B, Hkv, Tk, D), 62)); v = rc.bf16_round(rand((B, Hkv, Tk, D), 63))
    if mode == "bool":
        mask = rc.create_causal_mask(Tq, Tk - Tq, 100)
    elif mode == "additive":
        mask = rc.bf16_round(rand((Tq, Tk), 64) * 2)
    else:
        mask = mode
    scale = D ** -0.5
    monkeypatch.setenv("OMX_ATTN_PP", "1")
    base = _sdpa(omx, q, k, v, scale, mask)
    monkeypatch.setenv("OMX_ATTN_PP32", "1")
    got = _sdpa(omx, q, k, v, scale, mask)
    ref = rc.scaled_dot_product_attention(q, k, v, scale, mask, "bf16")
    _check(got, ref)
    assert_bf16_close(got, base, 1, atol=2.0 ** -7 * np.abs(base).max())
    assert not np.array_equal(got, base) or Tq <= 64      # (it IS another kernel)


@pytest.mark.parametrize("B,H,Hkv,Tq,Tk,D", [
    (1, 40, 40, 1700, 1700, 128),     # 7 x 40 = 280 units on 256 CUs, ragged last block and last tile
    (2, 16, 4, 2300, 2300, 64),       # batch 2, GQA, head_dim 64: 288 units
    (1, 48, 48, 1500, 1790, 128),     # Tq != Tk (img queries over [txt, img] keys), 288 units, 28 tiles each
])
def test_sdpa_stream_k_kernel(omx, monkeypatch, B, H, Hkv, Tq, Tk, D):
    """csrc/attn_prefill.hip attn_prefill_sk_kernel: unmasked attention with more 256-row units than CUs is cut into equal shares of
    (unit, key tile) steps; a unit that a share boundary runs through is computed as two pieces whose un-normalised (O, m, l) meet in
    whichever finishes last (opt-in: measured slower, see the launcher).  Held to: run-to-run bit-identical (the merge is two products and a sum: arrival order cannot show), within
    one bf16 ulp of the un-cut two-phase kernel, and on the oracle for three heads (first, one in the middle of the cuts, last)."""
    from conftest import needs_experiments
    needs_experiments(omx)
    q = rc.bf16_round(rand((B, H, Tq, D), 71)); k = rc.bf16_round(rand((B, Hkv, Tk, D), 72)); v = rc.bf16_round(rand((B, Hkv, Tk, D), 73))
    scale = D ** -0.5
    monkeypatch.setenv("OMX_ATTN_STREAMK", "1")
    a = _sdpa(omx, q, k, v, scale, None)
    b2 = _sdpa(omx, q, k, v, scale, None)
    np.testing.assert_array_equal(a, b2)
    monkeypatch.setenv("OMX_ATTN_STREAMK", "0")
    whole = _sdpa(omx, q, k, v, scale, None)
    assert_bf16_close(a, whole, 1, atol=2.0 ** -7 * np.abs(whole).max())      # (one ulp of the largest value: most units are cut)
    assert (a == whole).mean() > 0.5
    G = H // Hkv
    for h in (0, H // 2 + 1, H - 1):
        ref = rc.scaled_dot_product_attention(q[:, h:h + 1], k[:, h // G:h // G + 1], v[:, h // G:h // G + 1], scale, None, "bf16")
        _check(a[:, h:h + 1], ref)


@pytest.mark.parametrize("M,N,K,bias", [(501, 512, 560, True), (501, 2048, 512, True), (501, 512, 2048, True), (216, 8404, 512, True),
                                         (33, 70, 45, False), (1, 64, 64, False), (130, 66, 1026, True), (65, 70, 128, True),
                                         (40, 130, 200, True), (64, 64, 4, False)])
def test_linear_f32_on_the_f32_matrix_cores(omx, M, N, K, bias):
    """omx_linear(dtype = f32) -> gemm_f32.hip (f32-input matrix cores: exact f32 products and accumulation; split-K partials
    summed in split order): the Paraformer path's arithmetic (funasr-mlx/src/paraformer.rs is f32 throughout).  Against float64:
    f32-roundoff class, <= 1e-6 * sum|a b| per output (MI355X_MICROARCH: 0.75-3.5e-7 measured).  K % 4 == 0 goes to the pipelined
    64 x 64 kernel (K % 64 != 0: its k-tail form; N % 4 != 0: its element-wise epilogue), everything else to the staged kernels."""
    T = omx.ops.Tensor
    g = np.random.default_rng(M * 7 + N)
    x = g.standard_normal((M, K)).astype(np.float32)
    w = (g.standard_normal((N, K)) * 0.1).astype(np.float32)
    b = g.standard_normal(N).astype(np.float32) if bias else None
    got = omx.ops.linear(T.from_numpy(x, "f32"), T.from_numpy(w, "f32"), T.from_numpy(b, "f32") if bias else None).numpy()
    want = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + 1.0
    assert got.dtype == np.float32 and np.abs(got - want).max() <= 1e-6 * mag.max()
    assert (np.abs(got - want) <= 1e-6 * mag).all()


@pytest.mark.parametrize("M,N,K", [
    (5, 4096, 4096),         # a 5-token verify pass of speculative decoding: q / o projections of the 8B model
    (8, 1000, 1536),         # ragged N (not a multiple of the 16 rows of a block), K = 3 x 512: a partial chunk
    (7, 4096, 12288),        # three K chunks (the down projection)
    (6, 2048, 1032),         # K % 512 != 0: the last vector row of the chunk is masked
    (5, 152064 // 8, 1024),  # many blocks
])
def test_linear_rows_gemv(omx, monkeypatch, M, N, K):
    """csrc/gemv_rows.hip: a Linear over M <= 8 rows streams every weight row once against all rows (launch_gemm_impl routes there
    when N * K >= 2^20).  Against the oracle, with bias / relu / residual epilogues as the GEMM kernels define them, and against the
    matrix-core route (OMX_GEMV_ROWS=0) to one bf16 ulp."""
    T = omx.ops.Tensor
    x = rc.bf16_round(rand((M, K), 61))
    w = rc.bf16_round(rand((N, K), 62) * 0.05)
    b = rc.bf16_round(rand((N,), 63))
    tol = 2e-5 * np.sqrt(K) + 1e-4
    got = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    assert_bf16_close(got, rc.linear(x, w, b, "bf16"), 1, atol=tol)
    plain = omx.ops.linear(T.from_numpy(x), T.from_numpy(w)).numpy()
    assert_bf16_close(plain, rc.linear(x, w, None, "bf16"), 1, atol=tol)
    monkeypatch.setenv("OMX_GEMV_ROWS", "0")
    other = omx.ops.linear(T.from_numpy(x), T.from_numpy(w), T.from_numpy(b)).numpy()
    assert_bf16_close(got, other, 1, atol=tol)
    eye = np.zeros((M, K), np.float32)
    eye[np.arange(M), np.arange(M) * 3] = 1.0                  # transpose-detecting: row t picks column 3 t of the weight
    monkeypatch.delenv("OMX_GEMV_ROWS")
    wa = rc.bf16_round((np.arange(N * K).reshape(N, K) % 251 - 125).astype(np.float32) / 64)
    np.testing.assert_array_equal(omx.ops.linear(T.from_numpy(eye), T.from_numpy(wa)).numpy(), wa[:, np.arange(M) * 3].T)
