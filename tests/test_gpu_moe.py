"""GPU parity of the sparse-MoE block (a6 router, a7 SwitchGLU / grouped GEMM) against
oracle/ref_moe.py: Mixtral routing and Qwen3-MoE routing, the decode path (expert-selected GEMVs,
no sort) and the sorted grouped-GEMM path (B*L*k >= 64, model.rs:250-251), ragged expert loads,
an expert that receives no token, forced-uniform routing.

Tolerance: expert outputs go through three bf16 GEMMs; |d| <= 2 bf16 ulp + 2^-7 * max|ref|.
Routing (indices) must be identical whenever the oracle's k-th / (k+1)-th score gap exceeds one bf16 ulp."""
import numpy as np
import pytest

from oracle import ref_core as rc, ref_moe as rm
from test_gpu_primitives import assert_bf16_close, rand

pytestmark = pytest.mark.gpu


def _weights(E, h, I, seed):
    return (rc.bf16_round(rand((E, h), seed) * 0.5), rc.bf16_round(rand((E, I, h), seed + 1) * 0.05),
            rc.bf16_round(rand((E, I, h), seed + 2) * 0.05), rc.bf16_round(rand((E, h, I), seed + 3) * 0.05))


def _run(omx, x, gw, wg, wu, wd, k, mode, norm=True):
    from ominix_mlx_amd import moe
    T = omx.ops.Tensor
    blk = moe.SparseMoeBlock(T.from_numpy(gw), T.from_numpy(wg), T.from_numpy(wu), T.from_numpy(wd), k, mode, norm)
    out, inds, scores = blk.forward(T.from_numpy(x), return_routing=True)
    return out.numpy(), inds.numpy(), scores.numpy()


def _compare(got, ref):
    out, inds, scores = got
    rout, rinds, rscores = ref
    same = np.sort(inds, axis=1) == np.sort(rinds, axis=1)
    assert same.all(), f"routing differs on {(~same.all(axis=1)).sum()} tokens"
    np.testing.assert_array_equal(inds, rinds)                       # same (descending) order too
    assert_bf16_close(scores, rscores, 1, atol=1e-6)
    assert_bf16_close(out, rout, 2, atol=2.0 ** -7 * np.abs(rout).max())


@pytest.mark.parametrize("mode", ["mixtral", "qwen3_moe"])
@pytest.mark.parametrize("n_tokens", [1, 3, 70])        # decode, small batch, sorted grouped-GEMM path
def test_moe_block_parity(omx, mode, n_tokens):
    E, h, I, k = 8, 512, 1024, 2
    gw, wg, wu, wd = _weights(E, h, I, 60)
    x = rc.bf16_round(rand((n_tokens, h), 70 + n_tokens))
    _compare(_run(omx, x, gw, wg, wu, wd, k, mode), rm.moe_block(x, gw, wg, wu, wd, k, mode))


def test_moe_many_experts_top8_renorm_and_empty_experts(omx):
    """Qwen3-30B-A3B-like routing shape (128 experts, top-8) on a small width; with 40 tokens most
    experts receive no row at all (zero-length segments in the plan)."""
    E, h, I, k = 128, 512, 512, 8
    gw, wg, wu, wd = _weights(E, h, I, 80)
    x = rc.bf16_round(rand((40, h), 81))
    for norm in (True, False):
        _compare(_run(omx, x, gw, wg, wu, wd, k, "qwen3_moe", norm), rm.moe_block(x, gw, wg, wu, wd, k, "qwen3_moe", norm))


def test_moe_forced_uniform_and_single_expert_load(omx):
    """forced-uniform routing (token t -> experts (2t, 2t+1) mod 8) separates kernel behaviour from
    imbalance; the opposite extreme sends every token to the same two experts (one 300-row segment)."""
    E, h, I, k, N = 8, 512, 1024, 2, 150
    _, wg, wu, wd = _weights(E, h, I, 90)
    x = np.abs(rc.bf16_round(rand((N, h), 91))) + 0.25                  # positive activations: the gate row picks the expert
    gw = np.zeros((E, h), np.float32)
    for e in range(E):                                                   # uniform: expert score depends on a token-specific dim
        gw[e, e] = 1.0
    xu = x.copy()
    for t in range(N):
        xu[t, :E] = 0.0
        xu[t, (2 * t) % E] = 8.0
        xu[t, (2 * t + 1) % E] = 4.0
    xu = rc.bf16_round(xu); gw = rc.bf16_round(gw)
    got = _run(omx, xu, gw, wg, wu, wd, k, "mixtral")
    assert (got[1][:, 0] == (2 * np.arange(N)) % E).all() and (got[1][:, 1] == (2 * np.arange(N) + 1) % E).all()
    _compare(got, rm.moe_block(xu, gw, wg, wu, wd, k, "mixtral"))
    xs = x.copy(); xs[:, :E] = 0.0; xs[:, 3] = 8.0; xs[:, 5] = 4.0      # everyone -> experts 3 and 5
    xs = rc.bf16_round(xs)
    got = _run(omx, xs, gw, wg, wu, wd, k, "mixtral")
    assert (got[1] == np.array([3, 5])).all()
    _compare(got, rm.moe_block(xs, gw, wg, wu, wd, k, "mixtral"))


def test_moe_rejects_bad_config(omx):
    from ominix_mlx_amd import moe
    T = omx.ops.Tensor
    z = lambda *s: T.from_numpy(np.zeros(s, np.float32))
    with pytest.raises(omx.OmxError, match="top_k"):
        moe.SparseMoeBlock(z(4, 512), z(4, 512, 512), z(4, 512, 512), z(4, 512, 512), 9).forward(z(1, 512))


def _ep_loopback(omx, xs, gw, wg, wu, wd, k, mode):
    """Drive `world` expert-parallel ranks from this process on one GPU: the device stages of ep.py
    (route / gather / local experts / un-permute / combine) with the fabric replaced by host copies."""
    from ominix_mlx_amd import ep
    T = omx.ops.Tensor
    world, E, h = len(xs), gw.shape[0], gw.shape[1]
    ex = ep.LoopbackExchange(world)
    blocks = [ep.ExpertParallelMoe(T.from_numpy(gw), *(T.from_numpy(ep.shard_experts(w, r, world)) for w in (wg, wu, wd)),
                                   E, k, r, world, ex, mode) for r in range(world)]
    eid_rows = [None] * world
    for r, b in enumerate(blocks):                       # route + dispatch
        rows, eids = b.dispatch(T.from_numpy(xs[r]))
        ex.put_counts(r, b._send_counts)
        ex.put_rows(r, rows.numpy(), b._send_counts)
        eid_rows[r] = eids.numpy()
    got_rows = [ex.get_rows(r) for r in range(world)]
    for r, b in enumerate(blocks):
        ex.put_rows(r, eid_rows[r], b._send_counts)
    got_eids = [ex.get_rows(r) for r in range(world)]
    recv_counts = [ex.get_counts(r) for r in range(world)]
    ys = []
    for r, b in enumerate(blocks):                       # local experts
        m = sum(recv_counts[r])
        y = b.experts(T.from_numpy(got_rows[r].reshape(max(m, 1) if m == 0 else m, h) if m else np.zeros((1, h), np.float32)),
                      T.from_numpy(got_eids[r].astype(np.uint32) if m else np.zeros(1, np.uint32), "u32"), m)
        ys.append(y.numpy()[:m])
    for r in range(world):                               # the way back: roles of the counts swap
        ex.put_rows(r, ys[r], recv_counts[r])
    outs = []
    for r, b in enumerate(blocks):
        back = ex.get_rows(r)
        n_back = sum(b._send_counts)
        outs.append(b.combine(T.from_numpy(back if n_back else np.zeros((1, h), np.float32))).numpy())
    return np.concatenate(outs, axis=0)


@pytest.mark.parametrize("mode", ["mixtral", "qwen3_moe"])
@pytest.mark.parametrize("tokens", [(1, 1), (5, 3), (90, 70)])     # decode-sized, ragged, grouped-GEMM sized
def test_expert_parallel_stages_match_single_device_block(omx, mode, tokens):
    """experts sharded over 2 ranks, tokens sharded too; each rank's device stages with the
    all-to-all replaced by a loopback must reproduce the single-device block (same kernels per row up to the
    GEMV-vs-grouped-GEMM choice, which depends on the number of rows a rank receives) and the oracle."""
    E, h, I, k = 8, 512, 1024, 2
    gw, wg, wu, wd = _weights(E, h, I, 90)
    x = rc.bf16_round(rand((sum(tokens), h), 91))
    xs = [x[:tokens[0]], x[tokens[0]:]]
    got = _ep_loopback(omx, xs, gw, wg, wu, wd, k, mode)
    ref, _, _ = rm.moe_block(x, gw, wg, wu, wd, k, mode)
    assert_bf16_close(got, ref, 2, atol=2.0 ** -7 * np.abs(ref).max())
    single, _, _ = _run(omx, x, gw, wg, wu, wd, k, mode)
    assert_bf16_close(got, single, 2, atol=2.0 ** -7 * np.abs(ref).max())


def test_expert_parallel_world1_equals_fused_block(omx):
    """With one rank the staged path is the fused omx_moe_forward split at its seams: identical output."""
    from ominix_mlx_amd import ep
    E, h, I, k = 8, 512, 1024, 2
    gw, wg, wu, wd = _weights(E, h, I, 95)
    for n in (2, 80):
        x = rc.bf16_round(rand((n, h), 96 + n))
        got = _ep_loopback(omx, [x], gw, wg, wu, wd, k, "mixtral")
        single, _, _ = _run(omx, x, gw, wg, wu, wd, k, "mixtral")
        np.testing.assert_array_equal(got, single)


# ---- the reference's own Mixtral format: quantised expert stacks through gather_qmm (mixtral-mlx/src/model.rs:182-274) ----

@pytest.mark.parametrize("bits,group", [(4, 64), (8, 64), (4, 128)])
@pytest.mark.parametrize("n_tokens", [1, 5, 70])        # packed-weight GEMVs (<= 32 slots) and dequantise + grouped GEMM
def test_quantized_moe_block_parity(omx, bits, group, n_tokens):
    """Tolerance as the bf16 block plus the quantised-GEMV noise floor of tests/test_gpu_quant.py: the oracle accumulates
    q*scale+bias products in float64, the kernels in float32 in another order."""
    from ominix_mlx_amd import moe
    T = omx.ops.Tensor
    E, h, I, k = 8, 512, 1024, 2
    gw, wg, wu, wd = _weights(E, h, I, 160)
    qg, qu, qd = (rm.quantize_experts(w, group, bits) for w in (wg, wu, wd))
    x = rc.bf16_round(rand((n_tokens, h), 170 + n_tokens))
    ref = rm.moe_block_q(x, gw, qg, qu, qd, k, group, bits)
    up = lambda trip: (T.from_numpy(trip[0], "u32"), T.from_numpy(trip[1]), T.from_numpy(trip[2]))
    blk = moe.QuantizedSparseMoeBlock(T.from_numpy(gw), up(qg), up(qu), up(qd), k, group, bits)
    out, inds, scores = blk.forward(T.from_numpy(x), return_routing=True)
    np.testing.assert_array_equal(inds.numpy(), ref[1])
    assert_bf16_close(scores.numpy(), ref[2], 1, atol=1e-6)
    assert_bf16_close(out.numpy(), ref[0], 2, atol=2.0 ** -6 * np.abs(ref[0]).max())
    # the quantised block is the bf16 block on the dequantised stacks, to the same tolerance
    deq = [np.stack([rc.dequantize(t[0][e], t[1][e], t[2][e], group, bits, "bf16") for e in range(E)]) for t in (qg, qu, qd)]
    dense = rm.moe_block(x, gw, deq[0], deq[1], deq[2], k, "mixtral")
    assert_bf16_close(out.numpy(), dense[0], 2, atol=2.0 ** -6 * np.abs(dense[0]).max())


def test_quantized_moe_rejects_bad_format(omx):
    from ominix_mlx_amd import moe
    T = omx.ops.Tensor
    z = lambda *s: T.from_numpy(np.zeros(s, np.float32))
    zq = lambda *s: T.from_numpy(np.zeros(s, np.uint32), "u32")
    trip = lambda o, i: (zq(2, o, i // 8), z(2, o, i // 64), z(2, o, i // 64))
    with pytest.raises(omx.OmxError, match="bits"):
        moe.QuantizedSparseMoeBlock(z(2, 512), trip(512, 512), trip(512, 512), trip(512, 512), 1, 64, 3).forward(z(1, 512))
    with pytest.raises(omx.OmxError, match="group_size"):
        moe.QuantizedSparseMoeBlock(z(2, 512), trip(512, 512), trip(512, 512), trip(512, 512), 1, 48, 4).forward(z(1, 512))


def test_moe_decode_with_intermediate_not_a_multiple_of_512(omx):
    """Qwen3-30B-A3B: moe_intermediate_size 768 -- the expert-selected GEMV route with a partly filled last vector row
    (down projection K = 768) must agree with the grouped-GEMM route and the oracle."""
    E, h, I, k = 16, 512, 768, 4
    gw, wg, wu, wd = _weights(E, h, I, 210)
    for n_tokens in (1, 6, 40):
        x = rc.bf16_round(rand((n_tokens, h), 220 + n_tokens))
        _compare(_run(omx, x, gw, wg, wu, wd, k, "qwen3_moe"), rm.moe_block(x, gw, wg, wu, wd, k, "qwen3_moe"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,n_tokens,E,k", [("mixtral", 70, 8, 2), ("qwen3_moe", 333, 16, 4), ("mixtral", 1100, 4, 2)])
def test_moe_grouped_256_row_tiles(omx, monkeypatch, mode, n_tokens, E, k):
    """Expert-sorted rows through the 256-row-tile route (csrc/moe.hip grouped_glu_256: [gate | up] GEMM with fused_swiglu in
    the epilogue, then down; rows gathered by row_src, ragged and empty expert segments) -- forced on small shapes, and the
    last case (550 rows per expert) takes it by default.  Same oracle and tolerance as the 128-row route, and the two routes
    agree with each other to one more bf16 rounding."""
    h, I = 512, 768
    gw, wg, wu, wd = _weights(E, h, I, 300 + n_tokens)
    x = rc.bf16_round(rand((n_tokens, h), 301 + n_tokens))
    ref = rm.moe_block(x, gw, wg, wu, wd, k, mode)
    monkeypatch.setenv("OMX_MOE_TILE", "256")
    got256 = _run(omx, x, gw, wg, wu, wd, k, mode)
    _compare(got256, ref)
    monkeypatch.setenv("OMX_MOE_TILE", "128")
    got128 = _run(omx, x, gw, wg, wu, wd, k, mode)
    assert_bf16_close(got256[0], got128[0], 2, atol=2.0 ** -7 * np.abs(ref[0]).max())
    monkeypatch.delenv("OMX_MOE_TILE")
    if n_tokens * k // E >= 256:
        np.testing.assert_array_equal(_run(omx, x, gw, wg, wu, wd, k, mode)[0], got256[0])
