"""GPU parity of the fused decode engine against the CPU restatement of qwen3-mlx
(oracle/ref_qwen3.py): greedy token ids and last-step logits on tiny Qwen3-shaped models
(SURVEY.md section 7 step 0 fixture protocol: 128-token synthetic prompt, greedy).

Tolerance: logits max-abs-diff <= 2^-7 * max|logit| * sqrt(n_layers) (one bf16 ulp of the
largest logit per layer, accumulating as a random walk: bf16 kernels vs the order-free
restatement round the same op outputs differently at ties); token ids must be EQUAL wherever
the oracle's top-1/top-2 margin exceeds twice that bound (fp32 summation order differs,
SURVEY.md section 7 "hard parts" (i))."""
import os
import numpy as np
import pytest

from oracle import ref_core as rc
from oracle import ref_qwen3 as rq
from oracle import synth

pytestmark = pytest.mark.gpu

CONFIGS = {
    # hidden, layers, inter, heads, kv_heads, head_dim, vocab, tied
    "gqa2_d64": rq.Qwen3Config(512, 2, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False),
    "gqa4_d128": rq.Qwen3Config(1024, 3, 3072, 8, 2, 128, 4096, 1e-6, 1e6, True),
    "mha_d128_linear_rope": rq.Qwen3Config(512, 2, 1024, 4, 4, 128, 1000 // 8 * 8, 1e-5, 1e4, False,
                                           {"type": "linear", "factor": 2.0}),
}


def _engine(omx, cfg, weights=None, max_context=512):
    from ominix_mlx_amd import engine
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                     intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                     tie_word_embeddings=cfg.tie_word_embeddings, rope_scaling=cfg.rope_scaling,
                     max_context=max_context, qk_norm=cfg.qk_norm, attention_bias=cfg.attention_bias)
    if weights is None:
        m.synth_weights()
    else:
        m.load_weights(weights)
    return m


@pytest.mark.parametrize("name", list(CONFIGS))
def test_greedy_decode_matches_oracle(omx, name):
    cfg = CONFIGS[name]
    weights = rq.synth_weights(cfg)
    oracle = rq.Qwen3Oracle(cfg, weights)
    n_prompt, n_new = 128, 12
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)

    m = _engine(omx, cfg)                      # device-side synthetic weights (bit-identical generator)
    first = m.prefill(prompt)
    logits0 = m.last_logits()
    rest = m.decode(n_new - 1)
    got = np.concatenate([[first], rest]).astype(np.uint32)
    assert m.offset() == n_prompt + n_new - 1   # KeyValueCache::offset after the loop

    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f} > {2*bound:.4f}"
            break                                # sequences legitimately diverge after a near-tie
    else:
        np.testing.assert_array_equal(got, ref_tokens)
    # last-step logits, when no divergence happened
    if np.array_equal(got, ref_tokens):
        assert np.abs(m.last_logits() - ref_logits[-1]).max() <= bound


def test_uploaded_weights_equal_synth_weights(omx):
    """load_weights (host arrays by HF key) and synth_weights (device generator) give the same model."""
    cfg = CONFIGS["gqa2_d64"]
    prompt = synth.prompt_ids(40, cfg.vocab_size)
    a = _engine(omx, cfg)
    b = _engine(omx, cfg, rq.synth_weights(cfg))
    ta = np.concatenate([[a.prefill(prompt)], a.decode(8)])
    tb = np.concatenate([[b.prefill(prompt)], b.decode(8)])
    np.testing.assert_array_equal(ta, tb)
    np.testing.assert_array_equal(a.last_logits(), b.last_logits())


def test_reset_and_generate_iterator(omx):
    """KVCache::reset (cache.rs:130-132) + Generate iterator protocol (model.rs:804-843)."""
    from ominix_mlx_amd import engine
    cfg = CONFIGS["gqa2_d64"]
    prompt = synth.prompt_ids(33, cfg.vocab_size)
    m = _engine(omx, cfg)
    it = engine.Generate(m, 0.0, prompt, chunk=4)
    first_run = [next(it) for _ in range(9)]
    m.reset()
    assert m.offset() == 0
    it = engine.Generate(m, 0.0, prompt, chunk=3)
    second_run = [next(it) for _ in range(9)]
    assert first_run == second_run
    with pytest.raises(omx.OmxError):
        engine.Generate(m, -0.7, prompt)       # a negative temperature is an error; temp > 0: tests/test_gpu_random.py


def test_missing_weight_and_context_overflow_are_errors(omx):
    from ominix_mlx_amd import engine
    cfg = CONFIGS["gqa2_d64"]
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                     intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, vocab_size=cfg.vocab_size,
                     max_context=256)
    with pytest.raises(omx.OmxError, match="WeightNotFound"):
        m.prefill([1, 2, 3])
    m2 = _engine(omx, cfg, max_context=256)
    with pytest.raises(omx.OmxError, match="exceed"):
        m2.prefill(synth.prompt_ids(300, cfg.vocab_size))
    with pytest.raises(omx.OmxError, match="out of range"):
        m2.prefill([cfg.vocab_size + 5])


def test_batched_prefill_equals_token_serial_prefill(omx, monkeypatch):
    """The matrix-core prefill (GEMM + flash attention) and the token-serial prefill (decode
    kernels) are two implementations of the same arithmetic: KV caches and next-token logits
    must agree to bf16 rounding, and both must agree with the oracle."""
    import ctypes
    cfg = CONFIGS["gqa4_d128"]
    prompt = synth.prompt_ids(97, cfg.vocab_size)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OMX_PREFILL_SERIAL", mode)
        m = _engine(omx, cfg)
        first = m.prefill(prompt)
        n = cfg.num_key_value_heads * 512 * cfg.head_dim
        raw = np.empty(n, np.uint16)
        omx.check(omx.lib.omx_qwen3_debug_read(m._h, b"k1", raw.ctypes.data, n))
        k1 = rc.from_bf16_bits(raw).reshape(cfg.num_key_value_heads, 512, cfg.head_dim)[:, :97]
        outs[mode] = (first, m.last_logits(), k1)
    (_, ls, ks), (_, lb, kb) = outs["1"], outs["0"]
    scale = np.abs(ls).max()
    assert np.abs(ls - lb).max() <= 2.0 ** -7 * scale * np.sqrt(cfg.num_hidden_layers)
    # layer-1 keys depend on layer 0's full block output: agreement here checks GEMM + attention + MLP
    assert np.abs(ks - kb).max() <= 2.0 ** -6 * np.abs(ks).max()


@pytest.mark.parametrize("name", ["gqa4_d128", "gqa2_d64"])
def test_prefill_head_on_last_batched_row_matches_tail_step(omx, monkeypatch, name):
    """Default prefill pushes ALL prompt tokens through the matrix-core path and applies final norm + lm_head + sampler to the
    last row; OMX_PREFILL_TAIL_STEP=1 sends the last token through the decode step instead.  Two implementations of the same
    arithmetic (GEMM / flash attention vs GEMV / split-KV attention): logits agree to bf16 rounding, the cache offset and the
    tokens that follow are the oracle's either way."""
    cfg = CONFIGS[name]
    prompt = synth.prompt_ids(61, cfg.vocab_size)
    want, ref_logits = rq.Qwen3Oracle(cfg, rq.synth_weights(cfg)).generate(prompt, 6, return_logits=True)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OMX_PREFILL_TAIL_STEP", mode)
        m = _engine(omx, cfg)
        first = m.prefill(prompt)
        logits = m.last_logits()
        toks = [first] + [int(t) for t in m.decode(5)]
        outs[mode] = (toks, logits, m.offset())
        m.close()
    (tb, lb, ob), (ts, ls, os_) = outs["0"], outs["1"]
    assert ob == os_ == 61 + 5
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(lb - ls).max() <= bound
    assert np.abs(lb - ref_logits[0]).max() <= bound and np.abs(ls - ref_logits[0]).max() <= bound
    margins = rc.argmax_margin(ref_logits)
    for toks in (tb, ts):
        for i in range(6):
            if toks[i] != int(want[i]):
                assert margins[i] <= 2 * bound, f"token {i}: got {toks[i]} want {int(want[i])} with margin {margins[i]:.4f}"
                break


@pytest.mark.parametrize("name,T", [("gqa4_d128", 97), ("gqa2_d64", 300), ("mha_d128_linear_rope", 33)])
def test_short_prompt_qkv_in_one_ring_kernel_grid_is_bit_identical(omx, monkeypatch, name, T):
    """Short prompts: q/k/v are three segments of ONE grid of the 64 x 64 ring kernel (two of the three separate grids would be a
    few dozen blocks).  Every output element is computed by the same kernel with the same K order either way, so logits and
    tokens are EQUAL to the one-launch-per-projection schedule."""
    cfg = CONFIGS[name]
    prompt = synth.prompt_ids(T, cfg.vocab_size)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OMX_PREFILL_SEGMENTED", mode)
        m = _engine(omx, cfg)
        first = m.prefill(prompt)
        outs[mode] = (first, m.last_logits(), [int(t) for t in m.decode(4)])
        m.close()
    assert outs["1"][0] == outs["0"][0] and outs["1"][2] == outs["0"][2]
    np.testing.assert_array_equal(outs["1"][1], outs["0"][1])


def test_segmented_prefill_projections_match_separate_launches(omx, monkeypatch):
    """Long prompts run q/k/v as one segmented launch and gate/up/SiLU-mul as one launch with the activation in the GEMM
    epilogue (csrc/gemm.hpp: GemmSegs).  Same arithmetic per element as the separate launches up to the MFMA shape of the
    tile kernel that computes it: next-token logits and layer-1 keys agree to bf16 rounding, and the first token is the same
    (or a tie within that rounding)."""
    cfg = rq.Qwen3Config(1024, 2, 3072, 16, 8, 128, 2048, 1e-6, 1e6, False)
    T, cap = 2561, 2816
    prompt = synth.prompt_ids(T, cfg.vocab_size)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OMX_PREFILL_SEGMENTED", mode)
        m = _engine(omx, cfg, max_context=cap)
        first = m.prefill(prompt)
        n = cfg.num_key_value_heads * cap * cfg.head_dim
        raw = np.empty(n, np.uint16)
        omx.check(omx.lib.omx_qwen3_debug_read(m._h, b"k1", raw.ctypes.data, n))
        k1 = rc.from_bf16_bits(raw).reshape(cfg.num_key_value_heads, cap, cfg.head_dim)[:, :T]
        outs[mode] = (first, m.last_logits(), k1)
        m.close()
    (fs, ls, ks), (fp, lp, kp) = outs["1"], outs["0"]
    assert np.isfinite(ls).all() and np.abs(ls).max() > 0
    assert np.abs(ls - lp).max() <= 2.0 ** -7 * np.abs(lp).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(ks - kp).max() <= 2.0 ** -6 * np.abs(kp).max()
    tol = 2.0 ** -7 * np.abs(lp).max() * np.sqrt(cfg.num_hidden_layers)
    assert fs == fp or lp.ravel()[fs] >= lp.max() - tol


def test_tensor_parallel_code_path_with_one_rank_communicator(omx, monkeypatch):
    """The TP step (f32 partial GEMVs, RCCL all-reduce captured in the step graph, partial folded into
    the next prologue, packed-key argmax all-reduce) run with a 1-rank communicator must reproduce the
    non-TP engine token for token: with one rank the arithmetic is identical."""
    import bench
    cfg = CONFIGS["gqa4_d128"]
    prompt = synth.prompt_ids(48, cfg.vocab_size)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")     # the TP engine prefills token-serially; compare like with like
    plain = _engine(omx, cfg)
    want = np.concatenate([[plain.prefill(prompt)], plain.decode(10)])
    keep = bench.rccl_comm(None, 0, 1)
    m = _engine(omx, cfg)
    m.set_comm(keep[1], keep[2])
    got = np.concatenate([[m.prefill(prompt)], m.decode(10)])
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(m.last_logits(), plain.last_logits())
    # the batched tensor-parallel prefill on the same REAL communicator: two bf16 ncclAllReduce of [T, hidden] per layer, then the
    # step for the last token.  (One rank: the sums are the operands; the GEMM + all-reduce + add route rounds like the fused one.)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "0")
    m.reset(); plain.reset()
    want_b = np.concatenate([[plain.prefill(prompt)], plain.decode(6)])
    logits_p = plain.last_logits()
    got_b = np.concatenate([[m.prefill(prompt)], m.decode(6)])
    bound = 2.0 ** -7 * np.abs(logits_p).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(m.last_logits() - logits_p).max() <= bound
    assert (got_b == want_b).mean() >= 0.7, (got_b, want_b)


STEP_ATTN_CONFIGS = {
    "gqa4_d128": CONFIGS["gqa4_d128"],
    "gqa7_d128_qwen25_group": rq.Qwen3Config(896, 2, 1536, 7, 1, 128, 2048, 1e-6, 1e6, False),   # G = 7 -> the 8-head instantiation
    "mha_d64": rq.Qwen3Config(512, 2, 1024, 8, 8, 64, 2048, 1e-6, 1e6, True),
}


@pytest.mark.parametrize("name", list(STEP_ATTN_CONFIGS))
def test_step_attention_graph_equals_eager_across_context_buckets(omx, monkeypatch, name):
    """csrc/attn_step.hip: the split plan of the decode attention (token range per split) is part of the captured step graph
    and the graphs are rebuilt when the context enters another 1024-token bucket.  A generation that crosses the bucket
    boundary, replayed from graphs, must equal the same launches issued eagerly (OMX_NO_GRAPH=1) bit for bit -- and a
    reset() back into the first bucket must reproduce the first generation (granule tags keep counting, nothing is reset)."""
    cfg = STEP_ATTN_CONFIGS[name]
    prompt = synth.prompt_ids(1000, cfg.vocab_size)
    outs = {}
    for no_graph in ("0", "1"):
        monkeypatch.setenv("OMX_NO_GRAPH", no_graph)
        m = _engine(omx, cfg, max_context=1280)
        toks = np.concatenate([[m.prefill(prompt)], m.decode(60)])      # positions 1000 .. 1059: crosses 1024
        assert m.decode_path() == ("eager" if no_graph == "1" else "graph")
        logits = m.last_logits()
        m.reset()
        again = np.concatenate([[m.prefill(prompt[:50])], m.decode(20)])
        m.reset()
        again2 = np.concatenate([[m.prefill(prompt[:50])], m.decode(20)])
        np.testing.assert_array_equal(again, again2)
        outs[no_graph] = (toks, logits, again)
        m.close()
    for a, b in zip(outs["0"], outs["1"]):
        np.testing.assert_array_equal(a, b)


def test_step_attention_long_context_matches_oracle(omx):
    """More than 32 live splits (three gather batches of the consumer blocks) and several wave-instructions per wave: a
    1-layer model decoding at context 1100+ against the oracle, token-serial prefill through the same kernel included
    (its 1100 steps visit every position, split boundary and consumer configuration on the way)."""
    cfg = rq.Qwen3Config(256, 1, 512, 4, 2, 64, 512, 1e-6, 1e6, False)
    weights = rq.synth_weights(cfg)
    oracle = rq.Qwen3Oracle(cfg, weights)
    prompt = synth.prompt_ids(1100, cfg.vocab_size)
    n_new = 6
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)
    m = _engine(omx, cfg, max_context=1536)
    got = np.concatenate([[m.prefill(prompt)], m.decode(n_new - 1)])
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(m.last_logits() - ref_logits[-1]).max() <= 2 * bound
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f}"
            break


TP_CONFIGS = {
    "kv_split": CONFIGS["gqa4_d128"],                                                          # 8 heads / 2 KV heads over 2 ranks
    "kv_replicated": rq.Qwen3Config(1024, 2, 3072, 8, 1, 128, 4096, 1e-6, 1e6, False),          # 1 KV head: both ranks hold it
    # Qwen2 wiring under TP (q/k/v Linear with bias, no q/k norm; qwen2.rs:100-160): the biases are sharded like the rows
    "qwen2_bias_kv_split": rq.Qwen3Config(1024, 2, 3072, 8, 2, 128, 4096, 1e-6, 1e6, False, qk_norm=False, attention_bias=True),
    "qwen2_bias_kv_replicated": rq.Qwen3Config(1024, 2, 3072, 8, 1, 128, 4096, 1e-6, 1e6, False, qk_norm=False, attention_bias=True),
}


@pytest.mark.parametrize("name", list(TP_CONFIGS))
@pytest.mark.parametrize("serial_prefill", ["1", "0"])
def test_tensor_parallel_two_ranks_on_one_gpu(omx, monkeypatch, name, serial_prefill):
    """The TP path with REAL shards: two engine instances (half of the heads, KV heads -- or the one shared KV head --, MLP
    columns and vocabulary each; device-side synthetic shards of the same logical tensors) on one GPU, one host thread each,
    all-reducing through the in-process communicator (csrc/loopback_comm.hip) where bench.py hands the engine ncclAllReduce.
    serial_prefill 1: every prompt token is a decode step (f32 partials); 0: the batched matrix-core prefill on the shards with
    two bf16 all-reduces of [T, hidden] per layer (VERDICT r1 "Next" #2), then the step for the last token.  Both ranks must emit
    the same tokens; tokens and logits must agree with the oracle like the single-GPU engine does."""
    from ominix_mlx_amd import comm
    cfg = TP_CONFIGS[name]
    prompt = synth.prompt_ids(40, cfg.vocab_size)
    n_new = 10
    monkeypatch.setenv("OMX_PREFILL_SERIAL", serial_prefill)
    oracle = rq.Qwen3Oracle(cfg, rq.synth_weights(cfg))
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)
    world = 2
    group = comm.LoopbackGroup(world, 1 << 20)
    models = []
    from ominix_mlx_amd import engine
    for r in range(world):
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                         intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                         num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, vocab_size=cfg.vocab_size,
                         rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, tp_rank=r, tp_size=world,
                         qk_norm=cfg.qk_norm, attention_bias=cfg.attention_bias)
        if name.endswith("kv_replicated") and r == 1 or name == "qwen2_bias_kv_split" and r == 0:
            m.load_weights(rq.synth_weights(cfg))          # one rank from host arrays (tp.shard with the KV head rule), one generated
        else:
            m.synth_weights()
        m.set_comm(group.rank_comm(r), group.allreduce_fn)
        models.append(m)

    def run(r):
        m = models[r]
        first = m.prefill(prompt)
        logits0 = m.last_logits()
        rest = m.decode(n_new - 1)
        return np.concatenate([[first], rest]).astype(np.uint32), logits0, m.decode_path(), m.last_prefill_ms()

    outs = comm.run_ranks(world, run, group)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    assert outs[0][2] == "eager"                       # the loopback collective refuses stream capture
    got = outs[0][0]
    logits0 = np.concatenate([outs[0][1], outs[1][1]])          # vocabulary shards, rank order
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound * (1.0 if serial_prefill == "1" else 1.5)   # bf16 (not f32) partials in the batched pass
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f}"
            break
    for m in models:
        m.close()
    group.close()


@pytest.mark.parametrize("name,bits,serial_prefill", [("kv_split", 4, "1"), ("kv_split", 4, "0"), ("kv_replicated", 4, "0"), ("kv_split", 8, "1")])
def test_quantized_tensor_parallel_two_ranks_on_one_gpu(omx, monkeypatch, name, bits, serial_prefill):
    """Round 4: an MLX-quantized checkpoint under tensor parallelism.  q / k / v / gate / up / lm_head are the rank's packed rows (weight,
    scales, biases alike), o / down its K slices -- whole quantisation groups, so the slice of the triplet IS the triplet of the slice
    (tp.shard on the three leaves; the device generator quantises the rank's window of the synthetic bf16 matrix).  The packed GEMVs
    of the row-split projections leave unrounded f32 row sums (quant.hip EPI_F32), all-reduced and folded into the residual like the
    bf16 step's.  Rank 0 takes the host-sharded oracle triplets, rank 1 the generated ones; both must emit the same tokens, and
    tokens / logits must agree with the oracle on the unsharded quantized checkpoint within the quantized engine's bound."""
    from ominix_mlx_amd import comm, engine
    cfg = TP_CONFIGS[name]
    group_size = 64
    qw = rq.quantize_weights(cfg, rq.synth_weights(cfg), bits, group_size)
    oracle = rq.Qwen3Oracle(cfg, qw, quant=(bits, group_size))
    prompt = synth.prompt_ids(40, cfg.vocab_size)
    n_new = 10
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", serial_prefill)
    world = 2
    group = comm.LoopbackGroup(world, 1 << 20)
    models = []
    for r in range(world):
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                         num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                         vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, tp_rank=r, tp_size=world,
                         quantization={"bits": bits, "group_size": group_size})
        m.load_weights(qw) if r == 0 else m.synth_weights()
        m.set_comm(group.rank_comm(r), group.allreduce_fn)
        models.append(m)

    def run(r):
        first = models[r].prefill(prompt)
        logits0 = models[r].last_logits()
        return np.concatenate([[first], models[r].decode(n_new - 1)]).astype(np.uint32), logits0

    outs = comm.run_ranks(world, run, group)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    got = outs[0][0]
    logits0 = np.concatenate([outs[0][1], outs[1][1]])       # vocabulary shards
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(2 * cfg.num_hidden_layers)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound * (1.0 if serial_prefill == "1" else 1.5)   # bf16 partials in the batched pass
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f}"
            break
    for m in models:
        m.close()
    group.close()


@pytest.mark.parametrize("name,bits,group", [("gqa4_d128", 4, 64), ("gqa2_d64", 4, 64), ("gqa4_d128", 8, 64)])
def test_quantized_checkpoint_decode_matches_oracle(omx, name, bits, group):
    """MLX-quantized checkpoint (config.json "quantization", qwen3-mlx/src/model.rs:621-727): every Linear is a
    (weight, scales, biases) triplet, the embedding dequantises its rows, a tied head is as_linear.  The decode
    step streams the PACKED weights (csrc/quant.hip), the batched prefill dequantises per GEMM; tokens and logits
    against the oracle running quantized_matmul on the same triplets, with the tolerance of the bf16 engine.
    The device generator (synth_weights in quantized mode = mlx quantize() of the synthetic bf16 model) must give
    the same model as uploading the oracle's triplets."""
    from ominix_mlx_amd import engine
    cfg = CONFIGS[name]
    qw = rq.quantize_weights(cfg, rq.synth_weights(cfg), bits, group)
    oracle = rq.Qwen3Oracle(cfg, qw, quant=(bits, group))
    n_prompt, n_new = 48, 10
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)

    def make(upload):
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                         intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                         num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, vocab_size=cfg.vocab_size,
                         rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, tie_word_embeddings=cfg.tie_word_embeddings,
                         rope_scaling=cfg.rope_scaling, max_context=256, quantization={"bits": bits, "group_size": group})
        m.load_weights(qw) if upload else m.synth_weights()
        return m

    outs = []
    for upload in (True, False):
        m = make(upload)
        first = m.prefill(prompt)
        logits0 = m.last_logits()
        got = np.concatenate([[first], m.decode(n_new - 1)]).astype(np.uint32)
        assert m.decode_path() == "graph"
        outs.append((got, logits0))
        m.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    got, logits0 = outs[0]
    # the bf16 engine's bound times sqrt(2): the batched prefill multiplies with weights dequantised to bf16 (as MLX's qmm
    # does), one more 2^-9 relative perturbation per product than the oracle's float32 dequantisation
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(2 * cfg.num_hidden_layers)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f} > {2*bound:.4f}"
            break


def test_quantized_prompt_keeps_dequantised_matrices_between_prompts(omx, monkeypatch):
    """engine.hip dq_cache (round 4): a packed model's prompt pass dequantises each layer's matrices for its GEMMs; with HBM to spare the
    dequantised copies are kept, so every prompt after the first skips those launches.  Same values either way: the first prompt, the
    same prompt again on the emptied cache, and a run with OMX_DEQUANT_CACHE=0 give bit-identical logits and tokens."""
    from ominix_mlx_amd import engine
    cfg = CONFIGS["gqa4_d128"]
    prompt = synth.prompt_ids(80, cfg.vocab_size)
    kw = dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
              num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
              vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
              tie_word_embeddings=cfg.tie_word_embeddings, rope_scaling=cfg.rope_scaling, max_context=256)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OMX_DEQUANT_CACHE", mode)
        m = engine.Model(quantization={"bits": 4, "group_size": 64}, **kw)
        m.synth_weights()
        out = []
        for _ in range(3):
            m.reset()
            first = m.prefill(prompt)
            out.append((first, m.last_logits().copy(), m.decode(4).copy()))
        runs[mode] = out
        m.close()
    ref = runs["0"][0]
    for mode in ("1", "0"):
        for first, logits, toks in runs[mode]:
            assert first == ref[0]
            np.testing.assert_array_equal(logits, ref[1])
            np.testing.assert_array_equal(toks, ref[2])


@pytest.mark.parametrize("serial_prefill", ["0", "1"])
def test_quantized_checkpoint_with_float16_scales(omx, tmp_path, monkeypatch, serial_prefill):
    """(serial_prefill 0: the 48-token prompt takes the float16 matrix-core pass, 1: the decode step token by token.)
    A float16 MLX checkpoint carries float16 scales / biases and norm weights (mlx quantize() works in the model's dtype), and MLX
    runs it in float16 END TO END (nn/quantized.rs:361-385: the dequantised weight has the scales' dtype, the matmul its inputs').
    So does the engine since round 4: embedding rows dequantised to float16, float16 RMSNorm / RoPE / residual roundings, float16 K / V
    slabs and logits, f32 accumulation (csrc/act16.hpp Act16<true> in quant.hip, attn_step.hip, engine.hip; the prompt runs through
    the decode step).  Against the oracle in float16 on the same float16 VALUES, with the bound scaled to float16's 11-bit results;
    also through loader.load_model (dtype read off the safetensors header), and the loud mismatch errors."""
    from ominix_mlx_amd import engine
    cfg, bits, group = CONFIGS["gqa4_d128"], 4, 64
    base = rq.synth_weights(cfg)
    qw = {}
    for name, arr in rq.quantize_weights(cfg, base, bits, group).items():
        if name.endswith((".scales", ".biases")):
            prefix = name.rsplit(".", 1)[0]
            w2 = base[prefix + ".weight"].reshape(-1, base[prefix + ".weight"].shape[-1])
            _, s32, b32 = rc.quantize(w2, group, bits)
            arr = (s32 if name.endswith(".scales") else b32).astype(np.float16).reshape(arr.shape)
        qw[name] = arr
    n_prompt, n_new = 48, 10
    monkeypatch.setenv("OMX_PREFILL_SERIAL", serial_prefill)
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    kw = dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
              num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
              vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
              tie_word_embeddings=cfg.tie_word_embeddings, rope_scaling=cfg.rope_scaling, max_context=256)
    m = engine.Model(quantization={"bits": bits, "group_size": group, "scales_dtype": "float16"}, **kw)
    m.load_weights(qw)
    first = m.prefill(prompt)
    logits0 = m.last_logits()
    got = np.concatenate([[first], m.decode(n_new - 1)]).astype(np.uint32)
    assert m.decode_path() == "graph"
    with pytest.raises(omx.OmxError, match="float16-scale model takes uploaded triplets"):
        m.synth_weights()
    m.close()
    f16w = {k: (v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in qw.items()}     # the f16 VALUES, exactly
    oracle = rq.Qwen3Oracle(cfg, f16w, dt="f16", quant=(bits, group))
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)
    bound = 2.0 ** -10 * np.abs(ref_logits).max() * np.sqrt(2 * cfg.num_hidden_layers)   # test_quantized_checkpoint_decode_matches_oracle's, 3 bits finer
    # (and it is NOT the bf16-activation computation any more: that one sits outside this bound)
    ref_bf16 = rq.Qwen3Oracle(cfg, f16w, quant=(bits, group)).generate(prompt, 1, return_logits=True)[1]
    assert np.abs(ref_bf16[0] - ref_logits[0]).max() > bound
    assert np.abs(logits0 - ref_logits[0]).max() <= bound
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f} > {2*bound:.4f}"
            break
    # a bf16-scale model refuses float16 triplets (and the other way round) instead of reading them as the wrong format
    m2 = engine.Model(quantization={"bits": bits, "group_size": group}, **kw)
    with pytest.raises(omx.OmxError, match="float16 scales / biases, but the model was created"):
        m2.load_weights(qw)
    m2.close()


def _f16_triplets(cfg, bits=4, group=64):
    base = rq.synth_weights(cfg)
    qw = {}
    for name, arr in rq.quantize_weights(cfg, base, bits, group).items():
        if name.endswith((".scales", ".biases")):
            prefix = name.rsplit(".", 1)[0]
            w2 = base[prefix + ".weight"].reshape(-1, base[prefix + ".weight"].shape[-1])
            _, s32, b32 = rc.quantize(w2, group, bits)
            arr = (s32 if name.endswith(".scales") else b32).astype(np.float16).reshape(arr.shape)
        qw[name] = arr
    return qw


def test_float16_checkpoint_samples_and_encodes(omx):
    """a float16 checkpoint is not a greedy-only, prompt-only model.  (1) temperature > 0: the noise kernel reads
    the float16 logits row (random.hip sample_noise_kernel<true>), widened exactly to f32 before the 1/T product as
    `logits * array!(1/T)` promotes (model.rs:733-741, sampler.rs:9-18) -- the drawn tokens are the oracle's draws from the engine's
    own float16 logits, on the graph path.  (2) `encode` (qwen3_encoder.rs:403-455) without a padding mask, 77 rows and 5 rows (the
    128-row float16 GEMM tile with a handful of live rows): hidden-state taps in float16 against the float16 oracle within the
    float16 bound.  With a padding mask the reference computes 0 * f16(-1e9) = NaN (qwen3_encoder.rs:196-198): refused, by name."""
    from ominix_mlx_amd import engine
    from oracle import mlx_rng as rng
    cfg, bits, group = CONFIGS["gqa4_d128"], 4, 64
    qw = _f16_triplets(cfg, bits, group)
    kw = dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
              num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
              vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
              tie_word_embeddings=cfg.tie_word_embeddings, rope_scaling=cfg.rope_scaling, max_context=256)
    m = engine.Model(quantization={"bits": bits, "group_size": group, "scales_dtype": "float16"}, **kw)
    m.load_weights(qw)
    temp, seed = 0.8, 11
    m.set_sampler(temp, seed)
    state = rng.RandomState(seed)
    for n_prompt in (12, 40):                      # through the decode step, and through the float16 matrix-core pass
        m.reset()
        prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
        toks, logits = [m.prefill(prompt)], [m.last_logits()]
        for _ in range(8):
            toks.append(int(m.decode(1)[0]))
            logits.append(m.last_logits())
        assert m.decode_path() == "graph"
        assert all(l.dtype == np.float32 and np.array_equal(l, l.astype(np.float16).astype(np.float32)) for l in logits)   # float16 values
        want = [int(rc.sample(l[None, :], temp, state.next())[0]) for l in logits]
        assert toks == want
        assert toks != [int(np.argmax(l)) for l in logits]
    m.set_sampler(0.0)
    # the encoder taps
    f16w = {k: (v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in qw.items()}
    oracle = rq.Qwen3Oracle(cfg, f16w, dt="f16", quant=(bits, group))
    taps = (0, cfg.num_hidden_layers - 1)
    for n in (77, 5):
        ids = synth.prompt_ids(n, cfg.vocab_size)
        got_t = m.encode(ids, None, taps)
        got = got_t.numpy().astype(np.float32)
        want = oracle.encode(ids, None, taps)
        assert got.shape == want.shape == (n, len(taps) * cfg.hidden_size)
        bound = 2.0 ** -10 * np.abs(want).max() * np.sqrt(2 * cfg.num_hidden_layers)
        assert np.abs(got - want).max() <= bound, f"float16 encode of {n} rows off by {np.abs(got - want).max():.5f} (bound {bound:.5f})"
    am = np.ones(77, np.uint8); am[60:] = 0
    with pytest.raises(omx.OmxError, match="NaN in float16"):
        m.encode(synth.prompt_ids(77, cfg.vocab_size), am, taps)
    m.close()


def test_float16_checkpoint_batched_prompt_pass(omx, monkeypatch):
    """Round 4: a float16 checkpoint's prompt in ONE matrix-core pass (engine.hip prefill_prefix_batched with f16: weights dequantised
    to float16, the eight-wave GEMM kernel's float16 instantiations with the SwiGLU / residual epilogues rounding to float16, float16
    norm + RoPE + slab scatter, attention in float32 on widened copies with one rounding) instead of one decode step per token.  A
    300-token prompt (two 256-row tiles, a ragged second one) plus a 40-token follow-up on top of the cache: first-token logits within
    the float16 bound of the float16 oracle AND of the token-serial pass, layer-1 keys equal to float16 rounding, same tokens after."""
    from ominix_mlx_amd import engine
    cfg, bits, group = CONFIGS["gqa4_d128"], 4, 64
    base = rq.synth_weights(cfg)
    qw = {}
    for name, arr in rq.quantize_weights(cfg, base, bits, group).items():
        if name.endswith((".scales", ".biases")):
            prefix = name.rsplit(".", 1)[0]
            w2 = base[prefix + ".weight"].reshape(-1, base[prefix + ".weight"].shape[-1])
            _, s32, b32 = rc.quantize(w2, group, bits)
            arr = (s32 if name.endswith(".scales") else b32).astype(np.float16).reshape(arr.shape)
        qw[name] = arr
    n1, n2, n_new = 300, 40, 6
    prompt = synth.prompt_ids(n1 + n2, cfg.vocab_size)
    kw = dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
              num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
              vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
              tie_word_embeddings=cfg.tie_word_embeddings, rope_scaling=cfg.rope_scaling, max_context=512)
    runs = {}
    for serial in ("0", "1"):
        monkeypatch.setenv("OMX_PREFILL_SERIAL", serial)
        m = engine.Model(quantization={"bits": bits, "group_size": group, "scales_dtype": "float16"}, **kw)
        m.load_weights(qw)
        first = m.prefill(prompt[:n1])
        logits1 = m.last_logits()
        n = cfg.num_key_value_heads * 512 * cfg.head_dim
        raw = np.empty(n, np.uint16)
        omx.check(omx.lib.omx_qwen3_debug_read(m._h, b"k1", raw.ctypes.data, n))
        raw = raw.view(np.float16).astype(np.float32).reshape(cfg.num_key_value_heads, 512, cfg.head_dim)[:, :n1]
        m.prefill(prompt[n1:])              # (the first call's sampled token is replaced by the follow-up prompt, as in a chat turn)
        logits2 = m.last_logits()
        toks = m.decode(n_new)
        runs[serial] = (first, logits1, logits2, toks, raw, m.last_prefill_ms())
        m.close()
    f16w = {k: (v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in qw.items()}
    oracle = rq.Qwen3Oracle(cfg, f16w, dt="f16", quant=(bits, group))
    _, ref1 = oracle.generate(prompt[:n1], 1, return_logits=True)
    _, ref2 = oracle.generate(prompt, 1, return_logits=True)
    for ref, idx in ((ref1, 1), (ref2, 2)):
        bound = 2.0 ** -10 * np.abs(ref).max() * np.sqrt(2 * cfg.num_hidden_layers)
        for serial in ("0", "1"):
            assert np.abs(runs[serial][idx] - ref[0]).max() <= bound, (serial, idx)
        assert np.abs(runs["0"][idx] - runs["1"][idx]).max() <= bound
    # layer-1 keys depend on layer 0's whole block (GEMMs, attention, SwiGLU): the two passes agree to float16 rounding
    assert np.abs(runs["0"][4] - runs["1"][4]).max() <= 2.0 ** -9 * np.abs(runs["1"][4]).max()
    bound1 = 2.0 ** -10 * np.abs(ref1).max() * np.sqrt(2 * cfg.num_hidden_layers)
    if runs["0"][0] != runs["1"][0]:
        assert rc.argmax_margin(ref1)[0] <= 2 * bound1
    print(f"float16 prompt of {n1} tokens: batched {runs['0'][5]:.2f} ms, token-serial {runs['1'][5]:.2f} ms")


@pytest.mark.parametrize("serial_prefill", ["0", "1"])
def test_float16_checkpoint_tensor_parallel_two_ranks_on_one_gpu(omx, monkeypatch, serial_prefill):
    """(serial_prefill 0: the prompt in the float16 matrix-core pass on the shards -- the row-split projections' float16 partial products
    widened and all-reduced in f32; 1: through the decode step.)
    Round 4 (review of round 3: "TP = 2 loopback with f16 triplets"): a float16 MLX checkpoint sharded over two ranks -- packed rows
    and K slices with their float16 scales / biases (tp.shard), float16 activations, K / V and logits on every rank, the row-split
    projections' f32 partials all-reduced and folded into the float16 residual (two float16 roundings, as the single-rank packed GEMV's
    residual epilogue makes them).  Both ranks emit the same tokens; against the float16 oracle on the UNSHARDED checkpoint the logits
    stay within the float16 engine's 2^-10-scaled bound (partial sums round differently: sqrt(2) on top, as for bf16 TP)."""
    from ominix_mlx_amd import comm, engine
    cfg, bits, group_size = TP_CONFIGS["kv_split"], 4, 64
    base = rq.synth_weights(cfg)
    qw = {}
    for name, arr in rq.quantize_weights(cfg, base, bits, group_size).items():
        if name.endswith((".scales", ".biases")):
            prefix = name.rsplit(".", 1)[0]
            w2 = base[prefix + ".weight"].reshape(-1, base[prefix + ".weight"].shape[-1])
            _, s32, b32 = rc.quantize(w2, group_size, bits)
            arr = (s32 if name.endswith(".scales") else b32).astype(np.float16).reshape(arr.shape)
        qw[name] = arr
    prompt = synth.prompt_ids(40, cfg.vocab_size)
    n_new = 10
    monkeypatch.setenv("OMX_PREFILL_SERIAL", serial_prefill)
    f16w = {k: (v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in qw.items()}
    ref_tokens, ref_logits = rq.Qwen3Oracle(cfg, f16w, dt="f16", quant=(bits, group_size)).generate(prompt, n_new, return_logits=True)
    world = 2
    group = comm.LoopbackGroup(world, 1 << 20)
    models = []
    for r in range(world):
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                         num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                         vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, tp_rank=r, tp_size=world,
                         quantization={"bits": bits, "group_size": group_size, "scales_dtype": "float16"})
        m.load_weights(qw)
        m.set_comm(group.rank_comm(r), group.allreduce_fn)
        models.append(m)

    def run(r):
        first = models[r].prefill(prompt)
        logits0 = models[r].last_logits()
        return np.concatenate([[first], models[r].decode(n_new - 1)]).astype(np.uint32), logits0

    outs = comm.run_ranks(world, run, group)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    got, logits0 = outs[0][0], np.concatenate([outs[0][1], outs[1][1]])
    bound = 2.0 ** -10 * np.abs(ref_logits).max() * np.sqrt(2 * cfg.num_hidden_layers) * np.sqrt(2)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound, f"{np.abs(logits0 - ref_logits[0]).max():.5f} > {bound:.5f}"
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f}"
            break
    for m in models:
        m.close()
    group.close()


@pytest.mark.parametrize("quant", [None, {"bits": 4, "group_size": 64}, {"bits": 4, "group_size": 64, "scales_dtype": "float16"}])
def test_load_model_from_checkpoint_directory(omx, tmp_path, quant):
    """qwen3_mlx::load_model (model.rs:509-560, 621-727): config.json + model.safetensors.index.json + two shards
    (BF16 tensors as raw bits, packed U32 weights for the quantized variant) -> the same tokens and logits as handing
    the same tensors to the engine directly."""
    import json
    from ominix_mlx_amd import engine, loader
    cfg = CONFIGS["gqa4_d128"]
    w = rq.synth_weights(cfg)
    f16 = bool(quant) and quant.get("scales_dtype") == "float16"     # a float16 checkpoint: F16 scales / biases in the shards, and
    if quant:                                                         # config.json does not say so -- the loader reads the dtype
        w = rq.quantize_weights(cfg, w, quant["bits"], quant["group_size"])
        if f16:
            w = {k: (v.astype(np.float16) if k.endswith((".scales", ".biases")) else v) for k, v in w.items()}
        quant_cfg = {"bits": quant["bits"], "group_size": quant["group_size"]}
    d = str(tmp_path)
    json.dump({"hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_hidden_layers, "intermediate_size": cfg.intermediate_size,
               "num_attention_heads": cfg.num_attention_heads, "num_key_value_heads": cfg.num_key_value_heads, "head_dim": cfg.head_dim,
               "vocab_size": cfg.vocab_size, "rms_norm_eps": cfg.rms_norm_eps, "rope_theta": cfg.rope_theta,
               "tie_word_embeddings": cfg.tie_word_embeddings, **({"quantization": quant_cfg} if quant else {})},
              open(f"{d}/config.json", "w"))
    names = sorted(w)
    shards = {"model-00001-of-00002.safetensors": names[: len(names) // 2], "model-00002-of-00002.safetensors": names[len(names) // 2:]}
    for fn, keys in shards.items():
        raw = lambda k: w[k].dtype in (np.uint32, np.float16)
        tensors = {k: (w[k] if raw(k) else rc.to_bf16_bits(w[k])) for k in keys}
        loader.write_safetensors(f"{d}/{fn}", tensors, bf16_names=tuple(k for k in keys if not raw(k)))
    json.dump({"metadata": {}, "weight_map": {k: fn for fn, keys in shards.items() for k in keys}}, open(f"{d}/model.safetensors.index.json", "w"))

    prompt = synth.prompt_ids(20, cfg.vocab_size)
    m = loader.load_model(d, max_context=256)
    got = np.concatenate([[m.prefill(prompt)], m.decode(6)])
    logits = m.last_logits()
    ref = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                       num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                       vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                       tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, quantization=quant)
    ref.load_weights(w)
    want = np.concatenate([[ref.prefill(prompt)], ref.decode(6)])
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(logits, ref.last_logits())


def test_tensor_parallel_load_from_bf16_checkpoint_files(omx, tmp_path):
    """a REAL BF16 safetensors checkpoint loaded with tp_size=2 goes through tp.shard, whose numpy copies
    used to drop the raw-bits marking (weights uploaded as 16256.0 instead of 1.0).  Two ranks loading their shards from
    the files must reproduce the ranks that were handed float arrays, bit for bit, and agree with the single-GPU engine."""
    import json
    from ominix_mlx_amd import comm, engine, loader
    cfg = CONFIGS["gqa4_d128"]
    w = rq.synth_weights(cfg)
    d = str(tmp_path)
    json.dump({"hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_hidden_layers, "intermediate_size": cfg.intermediate_size,
               "num_attention_heads": cfg.num_attention_heads, "num_key_value_heads": cfg.num_key_value_heads, "head_dim": cfg.head_dim,
               "vocab_size": cfg.vocab_size, "rms_norm_eps": cfg.rms_norm_eps, "rope_theta": cfg.rope_theta,
               "tie_word_embeddings": cfg.tie_word_embeddings}, open(f"{d}/config.json", "w"))
    loader.write_safetensors(f"{d}/model.safetensors", {k: rc.to_bf16_bits(v) for k, v in w.items()}, bf16_names=tuple(w))
    prompt = synth.prompt_ids(24, cfg.vocab_size)
    world = 2

    def run_world(make):
        group = comm.LoopbackGroup(world, 1 << 20)
        models = [make(r) for r in range(world)]
        for r, m in enumerate(models):
            m.set_comm(group.rank_comm(r), group.allreduce_fn)

        def run(r):
            m = models[r]
            return np.concatenate([[m.prefill(prompt)], m.decode(5)]).astype(np.uint32), m.last_logits()

        outs = comm.run_ranks(world, run, group)
        for m in models:
            m.close()
        group.close()
        return outs

    from_files = run_world(lambda r: loader.load_model(d, max_context=256, tp_rank=r, tp_size=world))

    def from_arrays(r):
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                         num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                         vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, tp_rank=r, tp_size=world)
        m.load_weights(w)
        return m

    want = run_world(from_arrays)
    for r in range(world):
        np.testing.assert_array_equal(from_files[r][0], want[r][0])
        np.testing.assert_array_equal(from_files[r][1], want[r][1])
    single = _engine(omx, cfg, w, max_context=256)
    toks = np.concatenate([[single.prefill(prompt)], single.decode(5)]).astype(np.uint32)
    bound = 2.0 ** -7 * np.abs(single.last_logits()).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(np.concatenate([from_files[0][1], from_files[1][1]]) - single.last_logits()).max() <= bound
    assert toks[0] == from_files[0][0][0]


def test_load_weights_rejects_a_tensor_of_the_wrong_shape(omx):
    """the engine reads raw pointers, so a shape that disagrees with the config must be an error at load."""
    from ominix_mlx_amd import OmxError, engine
    cfg = CONFIGS["gqa2_d64"]
    w = rq.synth_weights(cfg)
    bad = dict(w)
    bad["model.layers.0.mlp.down_proj.weight"] = w["model.layers.0.mlp.down_proj.weight"][:, :-64]
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                     vocab_size=cfg.vocab_size, max_context=64)
    with pytest.raises(OmxError, match="ShapeMismatch: model.layers.0.mlp.down_proj.weight"):
        m.load_weights(bad)
    m.close()


def test_time_step_kernels_runs_real_steps(omx):
    """bench.py's roofline hook (omx_qwen3_time_step_kernels): the event-bracketed steps are ordinary decode steps -- a generation
    with two of them in the middle emits the same tokens as one without -- and every class reports a positive duration."""
    cfg = CONFIGS["gqa4_d128"]
    prompt = synth.prompt_ids(48, cfg.vocab_size)
    a = _engine(omx, cfg, max_context=256)
    want = np.concatenate([[a.prefill(prompt)], a.decode(12)])
    a.close()
    want = np.concatenate([want[:5], want[7:]])          # the two tokens of the timed steps stay in the engine's ring
    b = _engine(omx, cfg, max_context=256)
    got = [b.prefill(prompt)] + list(b.decode(4))
    us = b.time_step_kernels(2)
    got += list(b.decode(6))
    assert set(us) == {"qkv", "attention", "o", "gate_up", "down", "lm_head", "step_engine"} and us["step_engine"] == 0.0
    # ("o" is 0 when the O projection rides in the attention launch -- the default wherever the shape qualifies)
    assert all(0 < v < 1000 for k, v in us.items() if k not in ("o", "step_engine")) and 0 <= us["o"] < 1000, us
    np.testing.assert_array_equal(np.array(got, np.uint32), want.astype(np.uint32))
    b.close()


OPROJ_CONFIGS = dict(CONFIGS)
# K = H * D = 2048 -> the NVW = 4 register layout (Qwen3-0.6B's attention shape); 8 KV heads -> 32 splits at most, 3 gather batches never
OPROJ_CONFIGS["h16_kv8_d128_nvw4"] = rq.Qwen3Config(1024, 2, 2048, 16, 8, 128, 2048, 1e-6, 1e6, True)
# K = 28 x 128 = 3584 -> NVW = 7 (Qwen2.5-7B's attention shape: G = 7 runs on the 8-head instantiation), no q/k norm, biases
OPROJ_CONFIGS["h28_kv4_d128_nvw7_qwen25"] = rq.Qwen3Config(1024, 1, 1536, 28, 4, 128, 1024, 1e-6, 1e6, False, qk_norm=False, attention_bias=True)
# K = 4096 -> NVW = 8 with a hidden size that needs 2 rows per producer wave (the full-size shape is covered by test_gpu_fullsize_pin.py)
OPROJ_CONFIGS["h32_kv8_d128_nvw8"] = rq.Qwen3Config(2048, 1, 2048, 32, 8, 128, 1024, 1e-6, 1e6, False)


@pytest.mark.parametrize("name", list(OPROJ_CONFIGS))
def test_oproj_in_attention_launch_is_bit_identical(omx, monkeypatch, name):
    """csrc/attn_step.hip with the layer's O projection + residual in the attention launch (weight rows prefetched into registers
    while the attention runs, the merged heads handed over as tagged granules) reproduces the separate O GEMV's arithmetic exactly:
    same tokens and bit-equal logits as OMX_ATTN_OPROJ=0, in graph and eager form, across a context-bucket boundary."""
    cfg = OPROJ_CONFIGS[name]
    prompt = synth.prompt_ids(1000, cfg.vocab_size)
    outs = {}
    for mode in ("0", "1", "eager"):
        monkeypatch.setenv("OMX_ATTN_OPROJ", "0" if mode == "0" else "1")
        monkeypatch.setenv("OMX_NO_GRAPH", "1" if mode == "eager" else "0")
        m = _engine(omx, cfg, max_context=1280)
        toks = np.concatenate([[m.prefill(prompt)], m.decode(40)])      # positions 1000 .. 1039: crosses 1024
        outs[mode] = (toks, m.last_logits())
        m.close()
    for mode in ("1", "eager"):
        for a, b in zip(outs["0"], outs[mode]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name,group", [("h16_kv8_d128_nvw4", 64), ("h32_kv8_d128_nvw8", 64), ("h32_kv8_d128_nvw8", 32), ("h16_kv8_d128_nvw4", 128)])
def test_packed_oproj_in_attention_launch_is_bit_identical(omx, monkeypatch, name, group):
    """4-bit checkpoint: the packed O matrix rides in the attention launch too (csrc/attn_step.hip oproj_phase_q4: the rows' nibbles and
    scale | bias words prefetched into registers, the attention vector re-paired in LDS the way quant.hip's prologue stores it).  Same
    tokens and bit-equal logits as the separate packed GEMV (OMX_ATTN_OPROJ=0), graph and eager, across a context-bucket boundary;
    the two-launch form is the one test_quantized_checkpoint_decode_matches_oracle holds against the oracle.
    Round 6: bit-equality is a property of the two VALU forms (same fma chains) -- the separate GEMV of a K = 4096 matrix would otherwise
    run on the matrix cores (csrc/qgemv_mfma.hip, its own accumulation order; held against the oracle in test_gpu_quant.py), so the
    comparison pins OMX_QGEMV_MFMA=0."""
    from ominix_mlx_amd import engine
    monkeypatch.setenv("OMX_QGEMV_MFMA", "0")
    cfg = OPROJ_CONFIGS[name]
    prompt = synth.prompt_ids(1000, cfg.vocab_size)
    outs = {}
    for mode in ("0", "1", "eager"):
        monkeypatch.setenv("OMX_ATTN_OPROJ", "0" if mode == "0" else "1")
        monkeypatch.setenv("OMX_NO_GRAPH", "1" if mode == "eager" else "0")
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                         num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                         vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, rope_scaling=cfg.rope_scaling, max_context=1280,
                         quantization={"bits": 4, "group_size": group})
        m.synth_weights()
        toks = np.concatenate([[m.prefill(prompt)], m.decode(40)])      # positions 1000 .. 1039: crosses 1024
        outs[mode] = (toks, m.last_logits())
        m.close()
    for mode in ("1", "eager"):
        for a, b in zip(outs["0"], outs[mode]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_peer_store_allreduce_across_processes(omx, tmp_path, fused):
    """csrc/peer_allreduce.hip with one PROCESS per rank (tools/peer_allreduce_check.py under torch.distributed.run, gloo bootstrap,
    both ranks on this box's one GPU): inboxes exchanged as HIP IPC handles, every all-reduce of the tensor-parallel step (f32 hidden
    partials, u64 argmax key) one kernel of tagged peer stores, captured in the step graph.  The self-test must reproduce the
    rank-ordered sums exactly, both ranks must emit the same tokens, and -- the reduction order being the loopback communicator's --
    exactly the tokens of the in-process two-rank run.  fused 1: the O / down GEMVs reduce their own output rows over the peers in
    their epilogue (gemv.hip EPI_F32 + peer: no all-reduce launch at all); 0: the standalone kernel after each of them."""
    import json
    import subprocess
    import sys
    from ominix_mlx_amd import comm, engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + (os.getpid() % 400)
    # (fused 0 also shrinks the two-shot path's stage to 4 MB: its 16.7 MB messages then go in five chunks -- and asks for the
    #  SYSTEM-scope hand-offs that ranks on different GPUs get by default; fused 1 keeps what two ranks on one GPU get: agent scope)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMX_PEER_FUSED=fused, **({"OMX_PEER_STAGE_MB": "4", "OMX_PEER_SCOPE": "system"} if fused == "0" else {}))
    env.pop("OMX_PEER_SCOPE", None) if fused == "1" else None
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tools", "peer_allreduce_check.py"), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=600, stdin=subprocess.DEVNULL)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r in res:
        assert r["self_test"] and not r["aborted"] and not r["aborted_after_loop"]
        assert r["same_device"] and r["scope"] == ("system" if fused == "0" else "agent")      # (round 5: both hand-off forms, chosen as comm.py documents)
        # per-call latency report (what a first multi-GPU run is read from): three sizes, ordered percentiles, every call counted
        assert [h["n_floats"] for h in r["latency_us"]] == [4096, 2, 4096 * 64]
        for h in r["latency_us"]:
            assert 0 < h["min"] <= h["p50"] <= h["p90"] <= h["p99"] <= h["max"] and sum(h["log2_buckets_us"].values()) == h["calls"]
        # the two-shot path: seeded messages of 32 KB .. 16.7 MB in f32 and bf16 come back as the rank-ordered sum, bit for bit
        assert len(r["large"]) == 6 and all(c["rc"] == 0 and c["equal"] for c in r["large"]), r["large"]
        assert not r["aborted_after_large"]
        # expert parallel: the MoE block's weighted sum as the all-to-all combine + all-gather kernel against the f32 all-reduce of the
        # same prompt -- same tokens, same last logits (top-2: the two products of a token sum to the same f32 in either order)
        assert r["stage_bytes"] > 0 and not r["aborted_after_ep"]
        assert r["ep_exchange"]["tokens"] == r["ep_allreduce"]["tokens"] and r["ep_exchange"]["logits_crc"] == r["ep_allreduce"]["logits_crc"]
        # (which kernels ran: one combine per layer and no two-shot all-reduce in the exchange form, the reverse in the other)
        assert r["ep_exchange"]["launches"]["moe_combine"] == 2 and r["ep_exchange"]["launches"]["two_shot"] == 0, r["ep_exchange"]["launches"]
        assert r["ep_allreduce"]["launches"]["moe_combine"] == 0 and r["ep_allreduce"]["launches"]["two_shot"] >= 2, r["ep_allreduce"]["launches"]
    assert res[0]["ep_exchange"]["tokens"] == res[1]["ep_exchange"]["tokens"] and res[0]["ep_exchange"]["logits_crc"] == res[1]["ep_exchange"]["logits_crc"]
    assert res[0]["tokens"] == res[1]["tokens"]
    assert res[0]["tokens_batched"] == res[1]["tokens_batched"]
    assert res[0]["decode_path"] == "graph"          # the peer all-reduce is an ordinary kernel: the step stays captured
    # the same two shards through the in-process communicator (same rank-ordered f32 sums)
    cfg = dict(hidden_size=1024, num_hidden_layers=2, intermediate_size=3072, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
               vocab_size=4096, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
    os.environ["OMX_PREFILL_SERIAL"] = "1"
    try:
        group = comm.LoopbackGroup(2, 1 << 20)
        models = []
        for r in range(2):
            m = engine.Model(max_context=256, tp_rank=r, tp_size=2, **cfg)
            m.synth_weights()
            m.set_comm(group.rank_comm(r), group.allreduce_fn)
            models.append(m)
        prompt = synth.prompt_ids(40, cfg["vocab_size"])

        def run(r):
            first = models[r].prefill(prompt)
            return [int(first)] + [int(x) for x in models[r].decode(15)]

        outs = comm.run_ranks(2, run, group)
    finally:
        del os.environ["OMX_PREFILL_SERIAL"]
    assert outs[0] == res[0]["tokens"]
    # the batched prompt through the in-process communicator: its bf16 all-reduce sums in rank order in f32 and rounds once, like the
    # two-shot path's slice owners
    os.environ["OMX_PREFILL_SERIAL"] = "0"
    try:
        for mm in models:
            mm.reset()
        outs_b = comm.run_ranks(2, run, group)
    finally:
        del os.environ["OMX_PREFILL_SERIAL"]
    assert outs_b[0] == res[0]["tokens_batched"]
    # ... and the expert-parallel pair against the single-GPU sparse-MoE engine (batched prompt, grouped GEMMs over all experts)
    moe_cfg = dict(hidden_size=512, num_hidden_layers=2, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=64,
                   vocab_size=2048, rms_norm_eps=1e-5, rope_theta=1e6, tie_word_embeddings=False, num_experts=8, num_experts_per_tok=2,
                   moe_intermediate_size=1024, moe_mode="mixtral", norm_topk_prob=0, qk_norm=False)
    single = engine.Model(max_context=512, **moe_cfg)
    single.synth_weights()
    moe_prompt = synth.prompt_ids(200, moe_cfg["vocab_size"])
    want = [int(single.prefill(moe_prompt))] + [int(x) for x in single.decode(6)]
    single.reset()
    want += [int(single.prefill(moe_prompt[:9]))] + [int(x) for x in single.decode(3)]
    single.close()
    assert res[0]["ep_exchange"]["tokens"] == want
    print(f"peer all-reduce of 16 KB, two processes on one GPU: {res[0]['allreduce_16k_us']:.1f} us per call; "
          f"TP = 2 step (fused={fused}): {res[0]['step_ms']:.3f} ms")


def test_peer_communicator_four_processes_on_one_gpu(omx, tmp_path):
    """The same tool with FOUR ranks sharing this GPU: four slices / owners in the two-shot all-reduce and in the expert-parallel exchange
    (8 experts: two per rank, tokens owned in quarters), KV heads one per two ranks in the TP 4 model.  Every rank's self-test and seeded
    large messages exact, all ranks emit the same tokens in every section, the exchange combine equals the all-reduce form bit for bit
    and the single-GPU sparse-MoE engine's tokens."""
    import json
    import subprocess
    import sys
    from ominix_mlx_amd import engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29900 + (os.getpid() % 90)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMX_PEER_FUSED="0", OMX_PEER_SCOPE="system")     # (the form ranks on different GPUs get)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tools", "peer_allreduce_check.py"), str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=900, stdin=subprocess.DEVNULL)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(4)]
    for r in res:
        assert r["self_test"] and not r["aborted"] and not r["aborted_after_large"] and not r["aborted_after_ep"]
        assert r["scope"] == "system"
        assert all(c["rc"] == 0 and c["equal"] for c in r["large"]), r["large"]
        assert r["ep_exchange"]["tokens"] == r["ep_allreduce"]["tokens"] and r["ep_exchange"]["logits_crc"] == r["ep_allreduce"]["logits_crc"]
        assert r["ep_exchange"]["launches"]["moe_combine"] == 2 and r["ep_exchange"]["launches"]["two_shot"] == 0
        for key in ("tokens", "tokens_batched"):
            assert r[key] == res[0][key]
        assert r["ep_exchange"]["tokens"] == res[0]["ep_exchange"]["tokens"]
    moe_cfg = dict(hidden_size=512, num_hidden_layers=2, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=64,
                   vocab_size=2048, rms_norm_eps=1e-5, rope_theta=1e6, tie_word_embeddings=False, num_experts=8, num_experts_per_tok=2,
                   moe_intermediate_size=1024, moe_mode="mixtral", norm_topk_prob=0, qk_norm=False)
    single = engine.Model(max_context=512, **moe_cfg)
    single.synth_weights()
    moe_prompt = synth.prompt_ids(200, moe_cfg["vocab_size"])
    want = [int(single.prefill(moe_prompt))] + [int(x) for x in single.decode(6)]
    single.reset()
    want += [int(single.prefill(moe_prompt[:9]))] + [int(x) for x in single.decode(3)]
    single.close()
    assert res[0]["ep_exchange"]["tokens"] == want


def test_tensor_parallel_oproj_in_attention_launch_is_bit_identical(omx, monkeypatch):
    """Tensor parallel: the rank's f32 partial of the O projection comes out of the attention launch (attn_step.hip, o_out_f32) with
    the arithmetic of the separate EPI_F32 GEMV -- two ranks on one GPU through the in-process communicator, tokens and last logits
    equal with the O projection inside (default) and outside (OMX_ATTN_OPROJ=0) the launch.  The shape is one whose shards qualify
    (16 heads / 8 KV heads over 2 ranks at head_dim 128: K = 1024 per rank)."""
    from ominix_mlx_amd import comm, engine
    cfg = dict(hidden_size=1024, num_hidden_layers=2, intermediate_size=3072, num_attention_heads=16, num_key_value_heads=8, head_dim=128,
               vocab_size=4096, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
    prompt = synth.prompt_ids(24, cfg["vocab_size"])
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")

    def run_all():
        group = comm.LoopbackGroup(2, 1 << 20)
        models = []
        for r in range(2):
            m = engine.Model(max_context=2304, tp_rank=r, tp_size=2, **cfg)
            m.synth_weights()
            m.set_comm(group.rank_comm(r), group.allreduce_fn)
            models.append(m)

        def run(r):
            first = models[r].prefill(prompt)
            toks = [int(first)] + [int(x) for x in models[r].decode(12)]
            return toks, models[r].last_logits()

        outs = comm.run_ranks(2, run, group)
        for m in models:
            m.close()
        return outs

    fused = run_all()
    monkeypatch.setenv("OMX_ATTN_OPROJ", "0")
    plain = run_all()
    for r in range(2):
        assert fused[r][0] == plain[r][0]
        np.testing.assert_array_equal(fused[r][1], plain[r][1])
    assert fused[0][0] == fused[1][0]


def test_tensor_parallel_at_real_shard_shapes(omx):
    """Pre-flight of `bench.py --gpus N` (tools/tp_shapes_check.py): 2 layers of the REAL Qwen3-8B shard shapes at TP = 2, 4, 8 -- N
    engines on N host threads of one GPU through the in-process communicator -- batched TP prefill of 192 tokens + 12 decode steps.
    All ranks emit the same tokens, and the first token equals the single-GPU engine's (later ones may part at near-ties)."""
    import ast
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "tp_shapes_check.py")], capture_output=True, text=True, timeout=900,
                       stdin=subprocess.DEVNULL)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("tp")]
    assert len(lines) == 4
    ref = ast.literal_eval(lines[0].split(None, 1)[1])
    for ln in lines[1:]:
        assert "ranks agree: True" in ln
        toks = ast.literal_eval(ln.split("|")[0].split(None, 1)[1])
        assert toks[0] == ref[0]


ENGINE_CONFIGS = {k: v for k, v in OPROJ_CONFIGS.items() if not v.attention_bias}


@pytest.mark.parametrize("name", list(ENGINE_CONFIGS))
@pytest.mark.parametrize("n_prompt", [1000, 5])
def test_persistent_step_is_bit_identical(omx, monkeypatch, name, n_prompt):
    """csrc/step_engine.hip -- every layer of a decode step in ONE persistent launch (one loader wave streaming weights and K/V into
    an LDS ring per CU, three consumer waves, op outputs handed between CUs as tagged granules; OMX_STEP_ENGINE=1), or the hybrid step
    (=2: [gate/up, down, next q/k/v] as one engine segment between two attention launches) -- reproduces the launch-per-op step
    exactly: same tokens and bit-equal logits as OMX_STEP_ENGINE=0, in graph and eager form, across a context-bucket boundary
    (1000 -> 1039 crosses 1024) and from a nearly empty cache (most attention splits idle)."""
    from conftest import needs_experiments
    needs_experiments(omx)
    cfg = ENGINE_CONFIGS[name]
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    outs = {}
    for mode in ("0", "1", "eager", "2", "2eager"):
        monkeypatch.setenv("OMX_STEP_ENGINE", mode[0] if mode[0] in "012" else "1")
        monkeypatch.setenv("OMX_NO_GRAPH", "1" if mode.endswith("eager") else "0")
        m = _engine(omx, cfg, max_context=1280)
        toks = np.concatenate([[m.prefill(prompt)], m.decode(40)])
        outs[mode] = (toks, m.last_logits())
        m.close()
    for mode in ("1", "eager", "2", "2eager"):
        np.testing.assert_array_equal(outs["0"][0], outs[mode][0])
        np.testing.assert_array_equal(outs["0"][1], outs[mode][1])


@pytest.mark.parametrize("name", ["h32_kv8_d128_nvw8", "h16_kv8_d128_nvw4"])
@pytest.mark.parametrize("quant", [0, 4])
def test_aql_replay_is_bit_identical(omx, monkeypatch, name, quant):
    """csrc/aql_step.hip -- the step's launches recorded once and replayed as raw AQL packets on the engine's own HSA queue
    (OMX_STEP_AQL=1: kernel descriptors looked up in the executables HIP loaded, kernargs = recorded arguments + code-object-v5
    hidden arguments) -- reproduces the hipGraph step exactly: same tokens, bit-equal logits, across a context-bucket boundary
    (the program is rebuilt with the graphs) and in calls that straddle it (those fall back to the graph)."""
    from conftest import needs_experiments
    needs_experiments(omx)
    cfg = OPROJ_CONFIGS[name]
    prompt = synth.prompt_ids(1000, cfg.vocab_size)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OMX_STEP_AQL", mode)
        if quant:
            from ominix_mlx_amd import engine
            m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                             num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                             vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                             tie_word_embeddings=cfg.tie_word_embeddings, max_context=1280, qk_norm=cfg.qk_norm,
                             quantization={"bits": 4, "group_size": 64})
            m.synth_weights()
        else:
            m = _engine(omx, cfg, max_context=1280)
        toks = np.concatenate([[m.prefill(prompt)], m.decode(10), m.decode(30), m.decode(5)])   # 1010 .. 1040 crosses 1024 inside a call
        outs[mode] = (toks, m.last_logits(), m.decode_path())
        m.close()
    assert outs["1"][2] == "aql", "the AQL program was not built (OMX_STEP_AQL_VERBOSE=1 prints why)"
    np.testing.assert_array_equal(outs["0"][0], outs["1"][0])
    np.testing.assert_array_equal(outs["0"][1], outs["1"][1])


def test_persistent_step_options_are_bit_identical(omx, monkeypatch):
    """The engine's tuning knobs (sweeping waves per edge, fills in flight) change timing only."""
    from conftest import needs_experiments
    needs_experiments(omx)
    cfg = ENGINE_CONFIGS["h32_kv8_d128_nvw8"]
    prompt = synth.prompt_ids(300, cfg.vocab_size)
    outs = []
    for nsweep, inflight in (("1", "2"), ("3", "3"), ("3", "1")):
        monkeypatch.setenv("OMX_STEP_ENGINE", "1")
        monkeypatch.setenv("OMX_SE_NSWEEP", nsweep)
        monkeypatch.setenv("OMX_SE_INFLIGHT", inflight)
        m = _engine(omx, cfg, max_context=512)
        toks = np.concatenate([[m.prefill(prompt)], m.decode(24)])
        outs.append((toks, m.last_logits()))
        m.close()
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0][0], o[0])
        np.testing.assert_array_equal(outs[0][1], o[1])


def test_set_weight_checks_the_size_in_c(omx):
    """omx_qwen3_set_weight takes the tensor's byte length and refuses one that disagrees with the config (the engine reads raw
    pointers; the reference raises a shape error on load) -- checked in C, so a Rust caller following INTEGRATION.md section 3 is
    covered too, not only the Python loader."""
    import ctypes
    from ominix_mlx_amd import engine, ops
    cfg = CONFIGS["gqa4_d128"]
    m = _engine(omx, cfg)
    t = ops.Tensor.from_numpy(np.zeros((cfg.hidden_size, cfg.hidden_size), np.float32), "bf16")      # q_proj wants [H*D, hidden]
    lib = omx.lib
    assert lib.omx_qwen3_set_weight(m._h, b"model.layers.0.self_attn.k_proj.weight", t.ptr, t.nbytes) != 0
    assert b"ShapeMismatch" in lib.omx_last_error()
    lib.omx_clear_error()
    assert lib.omx_qwen3_set_weight(m._h, b"model.layers.0.self_attn.q_proj.weight", t.ptr, t.nbytes) == 0   # 8 heads x 128 = 1024 rows
    m.close()
