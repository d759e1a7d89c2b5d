"""The decode engine with a sparse-MoE feed-forward in every layer: Qwen3-MoE (qwen3-mlx/src/qwen3_moe.rs:304-508) and Mixtral
(mixtral-mlx/src/model.rs:96-347: no q/k norm, top-2 softmax over the selected logits).  Oracle = Qwen3Oracle with the MoE
block of oracle/ref_moe.py.  Tolerances as the dense engine (tests/test_gpu_qwen3.py); a router near-tie (k-th / (k+1)-th
logit closer than the bf16 resolution of the logits) legitimately sends a token to another expert, so a step is only
compared while the oracle's routing margins of that step exceed that resolution."""
import numpy as np
import pytest

from oracle import ref_core as rc
from oracle import ref_moe as rm
from oracle import ref_qwen3 as rq
from oracle import synth

pytestmark = pytest.mark.gpu

CONFIGS = {
    "qwen3_moe": rq.Qwen3Config(1024, 2, 3072, 8, 2, 128, 2048, 1e-6, 1e6, False, None, 40960, 8, 2, 512, "qwen3_moe", True, True),
    "qwen3_moe_no_renorm_top4": rq.Qwen3Config(512, 2, 1024, 4, 2, 128, 1024, 1e-6, 1e6, True, None, 40960, 16, 4, 512, "qwen3_moe", False, True),
    "mixtral": rq.Qwen3Config(1024, 2, 1024, 8, 2, 128, 2048, 1e-5, 1e6, False, None, 40960, 4, 2, 1024, "mixtral", False, False),
}
# wide enough for PACKED shards on four ranks (round 5: a rank's attention width, expert columns and vocabulary are multiples of 512)
WIDE = rq.Qwen3Config(2048, 2, 2048, 16, 4, 128, 2048, 1e-5, 1e6, False, None, 40960, 4, 2, 2048, "mixtral", False, False)


def _engine(omx, cfg, weights=None, max_context=256, quantization=None):
    from ominix_mlx_amd import engine
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                     vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                     tie_word_embeddings=cfg.tie_word_embeddings, max_context=max_context, num_experts=cfg.num_experts,
                     num_experts_per_tok=cfg.num_experts_per_tok, moe_intermediate_size=cfg.moe_intermediate_size,
                     moe_mode=cfg.moe_mode, norm_topk_prob=cfg.norm_topk_prob, qk_norm=cfg.qk_norm, quantization=quantization)
    m.synth_weights() if weights is None else m.load_weights(weights)
    return m


@pytest.mark.parametrize("name", list(CONFIGS))
def test_moe_engine_decode_matches_oracle(omx, name):
    cfg = CONFIGS[name]
    weights = rq.synth_weights(cfg)
    oracle = rq.Qwen3Oracle(cfg, weights)
    n_prompt, n_new = 40, 8
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)
    m = _engine(omx, cfg)                                    # device generator == oracle/synth.py
    first = m.prefill(prompt)
    logits0 = m.last_logits()
    got = np.concatenate([[first], m.decode(n_new - 1)]).astype(np.uint32)
    assert m.decode_path() == "graph" and m.offset() == n_prompt + n_new - 1
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers) * 2     # two extra bf16 GEMMs per layer (experts)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: got {got[i]} want {ref_tokens[i]} with margin {margins[i]:.4f}"
            break
    else:
        np.testing.assert_array_equal(got, ref_tokens)
    # uploaded weights == synthesized weights, bit for bit
    m2 = _engine(omx, cfg, weights)
    got2 = np.concatenate([[m2.prefill(prompt)], m2.decode(n_new - 1)]).astype(np.uint32)
    np.testing.assert_array_equal(got2, got)
    np.testing.assert_array_equal(m2.last_logits(), m.last_logits())


def test_moe_engine_serial_and_batched_prefill_agree(omx, monkeypatch):
    """The batched prefill takes the grouped-GEMM MoE route, the token-serial one the expert-selected GEMVs; same tokens
    while routing margins are comfortable (checked through the logits bound)."""
    cfg = CONFIGS["qwen3_moe"]
    prompt = synth.prompt_ids(70, cfg.vocab_size)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OMX_PREFILL_SERIAL", mode)
        m = _engine(omx, cfg)
        outs[mode] = (m.prefill(prompt), m.last_logits())
    bound = 2.0 ** -7 * np.abs(outs["1"][1]).max() * np.sqrt(cfg.num_hidden_layers) * 2
    assert np.abs(outs["0"][1] - outs["1"][1]).max() <= bound


def test_moe_engine_rejects_bad_config(omx):
    from ominix_mlx_amd import engine
    base = dict(hidden_size=1024, num_hidden_layers=1, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
                vocab_size=1024, max_context=256)
    with pytest.raises(omx.OmxError, match="InvalidConfig"):
        engine.Model(**base, num_experts=4, num_experts_per_tok=5, moe_intermediate_size=512)
    with pytest.raises(omx.OmxError, match="InvalidConfig"):      # quantised experts: width must be a multiple of 512
        engine.Model(**base, num_experts=4, num_experts_per_tok=2, moe_intermediate_size=768, quantization={"bits": 4, "group_size": 64})
    with pytest.raises(omx.OmxError, match="InvalidConfig"):      # expert tensor parallel: 64-column granularity per rank, no EP on top
        engine.Model(**base, num_experts=4, num_experts_per_tok=2, moe_intermediate_size=320, tp_size=2)
    with pytest.raises(omx.OmxError, match="InvalidConfig"):
        engine.Model(**base, num_experts=4, num_experts_per_tok=2, moe_intermediate_size=512, tp_size=2, ep_size=2)
    engine.Model(**base, num_experts=4, num_experts_per_tok=2, moe_intermediate_size=512, tp_size=2).close()
    m = engine.Model(**base, num_experts=4, num_experts_per_tok=2, moe_intermediate_size=512)
    with pytest.raises(omx.OmxError, match="WeightNotFound"):
        m.prefill([1, 2, 3])


def test_load_mixtral_checkpoint_directory(omx, tmp_path):
    """mixtral_mlx::load_model (model.rs:466-600) on a bf16 checkpoint in the ORIGINAL layout (per-expert w1/w2/w3): config.json
    -> MoE engine arguments, `sanitize_weights` stacks the experts, the engine generates what it generates from the same
    tensors handed over directly."""
    import json
    from ominix_mlx_amd import loader
    cfg = CONFIGS["mixtral"]
    w = rq.synth_weights(cfg)
    d = str(tmp_path)
    json.dump({"model_type": "mixtral", "hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_hidden_layers,
               "intermediate_size": cfg.moe_intermediate_size, "num_attention_heads": cfg.num_attention_heads,
               "num_key_value_heads": cfg.num_key_value_heads, "vocab_size": cfg.vocab_size, "rms_norm_eps": cfg.rms_norm_eps,
               "rope_theta": cfg.rope_theta, "num_local_experts": cfg.num_experts, "num_experts_per_tok": cfg.num_experts_per_tok},
              open(f"{d}/config.json", "w"))
    raw = {}
    for k, v in w.items():
        if ".switch_mlp." in k:
            proj = k.split(".switch_mlp.")[1].split(".")[0]
            old = {"gate_proj": "w1", "down_proj": "w2", "up_proj": "w3"}[proj]
            for e in range(cfg.num_experts):
                raw[k.split(".switch_mlp.")[0] + f".experts.{e}.{old}.weight"] = v[e]
        else:
            raw[k] = v
    loader.write_safetensors(f"{d}/model.safetensors", {k: rc.to_bf16_bits(v) for k, v in raw.items()}, bf16_names=tuple(raw))
    prompt = synth.prompt_ids(24, cfg.vocab_size)
    m = loader.load_model(d, max_context=256)
    got = np.concatenate([[m.prefill(prompt)], m.decode(5)])
    ref = _engine(omx, cfg, w)
    want = np.concatenate([[ref.prefill(prompt)], ref.decode(5)])
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(m.last_logits(), ref.last_logits())


@pytest.mark.parametrize("name,bits", [("mixtral", 4), ("qwen3_moe", 4), ("mixtral", 8)])
def test_quantized_moe_engine_matches_oracle(omx, name, bits):
    """The reference's own Mixtral format: every Linear, the embedding, the router gate and the expert stacks are MLX affine
    triplets (mixtral-mlx/src/model.rs:560-600).  Device-side `synth_weights` = mlx quantize() of the synthetic bf16 model, the
    oracle runs quantized_matmul / gather_qmm on the same triplets.  Logit tolerance of the quantised dense engine, doubled
    for the two extra quantised GEMVs per layer."""
    cfg = CONFIGS[name]
    quant = {"bits": bits, "group_size": 64}
    qw = rq.quantize_weights(cfg, rq.synth_weights(cfg), bits, 64)
    oracle = rq.Qwen3Oracle(cfg, qw, quant=(bits, 64))
    prompt = synth.prompt_ids(24, cfg.vocab_size)
    ref_tokens, ref_logits = oracle.generate(prompt, 6, return_logits=True)
    m = _engine(omx, cfg, qw, quantization=quant)
    first = m.prefill(prompt)
    logits0 = m.last_logits()
    got = np.concatenate([[first], m.decode(5)]).astype(np.uint32)
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers) * 2 * np.sqrt(2)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound
    margins = rc.argmax_margin(ref_logits)
    for i in range(6):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound
            break
    # the device generator builds the same model
    m2 = _engine(omx, cfg, None, quantization=quant)
    got2 = np.concatenate([[m2.prefill(prompt)], m2.decode(5)]).astype(np.uint32)
    np.testing.assert_array_equal(got2, got)


@pytest.mark.parametrize("serial_prefill", ["0", "1"])
@pytest.mark.parametrize("name", ["mixtral", "qwen3_moe"])
def test_float16_quantized_moe_checkpoint_runs_in_float16(omx, monkeypatch, name, serial_prefill):
    """(serial_prefill 0: the 24-token prompt in the float16 matrix-core pass -- router per token exactly as the decode form computes it,
    expert stacks dequantised to float16, the grouped 256-row GEMMs' float16 instantiation; 1: through the decode step.)
    The reference's only Mixtral format -- 4-bit triplets (mixtral-mlx/src/model.rs:554-556 refuses anything else), float16 in the MLX
    community builds -- the way MLX runs it: float16 END TO END (nn/quantized.rs:361-385, ops/quantization.rs:226-279).  float16
    scales / biases / norm weights uploaded as they are, float16 activations in every kernel of the step (packed GEMVs incl. the
    router and the expert stacks, decode attention with a float16 cache, routing scores and the weighted sum rounded to float16),
    f32 accumulation.  Against the oracle in float16 on the same float16 VALUES within the quantised MoE engine's bound scaled to
    float16 (2^-10 instead of 2^-7); the bf16-activation computation of the same checkpoint is shown to lie outside it."""
    from ominix_mlx_amd import engine
    cfg, bits, group = CONFIGS[name], 4, 64
    qw = _f16_triplets(cfg, bits, group)
    prompt = synth.prompt_ids(24, cfg.vocab_size)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", serial_prefill)
    m = _engine(omx, cfg, qw, quantization={"bits": bits, "group_size": group, "scales_dtype": "float16"})
    first = m.prefill(prompt)
    logits0 = m.last_logits()
    got = np.concatenate([[first], m.decode(5)]).astype(np.uint32)
    m.close()
    f16w = {k: (v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in qw.items()}     # the f16 VALUES, exactly
    ref_tokens, ref_logits = rq.Qwen3Oracle(cfg, f16w, dt="f16", quant=(bits, group)).generate(prompt, 6, return_logits=True)
    bound = 2.0 ** -10 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers) * 2 * np.sqrt(2)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound, f"float16 MoE logits off by {np.abs(logits0 - ref_logits[0]).max():.5f} (bound {bound:.5f})"
    margins = rc.argmax_margin(ref_logits)
    for i in range(6):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound
            break
    ref_bf16 = rq.Qwen3Oracle(cfg, f16w, quant=(bits, group)).generate(prompt, 1, return_logits=True)[1]
    assert np.abs(ref_bf16[0] - ref_logits[0]).max() > bound       # bf16 activations on float16 triplets are a different computation


def _f16_triplets(cfg, bits=4, group=64):
    """a float16 MLX checkpoint of the synthetic model: packed words as quantize() gives them, scales / biases float16 (mlx quantize() works in
    the model's dtype), norm weights as they are (uploaded as float16 values)"""
    base = rq.synth_weights(cfg)
    qw = {}
    for k, arr in rq.quantize_weights(cfg, base, bits, group).items():
        if k.endswith((".scales", ".biases")):
            prefix = k.rsplit(".", 1)[0]
            w2 = base[prefix + ".weight"].reshape(-1, base[prefix + ".weight"].shape[-1])
            _, s32, b32 = rc.quantize(w2, group, bits)
            arr = (s32 if k.endswith(".scales") else b32).astype(np.float16).reshape(arr.shape)
        qw[k] = arr
    return qw


@pytest.mark.parametrize("mode,name,world", [("ep", "mixtral", 2), ("ep", "mixtral", 4), ("ep", "qwen3_moe", 4), ("tp", "mixtral", 2), ("tp", "wide", 4)])
def test_float16_packed_moe_checkpoint_on_several_ranks(omx, monkeypatch, mode, name, world):
    """Round 5: the checkpoint a user of mixtral-mlx actually has -- 4-bit triplets with FLOAT16 scales -- on more than one
    rank, in float16 end to end like on one: expert parallel (the rank's experts' packed GEMVs with the expert filter, float16 products
    summed in f32, one all-reduce, the fold with the float16 roundings: bit-identical to the single-GPU float16 engine) and expert
    tensor parallel (every expert's column slice, f32 slot partials all-reduced, omx_moe_combine_slots_ex in float16: ranks agree bit for
    bit, logits within the float16 bound of the single GPU).  The prompt goes through the decode step here (OMX_PREFILL_SERIAL=1); its batched form
    has its own test below."""
    from ominix_mlx_amd import comm, engine
    cfg = WIDE if name == "wide" else CONFIGS[name]
    bits, group_size = 4, 64
    qw = _f16_triplets(cfg, bits, group_size)
    quantization = {"bits": bits, "group_size": group_size, "scales_dtype": "float16"}
    prompt = synth.prompt_ids(20, cfg.vocab_size)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")      # token-serial prompts on both sides: compare like with like
    single = _engine(omx, cfg, qw, quantization=quantization)
    want = np.concatenate([[single.prefill(prompt)], single.decode(6)]).astype(np.uint32)
    want_logits = single.last_logits()
    single.close()
    group = comm.LoopbackGroup(world, 1 << 20)
    models = []
    for r in range(world):
        shard = dict(ep_rank=r, ep_size=world) if mode == "ep" else dict(tp_rank=r, tp_size=world)
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                         num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                         vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, num_experts=cfg.num_experts,
                         num_experts_per_tok=cfg.num_experts_per_tok, moe_intermediate_size=cfg.moe_intermediate_size,
                         moe_mode=cfg.moe_mode, norm_topk_prob=cfg.norm_topk_prob, qk_norm=cfg.qk_norm, quantization=quantization, **shard)
        m.load_weights(qw)
        m.set_comm(group.rank_comm(r), group.allreduce_fn)
        models.append(m)

    def run(r):
        m = models[r]
        toks = np.concatenate([[m.prefill(prompt)], m.decode(6)]).astype(np.uint32)
        return toks, m.last_logits()

    outs = comm.run_ranks(world, run, group)
    for m in models:
        m.close()
    group.close()
    if mode == "ep":
        for r in range(world):
            np.testing.assert_array_equal(outs[r][0], want)
            np.testing.assert_array_equal(outs[r][1], want_logits)
        return
    for r in range(1, world):
        np.testing.assert_array_equal(outs[r][0], outs[0][0])
    toks, logits = outs[0][0], np.concatenate([o[1] for o in outs])       # vocabulary shards
    bound = 2.0 ** -10 * float(np.abs(want_logits).max()) * np.sqrt(2 * cfg.num_hidden_layers) * 2
    first_diff = next((i for i in range(len(want)) if toks[i] != want[i]), len(want))
    assert first_diff >= 1, "the sharded float16 engines disagree with the single-GPU engine on the token after the prompt"
    if first_diff == len(want):
        assert float(np.abs(logits - want_logits).max()) <= bound, f"float16 expert-TP logits off by {np.abs(logits - want_logits).max():.5f} (bound {bound:.5f})"


@pytest.mark.parametrize("mode,name,world", [("ep", "mixtral", 2), ("ep", "qwen3_moe", 4), ("tp", "mixtral", 2), ("tp", "wide", 4)])
def test_float16_packed_moe_prompt_is_one_batched_pass_on_several_ranks(omx, monkeypatch, mode, name, world):
    """Round 6: a sharded float16 MoE model's PROMPT as one batched pass (token by token before) -- the router per
    token exactly as the decode form computes it, the rank's stacks dequantised to float16, its slots through the grouped GEMMs' float16
    instantiation, the f32 partial all-reduced once per layer and folded with the float16 roundings.  Expert parallel: the same values as the
    single-GPU float16 engine's batched pass (a slot's row does not depend on which other rows share its tile); expert tensor parallel:
    ranks agree bit for bit, logits within the float16 bound of the single GPU.  And the batched pass agrees with the token-serial one of
    the same shards within that bound."""
    from ominix_mlx_amd import comm, engine
    cfg = WIDE if name == "wide" else CONFIGS[name]
    bits, group_size = 4, 64
    qw = _f16_triplets(cfg, bits, group_size)
    quantization = {"bits": bits, "group_size": group_size, "scales_dtype": "float16"}
    prompt = synth.prompt_ids(40, cfg.vocab_size)          # 80 routed slots: the batched form (> 32), and > 16 tokens
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "0")
    single = _engine(omx, cfg, qw, quantization=quantization)
    want = np.concatenate([[single.prefill(prompt)], single.decode(4)]).astype(np.uint32)
    want_logits = single.last_logits()
    single.close()

    def sharded(serial):
        monkeypatch.setenv("OMX_PREFILL_SERIAL", serial)
        group = comm.LoopbackGroup(world, 1 << 22)
        models = []
        for r in range(world):
            shard = dict(ep_rank=r, ep_size=world) if mode == "ep" else dict(tp_rank=r, tp_size=world)
            m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                             num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                             vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                             tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, num_experts=cfg.num_experts,
                             num_experts_per_tok=cfg.num_experts_per_tok, moe_intermediate_size=cfg.moe_intermediate_size,
                             moe_mode=cfg.moe_mode, norm_topk_prob=cfg.norm_topk_prob, qk_norm=cfg.qk_norm, quantization=quantization, **shard)
            m.load_weights(qw)
            m.set_comm(group.rank_comm(r), group.allreduce_fn)
            models.append(m)

        def run(r):
            m = models[r]
            toks = np.concatenate([[m.prefill(prompt)], m.decode(4)]).astype(np.uint32)
            return toks, m.last_logits()

        outs = comm.run_ranks(world, run, group)
        for m in models:
            m.close()
        group.close()
        return outs

    outs = sharded("0")
    bound = 2.0 ** -10 * float(np.abs(want_logits).max()) * np.sqrt(2 * cfg.num_hidden_layers) * 2
    for r in range(1, world):
        np.testing.assert_array_equal(outs[r][0], outs[0][0])
    if mode == "ep":
        for r in range(world):
            np.testing.assert_array_equal(outs[r][0], want)
            np.testing.assert_array_equal(outs[r][1], want_logits)
        logits = outs[0][1]
    else:
        logits = np.concatenate([o[1] for o in outs])       # vocabulary shards
        first_diff = next((i for i in range(len(want)) if outs[0][0][i] != want[i]), len(want))
        assert first_diff >= 1, "the sharded float16 engines' batched prompt disagrees with the single-GPU engine on the token after the prompt"
        if first_diff == len(want):
            assert float(np.abs(logits - want_logits).max()) <= bound
    serial = sharded("1")
    s_logits = serial[0][1] if mode == "ep" else np.concatenate([o[1] for o in serial])
    if np.array_equal(serial[0][0], outs[0][0]):
        assert float(np.abs(logits - s_logits).max()) <= bound, f"batched vs token-serial float16 prompt: {np.abs(logits - s_logits).max():.5f} (bound {bound:.5f})"
    else:
        assert serial[0][0][0] == outs[0][0][0]            # (a later near-tie may flip; the token after the prompt does not)


@pytest.mark.parametrize("name,world,quant", [("qwen3_moe", 2, None), ("mixtral", 4, None), ("qwen3_moe_no_renorm_top4", 4, None),
                                              ("mixtral", 2, 4), ("mixtral", 4, 4), ("qwen3_moe", 4, 4), ("mixtral", 2, 8)])
@pytest.mark.parametrize("use_synth", [True, False])
def test_expert_parallel_engine_on_one_gpu(omx, monkeypatch, name, world, quant, use_synth):
    """SURVEY.md 8e row 2 at engine level: `world` ranks (host threads, in-process communicator standing in for RCCL), each
    holding E / world experts, attention and router replicated, ONE all-reduce of the f32 partial per MoE block.  The partials
    are sums of bf16-rounded products accumulated in f32, so the sharded run reproduces the single-GPU engine bit for bit.
    quant (round 5): the reference's REAL Mixtral format -- MLX-packed expert stacks (mixtral-mlx/src/model.rs:466-615 refuses anything
    else): a rank holds the packed triplets of ITS experts (ep.shard_experts on weight / scales / biases; the device generator quantises
    the rank's window of the logical stack), the packed GEMVs skip the slots routed elsewhere."""
    from ominix_mlx_amd import comm
    cfg = CONFIGS[name]
    quantization = {"bits": quant, "group_size": 64} if quant else None
    weights = rq.synth_weights(cfg)
    if quant:
        weights = rq.quantize_weights(cfg, weights, quant, 64)
    prompt = synth.prompt_ids(20, cfg.vocab_size)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")      # the sharded engines prefill token-serially: compare like with like
    single = _engine(omx, cfg, quantization=quantization)
    want = np.concatenate([[single.prefill(prompt)], single.decode(6)]).astype(np.uint32)
    want_logits = single.last_logits()
    group = comm.LoopbackGroup(world, 1 << 20)
    from ominix_mlx_amd import engine
    models = []
    for r in range(world):
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                         num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                         vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, num_experts=cfg.num_experts,
                         num_experts_per_tok=cfg.num_experts_per_tok, moe_intermediate_size=cfg.moe_intermediate_size,
                         moe_mode=cfg.moe_mode, norm_topk_prob=cfg.norm_topk_prob, qk_norm=cfg.qk_norm, ep_rank=r, ep_size=world,
                         quantization=quantization)
        m.synth_weights() if use_synth else m.load_weights(weights)
        m.set_comm(group.rank_comm(r), group.allreduce_fn)
        models.append(m)

    def run(r):
        m = models[r]
        toks = np.concatenate([[m.prefill(prompt)], m.decode(6)]).astype(np.uint32)
        return toks, m.last_logits()

    outs = comm.run_ranks(world, run, group)
    for r in range(world):
        np.testing.assert_array_equal(outs[r][0], want)
        np.testing.assert_array_equal(outs[r][1], want_logits)
    for m in models:
        m.close()
    group.close()


@pytest.mark.parametrize("name,world,batched,quant", [("mixtral", 2, False, None), ("qwen3_moe", 2, False, None), ("mixtral", 4, False, None),
                                                      ("mixtral", 2, True, None), ("qwen3_moe", 2, True, None), ("mixtral", 4, True, None),
                                                      ("mixtral", 2, False, 4), ("wide", 4, False, 4), ("wide", 2, True, 4), ("wide", 4, True, 4)])
def test_expert_tensor_parallel_engine_on_one_gpu(omx, monkeypatch, name, world, batched, quant):
    """Expert TENSOR parallelism (round 4; tp_size > 1 on a sparse-MoE model): attention heads sharded like the dense model, every
    expert's intermediate columns split over the ranks (gate / up rows, down columns), router replicated, per layer one all-reduce of
    the O partial and one of the routed slots' f32 down partials, then the weighted sum with the single-device roundings.  What
    expert parallelism cannot do -- a token's two experts streaming from ALL ranks -- at the price of partial sums that round
    differently from the single-GPU dot products: every rank must agree with every other rank bit for bit (rank-ordered sums),
    the synthetic shards must equal the host-sharded checkpoint (tp.py expert rules), and the logits must stay within the engine's
    bound of the single-GPU run (tokens equal wherever its top-1 margin allows).
    batched: a 200-token prompt as ONE pass -- the expert-parallel grouped-GEMM form over all experts at this rank's columns, the f32
    partial of every row's weighted sum all-reduced (engine.hip prefill_prefix_batched) -- against the single GPU's batched pass."""
    from ominix_mlx_amd import comm, engine
    # quant (round 5): MLX-packed expert stacks -- every expert's packed gate / up rows and the whole-group K slice of its down projection
    # on each rank (tp.shard on the three leaves; the device generator quantises the rank's windows); a prompt dequantises the rank's stacks
    cfg = WIDE if name == "wide" else CONFIGS[name]
    quantization = {"bits": quant, "group_size": 64} if quant else None
    weights = rq.synth_weights(cfg)
    if quant:
        weights = rq.quantize_weights(cfg, weights, quant, 64)
    n_prompt, ctx = (200, 512) if batched else (20, 256)
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "0" if batched else "1")
    single = _engine(omx, cfg, max_context=ctx, quantization=quantization)
    want = np.concatenate([[single.prefill(prompt)], single.decode(6)]).astype(np.uint32)
    want_logits = single.last_logits()
    single.close()
    results = {}
    for use_synth in (True, False):
        group = comm.LoopbackGroup(world, max(1 << 20, n_prompt * cfg.hidden_size * 4))
        models = []
        for r in range(world):
            m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                             num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                             vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                             tie_word_embeddings=cfg.tie_word_embeddings, max_context=ctx, num_experts=cfg.num_experts,
                             num_experts_per_tok=cfg.num_experts_per_tok, moe_intermediate_size=cfg.moe_intermediate_size,
                             moe_mode=cfg.moe_mode, norm_topk_prob=cfg.norm_topk_prob, qk_norm=cfg.qk_norm, tp_rank=r, tp_size=world,
                             quantization=quantization)
            m.synth_weights() if use_synth else m.load_weights(weights)
            m.set_comm(group.rank_comm(r), group.allreduce_fn)
            models.append(m)

        def run(r):
            m = models[r]
            toks = np.concatenate([[m.prefill(prompt)], m.decode(6)]).astype(np.uint32)
            return toks, m.last_logits()

        outs = comm.run_ranks(world, run, group)
        for m in models:
            m.close()
        group.close()
        for r in range(1, world):                      # vocabulary shards: the token is all-reduced, the logits are each rank's rows
            np.testing.assert_array_equal(outs[r][0], outs[0][0])
        results[use_synth] = (outs[0][0], np.concatenate([o[1] for o in outs]))
    np.testing.assert_array_equal(results[True][0], results[False][0])
    np.testing.assert_array_equal(results[True][1], results[False][1])
    toks, logits = results[True]
    bound = 2.0 ** -7 * float(np.abs(want_logits).max()) * np.sqrt(2 * cfg.num_hidden_layers)
    srt = np.sort(want_logits)
    if srt[-1] - srt[-2] > 2 * bound:                  # (the last step's margin decides whether the last token may differ)
        assert toks[-1] == want[-1]
    first_diff = next((i for i in range(len(want)) if toks[i] != want[i]), len(want))
    if first_diff == len(want):                        # same token sequence -> same cache contents: the logits are comparable
        assert float(np.abs(logits - want_logits).max()) <= bound, f"expert-TP logits off by {np.abs(logits - want_logits).max():.4f} (bound {bound:.4f})"
    else:
        assert first_diff >= 1, "the sharded engines already disagree with the single-GPU engine on the token after the prompt"


def test_expert_parallel_load_from_bf16_checkpoint_files(omx, tmp_path, monkeypatch):
    """ADVICE r1 (high), EP side: ep.shard_experts on raw-bits (Bf16Bits) expert stacks read from a BF16 safetensors file must
    upload the same weights as float arrays do -- two expert-parallel ranks loaded from files equal the single-GPU engine."""
    import json
    from ominix_mlx_amd import comm, loader
    cfg = CONFIGS["mixtral"]
    w = rq.synth_weights(cfg)
    d = str(tmp_path)
    json.dump({"model_type": "mixtral", "hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_hidden_layers,
               "intermediate_size": cfg.moe_intermediate_size, "num_attention_heads": cfg.num_attention_heads,
               "num_key_value_heads": cfg.num_key_value_heads, "vocab_size": cfg.vocab_size, "rms_norm_eps": cfg.rms_norm_eps,
               "rope_theta": cfg.rope_theta, "num_local_experts": cfg.num_experts, "num_experts_per_tok": cfg.num_experts_per_tok},
              open(f"{d}/config.json", "w"))
    loader.write_safetensors(f"{d}/model.safetensors", {k: rc.to_bf16_bits(v) for k, v in w.items()}, bf16_names=tuple(w))
    prompt = synth.prompt_ids(20, cfg.vocab_size)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")
    single = _engine(omx, cfg, w)
    want = np.concatenate([[single.prefill(prompt)], single.decode(5)]).astype(np.uint32)
    world = 2
    group = comm.LoopbackGroup(world, 1 << 20)
    models = [loader.load_model(d, max_context=256, ep_rank=r, ep_size=world) for r in range(world)]
    for r, m in enumerate(models):
        m.set_comm(group.rank_comm(r), group.allreduce_fn)

    def run(r):
        m = models[r]
        return np.concatenate([[m.prefill(prompt)], m.decode(5)]).astype(np.uint32), m.last_logits()

    outs = comm.run_ranks(world, run, group)
    for r in range(world):
        np.testing.assert_array_equal(outs[r][0], want)
        np.testing.assert_array_equal(outs[r][1], single.last_logits())
    for m in models:
        m.close()
    group.close()


@pytest.mark.parametrize("name", ["mixtral", "qwen3_moe"])
def test_moe_weighted_sum_folded_into_next_gemv_is_bit_identical(omx, monkeypatch, name):
    """The decode step's MoE block without its weighted-sum launch (engine.hip moe_fold: the expert down GEMVs store
    bf16(bf16(y_j) * score_j) as f32, the next GEMV's prologue folds bf16(resid + bf16(sum_j)) -- gemv.hip x_partial_n, moe.hip
    omx_moe_block_partials) against the block with moe_combine_kernel (OMX_MOE_FOLD=0): same roundings, same order -> same tokens and
    the same last logits."""
    from ominix_mlx_amd import engine
    cfgs = {"mixtral": dict(hidden_size=512, num_hidden_layers=3, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2,
                            head_dim=64, vocab_size=2048, rms_norm_eps=1e-5, rope_theta=1e6, tie_word_embeddings=False,
                            num_experts=8, num_experts_per_tok=2, moe_intermediate_size=1024, moe_mode="mixtral", norm_topk_prob=0, qk_norm=False),
            "qwen3_moe": dict(hidden_size=512, num_hidden_layers=2, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2,
                              head_dim=64, vocab_size=2048, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False,
                              num_experts=16, num_experts_per_tok=4, moe_intermediate_size=512, moe_mode="qwen3_moe", norm_topk_prob=1)}
    from oracle import synth
    prompt = synth.prompt_ids(24, 2048)

    def run():
        m = engine.Model(max_context=128, **cfgs[name])
        m.synth_weights()
        toks = [int(m.prefill(prompt))] + [int(t) for t in m.decode(16)]
        logits = m.last_logits()
        m.close()
        return toks, logits

    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")      # the prompt through the step too
    monkeypatch.setenv("OMX_MOE_FOLD", "1")            # (default: top-2 routing only)
    folded = run()
    monkeypatch.setenv("OMX_MOE_FOLD", "0")
    plain = run()
    assert folded[0] == plain[0]
    np.testing.assert_array_equal(folded[1], plain[1])


@pytest.mark.parametrize("name,world", [("qwen3_moe", 2), ("mixtral", 4)])
def test_expert_parallel_batched_prefill_on_one_gpu(omx, monkeypatch, name, world):
    """Round 3: a prompt under expert parallelism is ONE batched pass (until now: one decode step per token).  Every rank routes all
    rows, runs the grouped matrix-core GEMMs over the slots of ITS experts (device-side plan: the other slots are sorted into a
    trailing pseudo-expert that gets no tiles) and one all-reduce per layer sums the [T, hidden] f32 partials.  A 200-token prompt
    (400 / 800 routed slots) through `world` sharded engines on this GPU must give the single-GPU engine's tokens, and -- with top-2
    routing, where the summation order cannot matter -- its logits bit for bit (top-k > 2: within a few bf16 ulps, see below)."""
    from ominix_mlx_amd import comm, engine
    cfg = CONFIGS[name]
    prompt = synth.prompt_ids(200, cfg.vocab_size)
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "0")
    single = _engine(omx, cfg, max_context=512)
    want = np.concatenate([[single.prefill(prompt)], single.decode(6)]).astype(np.uint32)
    want_logits = single.last_logits()
    single.close()
    group = comm.LoopbackGroup(world, 200 * cfg.hidden_size * 4)
    models = []
    for r in range(world):
        m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                         num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                         vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                         tie_word_embeddings=cfg.tie_word_embeddings, max_context=512, num_experts=cfg.num_experts,
                         num_experts_per_tok=cfg.num_experts_per_tok, moe_intermediate_size=cfg.moe_intermediate_size,
                         moe_mode=cfg.moe_mode, norm_topk_prob=cfg.norm_topk_prob, qk_norm=cfg.qk_norm, ep_rank=r, ep_size=world)
        m.synth_weights()
        m.set_comm(group.rank_comm(r), group.allreduce_fn)
        models.append(m)

    def run(r):
        m = models[r]
        toks = np.concatenate([[m.prefill(prompt)], m.decode(6)]).astype(np.uint32)
        return toks, m.last_logits(), m.last_prefill_ms()

    outs = comm.run_ranks(world, run, group)
    for r in range(world):
        np.testing.assert_array_equal(outs[r][0], want)
        np.testing.assert_array_equal(outs[r][0], outs[0][0])
        np.testing.assert_array_equal(outs[r][1], outs[0][1])            # every rank holds the same logits
        if cfg.num_experts_per_tok <= 2:
            np.testing.assert_array_equal(outs[r][1], want_logits)       # two summands: the order cannot matter
        else:
            # top-k > 2: the ranks' partials add the k products in another order than the single device's slot order; f32 sums of
            # bf16-rounded products are exact unless their exponents spread over > 13 bits, so a few hidden-state elements differ by
            # one bf16 ulp over 200 rows x layers -- the logits then move by at most a few ulps
            assert np.abs(outs[r][1] - want_logits).max() <= 2.0 ** -6 * np.abs(want_logits).max()
    for m in models:
        m.close()
    group.close()


ROUTE_CONFIGS = {
    "qwen3_moe": CONFIGS["qwen3_moe"],
    "mixtral": CONFIGS["mixtral"],
    # the full row width of Mixtral-8x7B (two router vectors per thread of the GEMV block), 8 experts
    "mixtral_h4096": rq.Qwen3Config(4096, 1, 512, 32, 8, 128, 1024, 1e-5, 1e6, False, None, 40960, 8, 2, 512, "mixtral", False, False),
    "qwen3_moe_top1_h2048": rq.Qwen3Config(2048, 2, 512, 16, 4, 128, 1024, 1e-6, 1e6, False, None, 40960, 8, 1, 512, "qwen3_moe", False, True),
}


@pytest.mark.parametrize("name", list(ROUTE_CONFIGS))
@pytest.mark.parametrize("fold", ["1", "0"])
def test_router_inside_the_expert_gemv_is_bit_identical(omx, monkeypatch, name, fold):
    """One token, few experts: no router launch -- every block of the experts' gate/up GEMV normalises and routes the row itself
    (csrc/gemv.hip PRO_ROUTE), with the router kernel's own sums (512-thread form: per-thread squares, wave sums, serial sum over the
    waves; one wave per expert logit).  Same tokens and bit-equal logits as OMX_MOE_ROUTE_FUSED=0, with the experts' weighted sum folded
    into the next GEMV (the default for top-2) or combined by its own launch, graph and eager."""
    cfg = ROUTE_CONFIGS[name]
    prompt = synth.prompt_ids(24, cfg.vocab_size)
    monkeypatch.setenv("OMX_MOE_FOLD", fold)
    outs = {}
    for mode in ("0", "1", "eager"):
        monkeypatch.setenv("OMX_MOE_ROUTE_FUSED", "0" if mode == "0" else "1")
        monkeypatch.setenv("OMX_NO_GRAPH", "1" if mode == "eager" else "0")
        m = _engine(omx, cfg)
        toks = np.concatenate([[m.prefill(prompt)], m.decode(24)])
        outs[mode] = (toks, m.last_logits())
        m.close()
    for mode in ("1", "eager"):
        for a, b in zip(outs["0"], outs[mode]):
            np.testing.assert_array_equal(a, b)
